// fHE/keygen.hpp — the one construction every key of the scheme is made of, written once:
//     b = -(e + a (*) s)  [ + c (*) t  on selected limbs ]          (all operands in the NTT basis)
// with a uniform, e small (sampled on the host, transformed on the GPU), s the secret.  PK, MultKey, RotKey and MixedRotKey
// are instances (reference: include/fHE/keys.hpp:39-47, src/mult_key.cpp:3-30, src/rot_key.cpp:23-120 describe the same
// algebra key by key).  Host plumbing: key generation runs once; only the forward NTT of e goes through the C-ABI.
#pragma once
#include "fHE/context.hpp"
namespace fHE {
struct SK;
namespace detail {
using word = yell::params::value_type;
using wide = yell::params::gt_value_type;

// forward / backward NTT of `count` consecutive limbs that live under the moduli first, first+1, ... (one device call)
inline void transform_limbs(word *limbs, size_t first_modulus, size_t count, bool inverse) {
  if (!count) return;
  std::vector<uint32_t> under(count);
  for (size_t k = 0; k < count; ++k) under[k] = (uint32_t)(first_modulus + k);
  fhe_hip_ctx *dev = yell::runtime::get().ctx;
  yell::check(inverse ? fhe_hip_ntt_backward(dev, limbs, under.data(), count, FHE_HIP_MEM_HOST)
                      : fhe_hip_ntt_forward(dev, limbs, under.data(), count, FHE_HIP_MEM_HOST), "key generation NTT");
}

// where the residues of the secret under global modulus index m are kept (normal primes / special primes apart)
struct SecretView {
  const context::poly_t *normal, *special;
  const word *under(size_t m) const {
    const size_t L = normal->moduli_count();
    return m < L ? normal->cptr_at(m) : special->cptr_at(m - L);
  }
};

// b <- -(e + a (*) s) on every limb of b (limb k under modulus k); b's previous content is discarded
inline void mask_with_secret(context::poly_t &b, context::poly_t const &a, SecretView const &s) {
  const size_t n = context::degree;
  b.set(context::gauss_struct(&context::fg_prng));
  b.forward();
  for (size_t k = 0; k < b.moduli_count(); ++k) {
    const word p = yell::params::P[k];
    word *bk = b.ptr_at(k);
    const word *ak = a.cptr_at(k), *sk = s.under(k);
    for (size_t d = 0; d < n; ++d) {
      const word t = (word)(((wide)ak[d] * sk[d] + bk[d]) % p);
      bk[d] = t ? p - t : 0;
    }
  }
}

// b[limb] += c * t  (mod the limb's prime)
inline void add_scaled_limb(context::poly_t &b, size_t limb, word c, const word *t) {
  const size_t n = context::degree;
  const word p = yell::params::P[limb];
  word *bl = b.ptr_at(limb);
  c %= p;
  for (size_t d = 0; d < n; ++d) bl[d] = (word)(((wide)c * t[d] + bl[d]) % p);
}

// product of the special primes P[first .. first+count) modulo the normal prime P[m]
inline word special_product_mod(size_t first, size_t count, size_t m) {
  const word p = yell::params::P[m];
  word acc = 1;
  for (size_t k = 0; k < count; ++k) acc = (word)(((wide)acc * (yell::params::P[first + k] % p)) % p);
  return acc;
}

// key material is neither copied nor moved (the reference deletes those members on every key type)
struct pinned_object {
  pinned_object() = default;
  pinned_object(pinned_object const &) = delete;
  pinned_object &operator=(pinned_object const &) = delete;
};

// keys are identified towards the device by a number that is never reused in this process (not by their address:
// a key destroyed and another built in its place must not pass for the first one)
inline uint64_t fresh_key_id() { static uint64_t next = 0; return ++next; }
}  // namespace detail
}  // namespace fHE

// fHE/encryptor.hpp — Encryptor with the reference's public shape (include/fHE/encryptor.hpp:11-23).  The three small
// polynomials (u, e0, e1) are sampled on the host; everything else — three forward NTTs and the two u*pk products of
// src/encryptor.cpp:15-35 — is ONE device call, fhe_hip_encrypt.
#pragma once
#include "fHE/cipher.hpp"
#include "fHE/encoder.hpp"
#include "fHE/keys.hpp"
#include <complex>
namespace fHE {
struct Encryptor {
  explicit Encryptor(std::shared_ptr<Encoder> encoder = nullptr) : encoder(encoder) {}
  ~Encryptor() {}

  // msg in the power basis; a message with fewer limbs than the ciphertext contributes its first limb under every prime
  bool encrypt(Cipher *rop, context::poly_t const &msg, PK const &pk) const {
    if (!rop) return false;
    const size_t Li = rop->moduli_count(), n = context::degree, have = msg.moduli_count();
    if (Li == 0 || have == 0 || Li > pk.bx.moduli_count()) return false;
    context::poly_t u(Li, yell::ZO_dist());
    context::poly_t e0m(Li, context::gauss_struct(&context::fg_prng));
    context::poly_t e1(Li, context::gauss_struct(&context::fg_prng));
    for (size_t m = 0; m < Li; ++m) {  // e0 + msg, still in the power basis
      const detail::word p = yell::params::P[m], *src = msg.cptr_at(have == Li ? m : 0);
      detail::word *dst = e0m.ptr_at(m);
      for (size_t d = 0; d < n; ++d) { const detail::word t = dst[d] + src[d] % p; dst[d] = t >= p ? t - p : t; }
    }
    // the result is written where it will live: one device block [half][limb][N] that the ciphertext's halves then view;
    // the public key's device copy is uploaded once and stays with the key
    auto out = std::make_shared<yell::dev_block>(yell::runtime::get().ctx, 2 * Li * n);
    if (fhe_hip_encrypt(yell::runtime::get().ctx, out->p, u.cdev(), e0m.cdev(), e1.cdev(), pk.bx.cdev(), pk.ax.cdev(),
                        (uint32_t)Li, 1, FHE_HIP_MEM_DEVICE) != FHE_HIP_OK) return false;
    rop->adopt_device(out, 0, Li);
    u.clear(); e0m.clear(); e1.clear();
    return true;
  }
  bool encrypt(Cipher *rop, double const &value, PK const &pk) const { return encode_then_encrypt(rop, std::vector<double>{value}, pk); }
  bool encrypt(Cipher *rop, std::vector<double> const &values, PK const &pk) const { return encode_then_encrypt(rop, values, pk); }
  bool encrypt(Cipher *rop, std::vector<std::complex<double>> const &values, PK const &pk) const { return encode_then_encrypt(rop, values, pk); }

 private:
  template <class Values> bool encode_then_encrypt(Cipher *rop, Values const &values, PK const &pk) const {
    if (!rop || !encoder) return false;
    context::poly_t plain(context::nr_ctxt_moduli);
    encoder->encode(&plain, values, context::encoder_scale);
    rop->scale(context::encoder_scale);
    return encrypt(rop, plain, pk);
  }
  std::shared_ptr<Encoder> encoder;
};
}  // namespace fHE

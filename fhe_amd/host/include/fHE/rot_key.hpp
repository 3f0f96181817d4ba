// fHE/rot_key.hpp — rotation keys with the reference's public shape (include/fHE/rot_key.hpp:9-118):
//   RotKey      beta, alpha over L + K moduli, beta = -(e + alpha s) + P sigma_g(s) on the normal limbs (P = product of the special primes)
//   MixedRotKey L pairs over L + 1 moduli (one special prime q_K), beta_j = -(e_j + alpha_j s) + q_K sigma_g(s) on limb j only
//   RotKeySet   left-rotation keys for the offsets 1, 2, 4, ..., N/4
// sigma_g = X -> X^g, g = 3^k mod 2N for a left rotation by k slots.
#pragma once
#include "fHE/keys.hpp"
#include "fHE/yell.hpp"
#include <memory>
#include <vector>
namespace fHE {
// galois element of a rotation: offset <= 0 rotates left by |offset| slots, offset > 0 right by offset (= left by N/2 - offset)
inline size_t rotation_offset_galois(int offset) {
  const size_t slots = context::degree >> 1, two_n = context::degree << 1;
  size_t k = (size_t)(offset < 0 ? -(long)offset : (long)offset) % slots;
  if (offset > 0) k = slots - k;
  size_t g = 1, base = 3;  // 3^k mod 2N by squaring
  for (; k; k >>= 1) {
    if (k & 1) g = (g * base) & (two_n - 1);
    base = (base * base) & (two_n - 1);
  }
  return g;
}

namespace detail {
template <class Dir> inline size_t galois_of(uint16_t offset, Dir dir) { return rotation_offset_galois(dir == Dir::RIGHT ? (int)offset : -(int)offset); }
// sigma_g(s) on the normal limbs (NTT basis: a permutation, done by the device)
inline context::poly_t rotated_secret(SK const &sk, size_t galois) {
  context::poly_t r(sk.sx);
  apply_galois(&r, galois);
  return r;
}
}  // namespace detail

struct RotKey : detail::pinned_object {
  enum class Direction { LEFT, RIGHT };
  const size_t galois;
  context::poly_t beta;   //! L + K moduli
  context::poly_t alpha;  //! L + K moduli
  const uint64_t id = detail::fresh_key_id();

  explicit RotKey(SK const &sk, uint16_t offset, Direction dir)
      : galois(detail::galois_of(offset, dir)),
        beta(context::nr_ctxt_moduli + context::nr_sp_primes),
        alpha(context::nr_ctxt_moduli + context::nr_sp_primes, yell::uniform{}) {
    const size_t L = context::nr_ctxt_moduli, K = context::nr_sp_primes;
    detail::mask_with_secret(beta, alpha, sk.view());
    context::poly_t turned = detail::rotated_secret(sk, galois);
    for (size_t m = 0; m < L; ++m) detail::add_scaled_limb(beta, m, detail::special_product_mod(L, K, m), turned.cptr_at(m));
    turned.clear();
  }
};

struct MixedRotKey : detail::pinned_object {
 private:
  using poly_t = context::poly_t;
  std::vector<poly_t> alpha, beta;  // L polynomials of L + 1 moduli each (limb L under the first special prime)

 public:
  enum class Direction { LEFT, RIGHT };
  const size_t galois;
  const uint64_t id = detail::fresh_key_id();

  explicit MixedRotKey(SK const &sk, uint16_t offset, Direction dir) : galois(detail::galois_of(offset, dir)) {
    const size_t L = context::nr_ctxt_moduli;
    context::poly_t turned = detail::rotated_secret(sk, galois);
    alpha.reserve(L); beta.reserve(L);
    for (size_t j = 0; j < L; ++j) {
      alpha.emplace_back(L + 1, yell::uniform{});
      beta.emplace_back(L + 1);
      detail::mask_with_secret(beta[j], alpha[j], sk.view());  // modulus L is the special prime: SecretView hands out ext_sx there
      detail::add_scaled_limb(beta[j], j, detail::special_product_mod(L, 1, j), turned.cptr_at(j));
    }
    turned.clear();
  }
  const poly_t *get_beta_at(size_t j) const { return &beta.at(j); }
  const poly_t *get_alpha_at(size_t j) const { return &alpha.at(j); }
  poly_t *get_beta_at(size_t j) { return &beta.at(j); }
  poly_t *get_alpha_at(size_t j) { return &alpha.at(j); }
};

template <class RotKeyType> struct RotKeySet : detail::pinned_object {
 private:
  std::vector<std::unique_ptr<RotKeyType>> by_log2_;  // entry i: left rotation by 2^i slots

 public:
  explicit RotKeySet(SK const &sk) {
    const size_t count = yell::runtime::get().logn - 1;  // log2(N/2) keys
    by_log2_.reserve(count);
    for (size_t i = 0; i < count; ++i) by_log2_.emplace_back(new RotKeyType(sk, (uint16_t)(1u << i), RotKeyType::Direction::LEFT));
  }
  // the key of a power-of-two offset, or nullptr when the set holds none for it
  const RotKeyType *get(uint16_t offset) const {
    if (offset == 0 || (offset & (offset - 1))) return nullptr;
    size_t i = 0;
    while ((1u << i) < offset) ++i;
    return i < by_log2_.size() ? by_log2_[i].get() : nullptr;
  }
};
}  // namespace fHE

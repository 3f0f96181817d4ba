// fHE/mult_key.hpp — relinearisation key with the reference's public shape (include/fHE/mult_key.hpp:5-38): L pairs
// (beta_j, alpha_j) of L-limb polynomials, beta_j = -(e_j + alpha_j s) + s^2 on limb j only (RNS-digit key, no special prime).
#pragma once
#include "fHE/keys.hpp"
#include <vector>
namespace fHE {
struct MultKey : detail::pinned_object {
 private:
  using poly_t = context::poly_t;
  std::vector<poly_t> alpha, beta;

 public:
  const uint64_t id = detail::fresh_key_id();
  explicit MultKey(SK const &sk) {
    const size_t L = context::nr_ctxt_moduli, n = context::degree;
    alpha.reserve(L); beta.reserve(L);
    std::vector<detail::word> s_squared(n);
    for (size_t j = 0; j < L; ++j) {
      alpha.emplace_back(L, yell::uniform{});
      beta.emplace_back(L);
      detail::mask_with_secret(beta[j], alpha[j], sk.view());
      const detail::word p = yell::params::P[j], *s = sk.sx.cptr_at(j);
      for (size_t d = 0; d < n; ++d) s_squared[d] = (detail::word)(((detail::wide)s[d] * s[d]) % p);
      detail::add_scaled_limb(beta[j], j, 1, s_squared.data());
    }
    std::fill(s_squared.begin(), s_squared.end(), 0);
  }
  // pair j of the key: beta_j / alpha_j, L limbs each, NTT basis
  poly_t const *get_beta_at(size_t j) const { return &beta.at(j); }
  poly_t *get_beta_at(size_t j) { return &beta.at(j); }
  poly_t const *get_alpha_at(size_t j) const { return &alpha.at(j); }
  poly_t *get_alpha_at(size_t j) { return &alpha.at(j); }
};
}  // namespace fHE

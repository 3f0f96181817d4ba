// fHE/mult_key.hpp — reference include/fHE/mult_key.hpp:5-38 + src/mult_key.cpp:3-30.
#pragma once
#include "fHE/context.hpp"
#include "fHE/keys.hpp"
#include <vector>
namespace fHE {
struct MultKey {
 private:
  using poly_t = context::poly_t;
  std::vector<poly_t> alpha;  // L polynomials of L moduli
  std::vector<poly_t> beta;

 public:
  explicit MultKey(SK const &sk) {
    const size_t degree = context::degree, L = context::nr_ctxt_moduli;
    for (size_t j = 0; j < L; ++j) { alpha.emplace_back(L, yell::uniform{}); beta.emplace_back(L); }
    auto sx_square(sk.sx);
    sx_square *= sk.sx;
    for (size_t j = 0; j < L; ++j) {
      poly_t *beta_ = get_beta_at(j);
      const poly_t *alpha_ = get_alpha_at(j);
      beta_->set(context::gauss_struct(&context::fg_prng));
      beta_->forward();
      beta_->add_product_of(*alpha_, sk.sx);
      beta_->negate();  // -e - alpha * sx
      yell::ops::addmod addmod;  // + [sx^2] on limb j only (mult_key.cpp:19-27)
      auto sx2_ptr = sx_square.cptr_at(j);
      auto beta_ptr = beta_->ptr_at(j);
      for (size_t d = 0; d < degree; ++d) addmod.compute(*beta_ptr++, *sx2_ptr++, j);
    }
    sx_square.clear();
  }
  MultKey(MultKey const &) = delete; MultKey(MultKey &&) = delete; MultKey &operator=(MultKey const &) = delete;
  ~MultKey() {}
  const poly_t *get_beta_at(size_t j) const { return &beta.at(j); }
  const poly_t *get_alpha_at(size_t j) const { return &alpha.at(j); }
  poly_t *get_beta_at(size_t j) { return &beta.at(j); }
  poly_t *get_alpha_at(size_t j) { return &alpha.at(j); }
};
}  // namespace fHE

// fHE/keys.hpp — SK / PK as in reference include/fHE/keys.hpp:4-56 (host plumbing; transforms run on the GPU).
#pragma once
#include "fHE/context.hpp"
namespace fHE {
struct SK {
  context::poly_t sx;      // sx over normal moduli
  context::poly_t ext_sx;  // sx over special moduli
  explicit SK(const size_t hwt = context::HWT) : sx(context::nr_ctxt_moduli), ext_sx(context::nr_sp_primes) {
    const size_t L = context::nr_ctxt_moduli, K = context::nr_sp_primes;
    const size_t bytes = sizeof(yell::params::value_type) * context::degree;
    context::poly_t tmp(L + K, yell::hwt_dist(hwt));
    tmp.forward();
    for (size_t cm = 0; cm < L; ++cm) std::memcpy(sx.ptr_at(cm), tmp.cptr_at(cm), bytes);
    for (size_t cm = 0; cm < K; ++cm) std::memcpy(ext_sx.ptr_at(cm), tmp.cptr_at(L + cm), bytes);
    tmp.clear();
  }
  SK(SK const &) = delete; SK(SK &&) = delete; SK &operator=(SK const &) = delete;
  ~SK() {}
};
struct PK {
  context::poly_t bx;
  context::poly_t ax;
  explicit PK(SK const &sk) : bx(context::nr_ctxt_moduli, context::gauss_struct(&context::fg_prng)), ax(context::nr_ctxt_moduli, yell::uniform{}) {
    bx.forward_lazy();  //! bx = -(e + ax * sk.sx)
    bx.add_product_of(ax, sk.sx);
    bx.negate();
  }
  PK(PK const &) = delete; PK(PK &&) = delete; PK &operator=(PK const &) = delete;
  ~PK() {}
};
}  // namespace fHE

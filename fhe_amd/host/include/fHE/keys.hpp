// fHE/keys.hpp — SK / PK with the reference's public shape (include/fHE/keys.hpp:4-56): members, constructors, no copies.
// Host plumbing; the transforms run on the GPU.
#pragma once
#include "fHE/keygen.hpp"
namespace fHE {
struct SK : detail::pinned_object {
  context::poly_t sx;      // residues of the secret under the L normal primes, NTT basis
  context::poly_t ext_sx;  // the same secret under the K special primes, NTT basis

  // sparse ternary secret of Hamming weight hwt: one integer polynomial, reduced under every prime of the table
  explicit SK(const size_t hwt = context::HWT) : sx(context::nr_ctxt_moduli), ext_sx(context::nr_sp_primes) {
    const size_t n = context::degree, L = context::nr_ctxt_moduli, K = context::nr_sp_primes;
    std::vector<signed char> coeff(n, 0);
    for (size_t placed = 0; placed < hwt && placed < n;) {
      const uint64_t r = yell::prng::global().next();
      const size_t at = (size_t)(r % n);
      if (coeff[at]) continue;
      coeff[at] = (r >> 63) ? 1 : -1;
      ++placed;
    }
    for (size_t m = 0; m < L + K; ++m) {
      detail::word *dst = m < L ? sx.ptr_at(m) : ext_sx.ptr_at(m - L);
      const detail::word p = yell::params::P[m];
      for (size_t d = 0; d < n; ++d) dst[d] = coeff[d] > 0 ? 1 : (coeff[d] < 0 ? p - 1 : 0);
    }
    std::fill(coeff.begin(), coeff.end(), 0);
    detail::transform_limbs(sx.ptr_at(0), 0, L, false);
    if (K) detail::transform_limbs(ext_sx.ptr_at(0), L, K, false);
  }
  ~SK() { sx.clear(); ext_sx.clear(); }
  detail::SecretView view() const { return detail::SecretView{&sx, &ext_sx}; }
};

struct PK : detail::pinned_object {
  context::poly_t bx;  // -(e + ax * s)
  context::poly_t ax;  // uniform
  explicit PK(SK const &sk) : bx(context::nr_ctxt_moduli), ax(context::nr_ctxt_moduli, yell::uniform{}) {
    detail::mask_with_secret(bx, ax, sk.view());
  }
};
}  // namespace fHE

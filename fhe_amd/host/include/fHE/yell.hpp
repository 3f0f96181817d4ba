// fHE/yell.hpp — helpers of reference include/fHE/yell.hpp:5-27 / src/yell.cpp on top of the C-ABI.
#pragma once
#include "fHE/context.hpp"
namespace fHE {
inline void apply_galois(context::poly_t *op, size_t galois) {  // src/yell.cpp:4-25
  if (!op) return;
  assert((galois & 1) && galois < 2 * context::degree);
  if (galois == 1 || op->moduli_count() == 0) return;
  yell::check(fhe_hip_apply_galois(yell::runtime::get().ctx, op->ptr_at(0), (uint32_t)op->moduli_count(), galois, 1, FHE_HIP_MEM_HOST), "apply_galois");
}
inline void apply_galois_non_ntt(context::poly_t *op, size_t galois) {  // src/yell.cpp:29-53
  if (!op) return;
  assert((galois & 1) && galois < 2 * context::degree);
  if (galois == 1 || op->moduli_count() == 0) return;
  yell::check(fhe_hip_apply_galois_non_ntt(yell::runtime::get().ctx, op->ptr_at(0), (uint32_t)op->moduli_count(), galois, 1, FHE_HIP_MEM_HOST), "apply_galois_non_ntt");
}
template <class Itr> yell::params::value_type product_of(Itr begin, Itr end, size_t moduli_index) {  // yell.hpp:9-21
  yell::params::value_type product{1};
  yell::ops::mulmod mulmod;
  while (begin != end) { auto i = *begin++; mulmod.compute(product, yell::params::P[i] % yell::params::P[moduli_index], moduli_index); }
  return product;
}
inline void lazy_muladd(yell::params::gt_value_type *rop, const yell::params::value_type *op0, const yell::params::value_type *op1, const size_t n) {
  if (!rop || !op0 || !op1) return;  // src/yell.cpp:56-65
  for (size_t i = 0; i != n; ++i, ++rop) *rop += (yell::params::gt_value_type)(*op0++) * (*op1++);
}
}  // namespace fHE

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
// product of the primes P[i], i in [begin, end), modulo P[moduli_index] (the reference's helper of the same name, include/fHE/yell.hpp)
template <class Itr> yell::params::value_type product_of(Itr begin, Itr end, size_t moduli_index) {
  const yell::params::value_type p = yell::params::P[moduli_index];
  yell::params::gt_value_type acc = 1;
  for (Itr it = begin; it != end; ++it) acc = acc * (yell::params::P[*it] % p) % p;
  return (yell::params::value_type)acc;
}
// rop[i] += op0[i] * op1[i] without reduction: the host-side form of the accumulation the device fuses into its key MAC kernels
inline void lazy_muladd(yell::params::gt_value_type *rop, const yell::params::value_type *op0, const yell::params::value_type *op1, const size_t n) {
  if (!rop || !op0 || !op1) return;
  for (size_t i = 0; i < n; ++i) rop[i] += (yell::params::gt_value_type)op0[i] * op1[i];
}
}  // namespace fHE

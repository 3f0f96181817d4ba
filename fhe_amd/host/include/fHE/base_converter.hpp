// fHE/base_converter.hpp — public interface of reference include/fHE/base_converter.hpp:5-37; the Impl forwards to
// the C-ABI (tables live in the device context, built as in src/base_converter.cpp:122-187).
#pragma once
#include "fHE/context.hpp"
namespace fHE {
struct BaseConverter {
  using T = yell::params::value_type;
  BaseConverter() {}
  ~BaseConverter() {}

  //! adding special primes (src/base_converter.cpp:189-203); throws std::invalid_argument for <= 1 moduli (:284-285)
  bool approximated_mod_up(context::poly_t *rop, context::poly_t const &op) const {
    if (!rop) return false;
    const size_t n = op.moduli_count();
    if (n <= 1) throw std::invalid_argument("nmoduli <= 1");
    if (rop->moduli_count() != n + context::nr_sp_primes) rop->resize_moduli_count(n + context::nr_sp_primes);
    return fhe_hip_mod_up(yell::runtime::get().ctx, rop->ptr_at(0), op.cptr_at(0), (uint32_t)n, 1, FHE_HIP_MEM_HOST) == FHE_HIP_OK;
  }
  // special primes part in the power basis, normal primes part in the NTT basis (:205-242)
  bool approximated_mod_down(context::poly_t *rop, context::poly_t const &op) const {
    if (!rop) return false;
    const size_t L0 = rop->moduli_count();
    if (L0 + context::nr_sp_primes != op.moduli_count()) return false;
    return fhe_hip_mod_down(yell::runtime::get().ctx, rop->ptr_at(0), op.cptr_at(0), (uint32_t)L0, 1, FHE_HIP_MEM_HOST) == FHE_HIP_OK;
  }
  // last moduli of [op] in the power basis, the others in the NTT basis (:244-274)
  bool rescale(context::poly_t *rop, context::poly_t const &op) const {
    if (!rop) return false;
    const size_t n = op.moduli_count();
    if (n < 2 || rop->moduli_count() + 1 != n) return false;
    return fhe_hip_rescale(yell::runtime::get().ctx, rop->ptr_at(0), op.cptr_at(0), (uint32_t)n, 1, FHE_HIP_MEM_HOST) == FHE_HIP_OK;
  }
  // single-target conversions (:276-311), expressed through the batched entry points
  void approx_convert_to_special_basis(T *rop, context::poly_t const &op, const size_t sp_moduli_index) const {
    if (!rop) return;
    const size_t n = op.moduli_count();
    if (n <= 1) throw std::invalid_argument("nmoduli <= 1");   // src/base_converter.cpp:284-285
    yell::check(fhe_hip_conv_to_special(yell::runtime::get().ctx, rop, op.cptr_at(0), (uint32_t)n, (uint32_t)sp_moduli_index, 1, FHE_HIP_MEM_HOST), "conv_to_special");
  }
  // op = the K special limbs in the power basis -> -conv_{P -> q_j}(op) for one normal modulus j (:297-311)
  void neg_approx_convert_to_normal_basis(T *rop, context::poly_t const &op, const size_t nrl_moduli_index) const {
    if (!rop) return;
    if (op.moduli_count() != context::nr_sp_primes || nrl_moduli_index >= context::nr_ctxt_moduli) throw std::invalid_argument("neg_approx_convert_to_normal_basis: shape");
    context::poly_t conv(nrl_moduli_index + 1);
    yell::check(fhe_hip_conv_to_normal(yell::runtime::get().ctx, conv.ptr_at(0), op.cptr_at(0), (uint32_t)(nrl_moduli_index + 1), 1, FHE_HIP_MEM_HOST), "conv_to_normal");
    std::memcpy(rop, conv.cptr_at(nrl_moduli_index), sizeof(T) * context::degree);
  }
};
}  // namespace fHE

// fHE/rot_key.hpp — RotKey / RotKeySet of reference include/fHE/rot_key.hpp:9-39,84-118 + src/rot_key.cpp:6-79.
// (MixedRotKey is a SURVEY §8(f) "next" row.)
#pragma once
#include "fHE/context.hpp"
#include "fHE/keys.hpp"
#include "fHE/yell.hpp"
#include <unordered_map>
namespace fHE {
inline size_t rotation_offset_galois(int offset) {  // rot_key.cpp:6-21
  const size_t nr_slots = context::degree >> 1u, m = context::degree << 1u;
  bool sign = offset > 0;
  offset = std::abs(offset) % (int)nr_slots;
  if (sign) offset = (int)nr_slots - offset;
  size_t galois{1};
  while (offset--) { galois *= 3; galois &= (m - 1); }
  return galois;
}
struct RotKey {
 private:
  void lift_normal_moduli(context::poly_t *op) const {  // [x] -> [P * x] on the normal moduli (rot_key.cpp:64-79)
    const size_t L = context::nr_ctxt_moduli, K = context::nr_sp_primes, degree = context::degree;
    if (!op) return;
    yell::ops::mulmod mulmod;
    std::vector<size_t> sp_primes(K);
    std::iota(sp_primes.begin(), sp_primes.end(), context::index_sp_prime(0));
    for (size_t cm = 0; cm < L; ++cm) {
      auto P_mod_qj = product_of(sp_primes.begin(), sp_primes.end(), cm);
      auto dst = op->ptr_at(cm);
      for (size_t d = 0; d < degree; ++d) mulmod.compute(*dst++, P_mod_qj, cm);
    }
  }

 public:
  enum class Direction { LEFT, RIGHT };
  const size_t galois;
  context::poly_t beta;   //! L + K moduli
  context::poly_t alpha;  //! L + K moduli

  explicit RotKey(SK const &sk, uint16_t offset, Direction dir)
      : galois(rotation_offset_galois(dir == Direction::RIGHT ? offset : -(int)offset)),
        beta(context::nr_ctxt_moduli + context::nr_sp_primes),
        alpha(context::nr_ctxt_moduli + context::nr_sp_primes, yell::uniform{}) {
    using T = yell::params::value_type; using gT = yell::params::gt_value_type;
    const size_t L = context::nr_ctxt_moduli, K = context::nr_sp_primes, degree = context::degree;
    auto rotated_sx(sk.sx);
    apply_galois(&rotated_sx, galois);
    lift_normal_moduli(&rotated_sx);
    beta.set(context::gauss_struct(&context::fg_prng));  // reuse beta as the 'e'
    beta.forward_lazy();
    for (size_t cm = 0; cm < L + K; ++cm) {
      auto bptr = beta.ptr_at(cm);
      auto aptr = alpha.cptr_at(cm);
      auto sptr = cm < L ? sk.sx.cptr_at(cm) : sk.ext_sx.cptr_at(cm - L);
      for (size_t d = 0; d < degree; ++d) {
        gT tmp = (gT)(*aptr++) * (*sptr++) + *bptr;
        yell::ops::barret_reduction(&tmp, cm);
        *bptr++ = (T)tmp;
      }
    }
    beta.negate();
    yell::ops::addmod addmod;  // + P * s'
    for (size_t cm = 0; cm < L; ++cm) {
      auto op0 = beta.ptr_at(cm);
      auto op1 = rotated_sx.cptr_at(cm);
      for (size_t d = 0; d < degree; ++d) addmod.compute(*op0++, *op1++, cm);
    }
    rotated_sx.clear();
  }
  RotKey(RotKey const &) = delete; RotKey(RotKey &&) = delete; RotKey &operator=(RotKey const &) = delete;
  ~RotKey() {}
};

template <class RotKeyType> struct RotKeySet {
 private:
  using rotkey_ptr = std::shared_ptr<RotKeyType>;
  std::unordered_map<size_t, rotkey_ptr> rot_keys;
  size_t logn;

 public:
  //! log(N/2) left-rotation keys for offsets 1, 2, 4, ... (rot_key.hpp:94-100)
  explicit RotKeySet(SK const &sk) : logn(yell::runtime::get().logn - 1) {
    for (size_t i = 0; i < logn; ++i) {
      int idx = 1 << i;
      rot_keys.insert({(size_t)idx, std::make_shared<RotKeyType>(sk, (uint16_t)idx, RotKeyType::Direction::LEFT)});
    }
  }
  RotKeySet(RotKeySet const &) = delete; RotKeySet(RotKeySet &&) = delete; RotKeySet &operator=(RotKeySet const &) = delete;
  const RotKeyType *get(uint16_t offset) const {
    if (offset >= (1LL << logn)) return nullptr;
    auto kv = rot_keys.find(offset);
    return kv == rot_keys.end() ? nullptr : kv->second.get();
  }
  ~RotKeySet() {}
};
}  // namespace fHE

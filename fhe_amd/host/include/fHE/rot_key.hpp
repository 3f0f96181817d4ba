// fHE/rot_key.hpp — RotKey / RotKeySet of reference include/fHE/rot_key.hpp:9-39,84-118 + src/rot_key.cpp:6-79.
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

// MixedRotKey (reference include/fHE/rot_key.hpp:41-82, src/rot_key.cpp:81-120): digit-decomposed rotation key with
// ONE special prime; L polynomials of L + 1 moduli each.
struct MixedRotKey {
 private:
  using poly_t = context::poly_t;
  std::vector<poly_t> alpha;  // L polynomials, each L + 1 moduli
  std::vector<poly_t> beta;

 public:
  enum class Direction { LEFT, RIGHT };
  const size_t galois;

  explicit MixedRotKey(SK const &sk, uint16_t offset, Direction dir)
      : galois(rotation_offset_galois(dir == Direction::RIGHT ? offset : -(int)offset)) {
    using T = yell::params::value_type;
    const size_t L = context::nr_ctxt_moduli, degree = context::degree, bytes = degree * sizeof(T);
    for (size_t j = 0; j < L; ++j) { beta.emplace_back(L + 1); alpha.emplace_back(L + 1, yell::uniform{}); }
    context::poly_t ext_sx(L + 1);
    for (size_t j = 0; j < L + 1; ++j) std::memcpy(ext_sx.ptr_at(j), j < L ? sk.sx.cptr_at(j) : sk.ext_sx.cptr_at(0), bytes);
    auto rotated_sx(ext_sx);
    apply_galois(&rotated_sx, galois);
    // beta_j = -sx * alpha_j + [qK]_qj * rotated(sx) + e   (rot_key.cpp:99-117)
    yell::ops::muladd muladd;
    const T qK = yell::params::P[context::index_sp_prime(0)];
    for (size_t j = 0; j < L; ++j) {
      poly_t *beta_ = get_beta_at(j);
      const poly_t *alpha_ = get_alpha_at(j);
      beta_->set(context::gauss_struct(&context::fg_prng));
      beta_->forward();
      beta_->add_product_of(*alpha_, ext_sx);
      beta_->negate();
      const T qK_j = qK % yell::params::P[j];
      auto b = beta_->ptr_at(j);
      auto s = rotated_sx.cptr_at(j);
      for (size_t d = 0; d < degree; ++d) b[d] = muladd(b[d], s[d], qK_j, j);
    }
    rotated_sx.clear(); ext_sx.clear();
  }
  MixedRotKey(MixedRotKey const &) = delete; MixedRotKey(MixedRotKey &&) = delete; MixedRotKey &operator=(MixedRotKey const &) = delete;
  const poly_t *get_beta_at(size_t j) const { return &beta.at(j); }
  const poly_t *get_alpha_at(size_t j) const { return &alpha.at(j); }
  poly_t *get_beta_at(size_t j) { return &beta.at(j); }
  poly_t *get_alpha_at(size_t j) { return &alpha.at(j); }
  ~MixedRotKey() {}
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

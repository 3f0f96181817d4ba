// fHE/evaluator.hpp — public interface of reference include/fHE/evaluator.hpp:12-40.  The reference's pimpl
// (src/evaluator.cpp:12-68) is replaced by calls into the gfx950 C-ABI; error conventions are kept: in place on
// rop, false + one std::cerr line on misuse (evaluator.cpp:130-134, 275-283, 581-584).
#pragma once
#include "fHE/base_converter.hpp"
#include "fHE/cipher.hpp"
#include "fHE/encoder.hpp"
#include "fHE/mult_key.hpp"
#include "fHE/rot_key.hpp"
namespace fHE {
struct Evaluator {
 private:
  std::shared_ptr<Encoder> encoder;
  using T = yell::params::value_type;

  static fhe_hip_ctx *ctx() { return yell::runtime::get().ctx; }
  static void pack(std::vector<T> &buf, size_t off, context::poly_t const &p, size_t n) { std::memcpy(buf.data() + off, p.cptr_at(0), n * context::degree * sizeof(T)); }
  // Which key a device context holds is recorded with the context (yell::runtime), under the key object's process-unique
  // id: Evaluators that share a context, keys that alternate, and a key built at the address of a destroyed one are all
  // told apart; whenever the record does not prove the key is there, it is uploaded again.
  static bool ensure_mult_key(MultKey const &mkey) {
    auto &held = yell::runtime::get().keys();
    if (held.mult == mkey.id) return true;
    const size_t L = context::nr_ctxt_moduli, N = context::degree;
    std::vector<T> beta(L * L * N), alpha(L * L * N);
    for (size_t j = 0; j < L; ++j) { pack(beta, j * L * N, *mkey.get_beta_at(j), L); pack(alpha, j * L * N, *mkey.get_alpha_at(j), L); }
    held.mult = 0;
    if (fhe_hip_load_mult_key(ctx(), beta.data(), alpha.data(), FHE_HIP_MEM_HOST) != FHE_HIP_OK) return false;
    held.mult = mkey.id;
    return true;
  }
  static bool ensure_rot_key(RotKey const &rkey) {
    auto &held = yell::runtime::get().keys();
    auto it = held.rot.find(rkey.galois);
    if (it != held.rot.end() && it->second == rkey.id) return true;
    held.rot.erase(rkey.galois);
    if (fhe_hip_load_rot_key(ctx(), rkey.galois, rkey.beta.cptr_at(0), rkey.alpha.cptr_at(0), FHE_HIP_MEM_HOST) != FHE_HIP_OK) return false;
    held.rot[rkey.galois] = rkey.id;
    return true;
  }
  static bool ensure_mixed_key(MixedRotKey const &rkey) {
    auto &held = yell::runtime::get().keys();
    auto it = held.mixed.find(rkey.galois);
    if (it != held.mixed.end() && it->second == rkey.id) return true;
    held.mixed.erase(rkey.galois);
    const size_t L = context::nr_ctxt_moduli, N = context::degree;
    std::vector<T> beta(L * (L + 1) * N), alpha(L * (L + 1) * N);
    for (size_t j = 0; j < L; ++j) { pack(beta, j * (L + 1) * N, *rkey.get_beta_at(j), L + 1); pack(alpha, j * (L + 1) * N, *rkey.get_alpha_at(j), L + 1); }
    if (fhe_hip_load_mixed_rot_key(ctx(), rkey.galois, beta.data(), alpha.data(), FHE_HIP_MEM_HOST) != FHE_HIP_OK) return false;
    held.mixed[rkey.galois] = rkey.id;
    return true;
  }
  static bool canonical_or_complain(Cipher const *c, const char *what) {
    if (c->canonical()) return true;
    std::cerr << "Can only " << what << " canonical ciphertext, call relinearize first." << std::endl;
    return false;
  }
  // one key switch per set bit of `left` (a left rotation by `left` slots), lowest bit first, with the power-of-two keys of the set
  template <class RotKeyT, class Apply> static bool walk_set_bits(size_t left, RotKeySet<RotKeyT> const &rkeys, Apply const &apply) {
    while (left) {
      const size_t step = left & (~left + 1);  // lowest set bit
      RotKeyT const *key = rkeys.get((uint16_t)step);
      if (!key) { std::cerr << "No such rotation key for offset " << step << "\n"; return false; }
      if (!apply(*key)) return false;
      left ^= step;
    }
    return true;
  }

  bool rotate_with(Cipher *rop, RotKey const &rkey, bool non_ntt) const {
    if (!rop || !canonical_or_complain(rop, "rotate")) return false;
    if (rkey.galois == 0) return true;
    const size_t Li = rop->moduli_count(), N = context::degree, limbs = Li * N;
    if (Li > context::nr_ctxt_moduli) throw std::invalid_argument("Too many moduli.");
    if (Li <= 1) throw std::invalid_argument("nmoduli <= 1");
    if (!ensure_rot_key(rkey)) return false;
    std::vector<T> ct(2 * limbs);
    pack(ct, 0, rop->bx, Li); pack(ct, limbs, rop->ax, Li);
    const int rc = non_ntt ? fhe_hip_rotate_non_ntt(ctx(), rkey.galois, ct.data(), (uint32_t)Li, 1, FHE_HIP_MEM_HOST)
                           : fhe_hip_rotate(ctx(), rkey.galois, ct.data(), (uint32_t)Li, 1, FHE_HIP_MEM_HOST);
    if (rc != FHE_HIP_OK) return false;
    std::memcpy(rop->bx.ptr_at(0), ct.data(), limbs * sizeof(T));
    std::memcpy(rop->ax.ptr_at(0), ct.data() + limbs, limbs * sizeof(T));
    return true;
  }

 public:
  explicit Evaluator(std::shared_ptr<Encoder> encoder = nullptr) : encoder(encoder) {}
  ~Evaluator() {}

  bool add(Cipher *rop, Cipher const &c1) const {  // Impl::add_inplace, evaluator.cpp:270-297
    if (!rop) return false;
    if (rop->moduli_count() != c1.moduli_count()) { std::cerr << "Can not add_inplace." << std::endl; return false; }
    if (!rop->canonical() && c1.canonical()) { std::cerr << "Can not add_inplace" << std::endl; return false; }
    const uint32_t n = (uint32_t)rop->moduli_count();
    bool ok = fhe_hip_add(ctx(), rop->bx.ptr_at(0), c1.bx.cptr_at(0), n, 1, FHE_HIP_MEM_HOST) == FHE_HIP_OK &&
              fhe_hip_add(ctx(), rop->ax.ptr_at(0), c1.ax.cptr_at(0), n, 1, FHE_HIP_MEM_HOST) == FHE_HIP_OK;
    if (!c1.canonical()) {
      if (rop->canonical()) rop->mul_aux = std::make_shared<context::poly_t>(*c1.mul_aux);
      else ok = ok && fhe_hip_add(ctx(), rop->mul_aux->ptr_at(0), c1.mul_aux->cptr_at(0), n, 1, FHE_HIP_MEM_HOST) == FHE_HIP_OK;
    }
    if (rop->scale() == 1.) rop->scale(c1.scale());
    return ok;
  }

  bool multiply(Cipher *rop, Cipher const &c1) const {  // Impl::multiply_inplace, evaluator.cpp:576-597
    if (!rop) return false;
    if (rop->moduli_count() != c1.moduli_count() || !(rop->canonical() && c1.canonical())) { std::cerr << "Can not multiply_inplace" << std::endl; return false; }
    const size_t Li = rop->moduli_count(), N = context::degree, limbs = Li * N;
    const double new_scale = rop->scale() / yell::params::P[Li - 1] * c1.scale();
    std::vector<T> in0(2 * limbs), in1(2 * limbs), out(3 * limbs);
    pack(in0, 0, rop->bx, Li); pack(in0, limbs, rop->ax, Li); pack(in1, 0, c1.bx, Li); pack(in1, limbs, c1.ax, Li);
    if (fhe_hip_cross_mult(ctx(), out.data(), in0.data(), in1.data(), (uint32_t)Li, 1, FHE_HIP_MEM_HOST) != FHE_HIP_OK) return false;
    std::memcpy(rop->bx.ptr_at(0), out.data(), limbs * sizeof(T));
    std::memcpy(rop->ax.ptr_at(0), out.data() + limbs, limbs * sizeof(T));
    rop->mul_aux = std::make_shared<context::poly_t>(Li);
    std::memcpy(rop->mul_aux->ptr_at(0), out.data() + 2 * limbs, limbs * sizeof(T));
    rop->scale(new_scale);
    return true;
  }

  bool multiply_plain(Cipher *rop, context::poly_t const &p, const double plain_scale = -1) const {  // evaluator.cpp:88-111
    if (!rop) return false;
    const size_t nm = p.moduli_count(), N = context::degree;
    if (rop->moduli_count() != nm) return false;
    std::vector<T> ct(2 * nm * N);
    pack(ct, 0, rop->bx, nm); pack(ct, nm * N, rop->ax, nm);
    if (fhe_hip_multiply_plain(ctx(), ct.data(), p.cptr_at(0), (uint32_t)nm, 1, FHE_HIP_MEM_HOST) != FHE_HIP_OK) return false;
    std::memcpy(rop->bx.ptr_at(0), ct.data(), nm * N * sizeof(T));
    std::memcpy(rop->ax.ptr_at(0), ct.data() + nm * N, nm * N * sizeof(T));
    if (plain_scale > 0.) rop->scale(rop->scale() * plain_scale);
    return true;
  }

  bool relinearize(Cipher *rop, MultKey const &mkey) const {  // Impl::relinerize_inplace (relin + rescale), evaluator.cpp:599-619
    if (!rop) return false;
    if (rop->canonical()) return false;
    const size_t Li = rop->moduli_count(), N = context::degree, limbs = Li * N;
    if (Li < 2 || !ensure_mult_key(mkey)) return false;
    std::vector<T> in(3 * limbs), out(2 * (Li - 1) * N);
    pack(in, 0, rop->bx, Li); pack(in, limbs, rop->ax, Li); pack(in, 2 * limbs, *rop->mul_aux, Li);
    if (fhe_hip_relinearize(ctx(), out.data(), in.data(), (uint32_t)Li, 1, FHE_HIP_MEM_HOST) != FHE_HIP_OK) return false;
    const double s = rop->scale();
    Cipher tmp(Li - 1);
    std::memcpy(tmp.bx.ptr_at(0), out.data(), (Li - 1) * N * sizeof(T));
    std::memcpy(tmp.ax.ptr_at(0), out.data() + (Li - 1) * N, (Li - 1) * N * sizeof(T));
    rop->bx = std::move(tmp.bx); rop->ax = std::move(tmp.ax); rop->mul_aux = nullptr;
    rop->scale(s);
    return true;
  }

  bool multiply_and_relin(Cipher *rop, Cipher const &c1, MultKey const &mkey) const {  // evaluator.cpp:119-126, one fused device call
    if (!rop) return false;
    if (rop->moduli_count() != c1.moduli_count() || !(rop->canonical() && c1.canonical())) { std::cerr << "Can not multiply_inplace" << std::endl; return false; }
    const size_t Li = rop->moduli_count(), N = context::degree, limbs = Li * N;
    if (Li < 2 || !ensure_mult_key(mkey)) return false;
    const double new_scale = rop->scale() / yell::params::P[Li - 1] * c1.scale();
    std::vector<T> in0(2 * limbs), in1(2 * limbs), out(2 * (Li - 1) * N);
    pack(in0, 0, rop->bx, Li); pack(in0, limbs, rop->ax, Li); pack(in1, 0, c1.bx, Li); pack(in1, limbs, c1.ax, Li);
    if (fhe_hip_mul_relin_rescale(ctx(), out.data(), in0.data(), in1.data(), (uint32_t)Li, 1, FHE_HIP_MEM_HOST) != FHE_HIP_OK) return false;
    Cipher tmp(Li - 1);
    std::memcpy(tmp.bx.ptr_at(0), out.data(), (Li - 1) * N * sizeof(T));
    std::memcpy(tmp.ax.ptr_at(0), out.data() + (Li - 1) * N, (Li - 1) * N * sizeof(T));
    rop->bx = std::move(tmp.bx); rop->ax = std::move(tmp.ax); rop->mul_aux = nullptr;
    rop->scale(new_scale);
    return true;
  }

  //! The rotation offset is specified in the RotKey (evaluator.cpp:128-136, 299-311, 343-399).
  bool rotate_slots(Cipher *rop, RotKey const &rkey) const { return rotate_with(rop, rkey, false); }
  //! Same on a power-basis ciphertext (evaluator.cpp:138-146, 327-339, 520-574); the device call reproduces the
  //! reference's arithmetic for this path literally (see include/fhe_hip.h).
  bool rotate_slots_non_ntt(Cipher *rop, RotKey const &rkey) const { return rotate_with(rop, rkey, true); }

  //! MixedRotKey variant (evaluator.cpp:148-156, 313-325, 403-516)
  bool rotate_slots(Cipher *rop, MixedRotKey const &rkey) const {
    if (!rop || !canonical_or_complain(rop, "rotate")) return false;
    if (rkey.galois == 0) return true;
    const size_t Li = rop->moduli_count(), N = context::degree, limbs = Li * N;
    if (!ensure_mixed_key(rkey)) return false;
    std::vector<T> ct(2 * limbs);
    pack(ct, 0, rop->bx, Li); pack(ct, limbs, rop->ax, Li);
    if (fhe_hip_rotate_mixed(ctx(), rkey.galois, ct.data(), (uint32_t)Li, 1, FHE_HIP_MEM_HOST) != FHE_HIP_OK) return false;
    std::memcpy(rop->bx.ptr_at(0), ct.data(), limbs * sizeof(T));
    std::memcpy(rop->ax.ptr_at(0), ct.data() + limbs, limbs * sizeof(T));
    return true;
  }

  //! Rotation by an arbitrary offset (reference src/evaluator.cpp:158-208): offset > 0 rotates left by offset slots, offset < 0
  //! right by |offset|, i.e. left by offset mod N/2; one key switch per set bit of that number with the set's power-of-two keys.
  template <class RotKeyT> bool rotate_slots(Cipher *rop, int offset, RotKeySet<RotKeyT> const &rkeys) const {
    if (!rop || !canonical_or_complain(rop, "rotate")) return false;
    const long slots = (long)(context::degree >> 1);
    const size_t left = (size_t)((((long)offset % slots) + slots) % slots);
    return walk_set_bits(left, rkeys, [&](RotKeyT const &key) { return rotate_slots(rop, key); });
  }
  //! Power-basis variant (reference src/evaluator.cpp:210-232); there the sign of the offset is ignored: |offset| mod N/2.
  template <class RotKeyT> bool rotate_slots_non_ntt(Cipher *rop, int offset, RotKeySet<RotKeyT> const &rkeys) const {
    if (!rop || !canonical_or_complain(rop, "rotate")) return false;
    const size_t left = (size_t)(offset < 0 ? -(long)offset : (long)offset) % (context::degree >> 1);
    return walk_set_bits(left, rkeys, [&](RotKeyT const &key) { return rotate_slots_non_ntt(rop, key); });
  }

  //! Replicate the slot at [pos] into every slot (reference src/evaluator.cpp:234-267): keep that slot alone by a plaintext
  //! mask, then double the number of copies log2(N/2) times, ct += rot(ct, 2^i), all in the power basis.
  template <class RotKeyT> bool replicate(Cipher *rop, size_t pos, RotKeySet<RotKeyT> const &rkeys) const {
    if (!rop || !encoder || !canonical_or_complain(rop, "replicate")) return false;
    const size_t slots = context::degree >> 1;
    const double mask_scale = context::encoder_scale / 8.;
    std::vector<double> one_hot(slots, 0.);
    one_hot[pos % slots] = 1.;
    context::poly_t mask(rop->moduli_count());
    encoder->encode(&mask, one_hot, mask_scale);
    mask.forward_lazy();
    if (!multiply_plain(rop, mask, mask_scale)) return false;
    rop->bx.backward();
    rop->ax.backward();
    for (size_t span = 1; span < slots; span <<= 1) {
      Cipher shifted(*rop);
      if (!rotate_slots_non_ntt(&shifted, (int)span, rkeys) || !add(rop, shifted)) return false;  // limb-wise add: basis-agnostic
    }
    rop->bx.forward();
    rop->ax.forward();
    return true;
  }
};
}  // namespace fHE

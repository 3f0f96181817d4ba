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

  using block = yell::dev_block;
  static yell::dev_block_ptr new_block(size_t words) { return std::make_shared<block>(ctx(), words); }

  // B ciphertexts of Li limbs as ONE device operand [ct][half][limb][N].  Ciphertexts that an earlier batched call left as
  // consecutive views of one block are used where they lie; otherwise they are brought together (device-to-device, or
  // uploaded if they live on the host) and stay views of the new block.
  // The block is returned with the pointer: the caller holds it until its device call is enqueued (gathering a second
  // operand may move polynomials out of the first one's block, which must not be freed under the pointer).
  template <class CipherPtr> static std::pair<const T *, yell::dev_block_ptr> device_batch(std::vector<CipherPtr> const &cts, size_t Li) {
    const size_t w = 2 * Li * context::degree;
    const T *base = cts[0]->device_parts();
    bool together = true;
    for (size_t b = 1; b < cts.size() && together; ++b) {
      Cipher const &c = *cts[b];
      together = c.bx.on_device() && c.ax.on_device() && c.bx.block() == cts[0]->bx.block() && c.bx.block_offset() == cts[0]->bx.block_offset() + b * w &&
                 c.ax.block() == c.bx.block() && c.ax.block_offset() == c.bx.block_offset() + w / 2;
    }
    if (together) return {base, cts[0]->bx.block()};
    auto blk = new_block(cts.size() * w);
    for (size_t b = 0; b < cts.size(); ++b) { cts[b]->bx.relocate(blk, b * w); cts[b]->ax.relocate(blk, b * w + w / 2); }
    return {blk->p, blk};
  }

  bool rotate_with(Cipher *rop, RotKey const &rkey, bool non_ntt) const {
    if (!rop || !canonical_or_complain(rop, "rotate")) return false;
    if (rkey.galois == 0) return true;
    const size_t Li = rop->moduli_count(), N = context::degree;
    if (Li > context::nr_ctxt_moduli) throw std::invalid_argument("Too many moduli.");
    if (Li <= 1) throw std::invalid_argument("nmoduli <= 1");
    if (!ensure_rot_key(rkey)) return false;
    if (non_ntt) return fhe_hip_rotate_non_ntt(ctx(), rkey.galois, rop->device_parts_mut(), (uint32_t)Li, 1, FHE_HIP_MEM_DEVICE) == FHE_HIP_OK;
    auto out = new_block(2 * Li * N);   // out of place on the device: no copy back over the input
    if (fhe_hip_rotate_to(ctx(), rkey.galois, out->p, rop->device_parts(), (uint32_t)Li, 1, FHE_HIP_MEM_DEVICE) != FHE_HIP_OK) return false;
    rop->adopt_device(out, 0, Li);
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
    bool ok = fhe_hip_add(ctx(), rop->bx.dev(), c1.bx.cdev(), n, 1, FHE_HIP_MEM_DEVICE) == FHE_HIP_OK &&
              fhe_hip_add(ctx(), rop->ax.dev(), c1.ax.cdev(), n, 1, FHE_HIP_MEM_DEVICE) == FHE_HIP_OK;
    if (!c1.canonical()) {
      if (rop->canonical()) rop->mul_aux = std::make_shared<context::poly_t>(*c1.mul_aux);
      else ok = ok && fhe_hip_add(ctx(), rop->mul_aux->dev(), c1.mul_aux->cdev(), n, 1, FHE_HIP_MEM_DEVICE) == FHE_HIP_OK;
    }
    if (rop->scale() == 1.) rop->scale(c1.scale());
    return ok;
  }

  bool multiply(Cipher *rop, Cipher const &c1) const {  // Impl::multiply_inplace, evaluator.cpp:576-597
    if (!rop) return false;
    if (rop->moduli_count() != c1.moduli_count() || !(rop->canonical() && c1.canonical())) { std::cerr << "Can not multiply_inplace" << std::endl; return false; }
    const size_t Li = rop->moduli_count(), N = context::degree, limbs = Li * N;
    const double new_scale = rop->scale() / yell::params::P[Li - 1] * c1.scale();
    auto out = new_block(3 * limbs);
    if (fhe_hip_cross_mult(ctx(), out->p, rop->device_parts(), c1.device_parts(), (uint32_t)Li, 1, FHE_HIP_MEM_DEVICE) != FHE_HIP_OK) return false;
    rop->adopt_device(out, 0, Li);
    rop->mul_aux = std::make_shared<context::poly_t>(0);
    rop->mul_aux->adopt(out, 2 * limbs, Li);   // (bx, ax, mul_aux) stay one [3][limb][N] operand for relinearize
    rop->scale(new_scale);
    return true;
  }

  bool multiply_plain(Cipher *rop, context::poly_t const &p, const double plain_scale = -1) const {  // evaluator.cpp:88-111
    if (!rop) return false;
    const size_t nm = p.moduli_count();
    if (rop->moduli_count() != nm) return false;
    if (fhe_hip_multiply_plain(ctx(), rop->device_parts_mut(), p.cdev(), (uint32_t)nm, 1, FHE_HIP_MEM_DEVICE) != FHE_HIP_OK) return false;
    if (plain_scale > 0.) rop->scale(rop->scale() * plain_scale);
    return true;
  }

  bool relinearize(Cipher *rop, MultKey const &mkey) const {  // Impl::relinerize_inplace (relin + rescale), evaluator.cpp:599-619
    if (!rop) return false;
    if (rop->canonical()) return false;
    const size_t Li = rop->moduli_count(), N = context::degree;
    if (Li < 2 || !ensure_mult_key(mkey)) return false;
    auto out = new_block(2 * (Li - 1) * N);
    if (fhe_hip_relinearize(ctx(), out->p, rop->device_parts(true), (uint32_t)Li, 1, FHE_HIP_MEM_DEVICE) != FHE_HIP_OK) return false;
    rop->adopt_device(out, 0, Li - 1);
    return true;
  }

  bool multiply_and_relin(Cipher *rop, Cipher const &c1, MultKey const &mkey) const {  // evaluator.cpp:119-126, one fused device call
    std::vector<Cipher *> rops{rop};
    std::vector<Cipher const *> c1s{&c1};
    return rop && multiply_and_relin(rops, c1s, mkey);
  }

  //! Batched form (no counterpart in the reference, which handles one ciphertext per call): rops[b] <- rops[b] * c1s[b],
  //! relinearised and rescaled, all in ONE device call over [ct][half][limb][N]; the results are views of one block, so
  //! a following batched call on the same ciphertexts finds its operand already in place.
  bool multiply_and_relin(std::vector<Cipher *> const &rops, std::vector<Cipher const *> const &c1s, MultKey const &mkey) const {
    if (rops.empty() || rops.size() != c1s.size()) return false;
    const size_t B = rops.size(), Li = rops[0] ? rops[0]->moduli_count() : 0, N = context::degree;
    for (size_t b = 0; b < B; ++b) {
      if (!rops[b] || !c1s[b]) return false;
      if (rops[b]->moduli_count() != Li || c1s[b]->moduli_count() != Li || !(rops[b]->canonical() && c1s[b]->canonical())) { std::cerr << "Can not multiply_inplace" << std::endl; return false; }
    }
    if (Li < 2 || !ensure_mult_key(mkey)) return false;
    const auto in0 = device_batch(rops, Li), in1 = device_batch(c1s, Li);
    auto out = new_block(B * 2 * (Li - 1) * N);
    if (fhe_hip_mul_relin_rescale(ctx(), out->p, in0.first, in1.first, (uint32_t)Li, B, FHE_HIP_MEM_DEVICE) != FHE_HIP_OK) return false;
    for (size_t b = 0; b < B; ++b) {
      const double new_scale = rops[b]->scale() / yell::params::P[Li - 1] * c1s[b]->scale();
      rops[b]->adopt_device(out, b * 2 * (Li - 1) * N, Li - 1);
      rops[b]->scale(new_scale);
    }
    return true;
  }

  //! Batched rotation with one key (same remark): rops[b] <- rotate(rops[b]) in ONE device call.
  bool rotate_slots(std::vector<Cipher *> const &rops, RotKey const &rkey) const {
    if (rops.empty()) return false;
    const size_t B = rops.size(), Li = rops[0] ? rops[0]->moduli_count() : 0, N = context::degree;
    for (size_t b = 0; b < B; ++b) if (!rops[b] || rops[b]->moduli_count() != Li || !canonical_or_complain(rops[b], "rotate")) return false;
    if (rkey.galois == 0) return true;
    if (Li > context::nr_ctxt_moduli) throw std::invalid_argument("Too many moduli.");
    if (Li <= 1) throw std::invalid_argument("nmoduli <= 1");
    if (!ensure_rot_key(rkey)) return false;
    const auto in = device_batch(rops, Li);
    auto out = new_block(B * 2 * Li * N);
    if (fhe_hip_rotate_to(ctx(), rkey.galois, out->p, in.first, (uint32_t)Li, B, FHE_HIP_MEM_DEVICE) != FHE_HIP_OK) return false;
    for (size_t b = 0; b < B; ++b) rops[b]->adopt_device(out, b * 2 * Li * N, Li);
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
    const size_t Li = rop->moduli_count();
    if (!ensure_mixed_key(rkey)) return false;
    return fhe_hip_rotate_mixed(ctx(), rkey.galois, rop->device_parts_mut(), (uint32_t)Li, 1, FHE_HIP_MEM_DEVICE) == FHE_HIP_OK;
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

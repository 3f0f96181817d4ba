// fHE/cipher.hpp — the ciphertext value type, same members and meaning as the reference's (include/fHE/cipher.hpp):
// bx / ax are the two halves, mul_aux the degree-2 part a multiplication leaves until relinearisation, scale the CKKS scale.
// The polynomials keep a device copy with lazy host synchronisation (yell/poly.hpp): the evaluator reads and writes
// ciphertexts where they lie on the GPU, and bx.ptr_at() / cptr_at() fault them back to the host when somebody looks.
#pragma once
#include "fHE/context.hpp"
#include <memory>
namespace fHE {
struct Cipher {
  context::poly_t bx, ax;
  std::shared_ptr<context::poly_t> mul_aux;  // shared between copies, like every other part of a copied ciphertext's history

  // a fresh ciphertext over the first `limbs` normal primes
  explicit Cipher(size_t limbs = context::nr_ctxt_moduli) : bx(limbs), ax(limbs) { assert(limbs < yell::params::kMaxNbModuli); }
  // value semantics: memberwise copy, move and assignment are exactly what is wanted
  Cipher(Cipher const &) = default;
  Cipher(Cipher &&) = default;
  Cipher &operator=(Cipher const &) = default;
  Cipher &operator=(Cipher &&) = default;

  size_t moduli_count() const { return bx.moduli_count(); }
  bool canonical() const { return !mul_aux; }  // degree 1: nothing waits for relinearisation
  double scale() const { return scale_; }
  void scale(double s) { assert(s > 0.); scale_ = s; }

  // ---- device side (not in the reference: a CPU library) --------------------------------------------------------------
  using word = yell::params::value_type;
  // the ciphertext as the C-ABI takes it: [half][limb][N] contiguous in device memory (with_aux: [3][limb][N], mul_aux
  // last), current.  Parts that are not there yet (host-resident, or scattered over blocks) are brought together once; the
  // polynomials then stay views of that block.
  const word *device_parts(bool with_aux = false) const {
    const context::poly_t *part[3] = {&bx, &ax, with_aux ? mul_aux.get() : nullptr};
    const size_t n = with_aux ? 3 : 2, w = moduli_count() * context::degree;
    bool together = true;
    for (size_t k = 0; k < n; ++k)
      together = together && part[k]->on_device() && part[k]->block() == part[0]->block() && part[k]->block_offset() == part[0]->block_offset() + k * w;
    if (!together) {
      auto blk = std::make_shared<yell::dev_block>(yell::runtime::get().ctx, n * w);
      for (size_t k = 0; k < n; ++k) part[k]->relocate(blk, k * w);
    }
    return bx.cdev();
  }
  // the same for an operation that updates the ciphertext in place on the device
  word *device_parts_mut(bool with_aux = false) {
    device_parts(with_aux);
    word *p = bx.dev();
    ax.dev();
    if (with_aux) mul_aux->dev();
    return p;
  }
  // become the `limbs`-limb ciphertext a device call has written at word offset `off` of `blk` ([half][limb][N]); degree 1
  void adopt_device(yell::dev_block_ptr const &blk, size_t off, size_t limbs) {
    bx.adopt(blk, off, limbs);
    ax.adopt(blk, off + limbs * context::degree, limbs);
    mul_aux = nullptr;
  }

 private:
  double scale_ = 1.;
};
}  // namespace fHE

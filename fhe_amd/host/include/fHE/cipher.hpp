// fHE/cipher.hpp — the ciphertext value type, same members and meaning as the reference's (include/fHE/cipher.hpp):
// bx / ax are the two halves, mul_aux the degree-2 part a multiplication leaves until relinearisation, scale the CKKS scale.
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

 private:
  double scale_ = 1.;
};
}  // namespace fHE

// fHE/decryptor.hpp — Decryptor with the reference's public shape (include/fHE/decryptor.hpp:11-23).  The polynomial-level
// decrypt (src/decryptor.cpp:11-38: bx + sx*ax per limb, then the inverse NTT) is ONE device call, fhe_hip_decrypt; it is
// the integer "decrypted-plaintext" parity probe.  The value overloads decode on the host (fHE/encoder.hpp).
#pragma once
#include "fHE/cipher.hpp"
#include "fHE/encoder.hpp"
#include "fHE/keys.hpp"
#include <complex>
namespace fHE {
struct Decryptor {
  explicit Decryptor(std::shared_ptr<Encoder> encoder = nullptr) : encoder(encoder) {}
  ~Decryptor() {}

  // throws std::runtime_error like the reference when the secret key has fewer limbs than the ciphertext or the
  // ciphertext still carries its degree-2 part; false on a null result pointer
  bool decrypt(context::poly_t *rop, Cipher const &ct, SK const &sk) const {
    const size_t Li = ct.moduli_count(), n = context::degree;
    if (Li > sk.sx.moduli_count()) throw std::runtime_error("decrypt: the ciphertext has more limbs than the secret key");
    if (!ct.canonical()) throw std::runtime_error("decrypt: relinearize the ciphertext first");
    if (!rop) return false;
    rop->resize_moduli_count(Li);
    if (Li == 0) return true;
    auto out = std::make_shared<yell::dev_block>(yell::runtime::get().ctx, Li * n);
    if (fhe_hip_decrypt(yell::runtime::get().ctx, out->p, ct.device_parts(), sk.sx.cdev(), (uint32_t)Li, 1, FHE_HIP_MEM_DEVICE) != FHE_HIP_OK) return false;
    rop->adopt(out, 0, Li);   // comes to the host when the caller reads it
    return true;
  }
  bool decrypt(std::vector<std::complex<double>> *rop, Cipher const &ct, SK const &sk) const { return decrypt_then_decode(rop, ct, sk); }
  bool decrypt(std::vector<double> *rop, Cipher const &ct, SK const &sk) const { return decrypt_then_decode(rop, ct, sk); }
  bool decrypt(double *rop, Cipher const &ct, SK const &sk) const {
    std::vector<double> slots;
    if (!rop || !decrypt(&slots, ct, sk) || slots.empty()) return false;
    *rop = slots.front();
    return true;
  }

 private:
  template <class Values> bool decrypt_then_decode(Values *rop, Cipher const &ct, SK const &sk) const {
    if (!rop || !encoder) return false;
    context::poly_t plain(ct.moduli_count());
    if (!decrypt(&plain, ct, sk)) return false;
    encoder->decode(rop, plain, ct.scale());
    return true;
  }
  std::shared_ptr<Encoder> encoder;
};
}  // namespace fHE

// fHE/decryptor.hpp — polynomial-level decrypt of reference src/decryptor.cpp:11-38 (the integer
// "decrypted-plaintext" parity probe).  The Encoder overloads (:40-78) decode on the host (fHE/encoder.hpp).
#pragma once
#include "fHE/cipher.hpp"
#include "fHE/encoder.hpp"
#include "fHE/keys.hpp"
#include "fHE/yell.hpp"
namespace fHE {
struct Decryptor {
  explicit Decryptor(std::shared_ptr<Encoder> encoder = nullptr) : encoder(encoder) {}
  ~Decryptor() {}
  bool decrypt(context::poly_t *rop, Cipher const &ctx, SK const &sk) const {
    using T = yell::params::value_type; using gT = yell::params::gt_value_type;
    const size_t degree = context::degree, nmoduli = ctx.moduli_count();
    if (nmoduli > sk.sx.moduli_count()) throw std::runtime_error("Impossible.");
    if (!ctx.canonical()) throw std::runtime_error("Call relinear before decryption.");
    if (!rop) return false;
    rop->resize_moduli_count(nmoduli);
    for (size_t cm = 0; cm < nmoduli; ++cm) {
      std::vector<gT> lazy(degree);
      std::transform(ctx.bx.cptr_at(cm), ctx.bx.cptr_end(cm), lazy.data(), [](T v) { return (gT)v; });
      lazy_muladd(lazy.data(), sk.sx.cptr_at(cm), ctx.ax.cptr_at(cm), degree);  //! m = bx + sx * ax
      auto dst = rop->ptr_at(cm);
      for (auto &v : lazy) { yell::ops::barret_reduction(&v, cm); *dst++ = (T)v; }
    }
    rop->backward();
    return true;
  }
  bool decrypt(std::vector<std::complex<double>> *rop, Cipher const &ctx, SK const &sk) const { return decrypt_values(rop, ctx, sk); }  // :40-50
  bool decrypt(std::vector<double> *rop, Cipher const &ctx, SK const &sk) const { return decrypt_values(rop, ctx, sk); }  // :52-61
  bool decrypt(double *rop, Cipher const &ctx, SK const &sk) const {  // :63-77
    if (!rop) return false;
    std::vector<double> val;
    if (!decrypt(&val, ctx, sk)) return false;
    *rop = val[0];
    return true;
  }

 private:
  template <class V>
  bool decrypt_values(V *rop, Cipher const &ctx, SK const &sk) const {
    if (!rop || !encoder) return false;
    context::poly_t plain(ctx.moduli_count());
    if (!decrypt(&plain, ctx, sk)) return false;
    encoder->decode(rop, plain, ctx.scale());
    return true;
  }
  std::shared_ptr<Encoder> encoder;
};
}  // namespace fHE

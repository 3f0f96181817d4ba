// fHE/decryptor.hpp — polynomial-level decrypt of reference src/decryptor.cpp:11-38 (the integer
// "decrypted-plaintext" parity probe).  The Encoder-based overloads (:40-78) are out of the hot path.
#pragma once
#include "fHE/cipher.hpp"
#include "fHE/keys.hpp"
#include "fHE/yell.hpp"
namespace fHE {
struct Encoder;
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

 private:
  std::shared_ptr<Encoder> encoder;
};
}  // namespace fHE

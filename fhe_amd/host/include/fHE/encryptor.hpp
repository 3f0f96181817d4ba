// fHE/encryptor.hpp — polynomial-level encrypt of reference src/encryptor.cpp:9-37 (host plumbing; NTTs on the GPU).
// The Encoder overloads (encryptor.cpp:39-64) encode on the host (fHE/encoder.hpp) and then take the same path.
#pragma once
#include "fHE/cipher.hpp"
#include "fHE/encoder.hpp"
#include "fHE/keys.hpp"
namespace fHE {
struct Encryptor {
  explicit Encryptor(std::shared_ptr<Encoder> encoder = nullptr) : encoder(encoder) {}
  ~Encryptor() {}
  bool encrypt(Cipher *rop, context::poly_t const &msg, PK const &pk) const {
    if (!rop) return false;
    const size_t msg_moduli = msg.moduli_count(), ctx_moduli = rop->moduli_count();
    context::poly_t u(ctx_moduli, yell::ZO_dist());
    u.forward();
    rop->bx.set(context::gauss_struct(&context::fg_prng));  // e0
    if (msg_moduli == ctx_moduli) {
      (rop->bx) += msg;  // e0 + msg in the power basis
    } else {
      yell::ops::addmod addmod;
      for (size_t cm = 0; cm < ctx_moduli; ++cm) {
        auto msg_ptr = msg.cptr_at(0);  // use the first moduli only
        auto bx_ptr = rop->bx.ptr_at(cm);
        for (size_t d = 0; d < context::degree; ++d) addmod.compute(*bx_ptr++, *msg_ptr++, cm);
      }
    }
    rop->bx.forward_lazy();
    rop->bx.add_product_of(u, pk.bx);  // ctx.bx = u * pk.bx + e0 + msg
    rop->ax.set(context::gauss_struct(&context::fg_prng));  // e1
    rop->ax.forward_lazy();
    rop->ax.add_product_of(u, pk.ax);  // ctx.ax = u * pk.ax + e1
    return true;
  }
  bool encrypt(Cipher *rop, double const &value, PK const &pk) const { return encrypt(rop, std::vector<double>(1, value), pk); }  // :39-42
  bool encrypt(Cipher *rop, std::vector<double> const &values, PK const &pk) const { return encrypt_values(rop, values, pk); }  // :44-52
  bool encrypt(Cipher *rop, std::vector<std::complex<double>> const &values, PK const &pk) const { return encrypt_values(rop, values, pk); }  // :54-64

 private:
  template <class V>
  bool encrypt_values(Cipher *rop, V const &values, PK const &pk) const {
    if (!rop || !encoder) return false;
    context::poly_t plain(context::nr_ctxt_moduli);
    encoder->encode(&plain, values, context::encoder_scale);
    rop->scale(context::encoder_scale);
    return encrypt(rop, plain, pk);
  }
  std::shared_ptr<Encoder> encoder;
};
}  // namespace fHE

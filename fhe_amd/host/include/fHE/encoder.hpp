// fHE/encoder.hpp — host Encoder with the reference's public shape (include/fHE/encoder.hpp:6-46):
// CKKS canonical-embedding encoder.  encode / decode run on the GPU through the C-ABI (floating point: complex double, like the
// reference); a host implementation of the same mathematics is kept beside them as a cross-check.  The reference goes through GMP
// (mpz_set_d / mpz_mod_ui / poly2mpz); neither path here needs it.
#pragma once
#include "fHE/context.hpp"
#include <algorithm>
#include <cmath>
#include <complex>
#include <vector>
namespace fHE {
struct Encoder {
 private:
  using T = yell::params::value_type;
  using cplx = std::complex<double>;
  const size_t degree = context::degree;
  const size_t nr_slots = degree >> 1u;
  const size_t logn = yell::runtime::get().logn;
  // Canonical embedding.  With zeta = exp(2 pi i / 2N), a real polynomial m is fixed by its values at the N odd powers
  // zeta^(2k+1), k = 0..N-1, and those values come in conjugate pairs.  Slot j is the value at zeta^(3^j) (3 generates the
  // odd residues mod 2N up to sign), its conjugate sits at zeta^(-3^j):
  //     point_[j] = (3^j mod 2N - 1) / 2,   point_[N/2 + j] = (2N - 3^j mod 2N - 1) / 2        (the k of each of them)
  // Evaluating at all odd powers is a twist followed by an ordinary cyclic DFT of size N:
  //     m(zeta^(2k+1)) = sum_d (m_d zeta^d) omega^(k d),  omega = zeta^2 = exp(2 pi i / N).
  // (The reference reaches the same numbers with a negacyclic butterfly network over bit-reversed roots,
  // src/encoder.cpp:6-100; values agree to floating-point rounding, which the tests bound.)
  std::vector<uint32_t> point_;
  std::vector<cplx> omega_;  // omega^k, k < N/2
  std::vector<cplx> twist_;  // zeta^d, d < N

  // in-place cyclic DFT of size N, x_k <- sum_d x_d omega^(sign k d): bit-reversal permutation, then log2 N butterfly stages
  void cyclic_dft(std::vector<cplx> &x, bool conjugate) const {
    const size_t n = degree;
    for (size_t i = 0; i < n; ++i) {
      const size_t r = (size_t)yell::math::reverse_bits(i, logn);
      if (i < r) std::swap(x[i], x[r]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
      const size_t half = len >> 1, stride = n / len;
      for (size_t start = 0; start < n; start += len) {
        for (size_t k = 0; k < half; ++k) {
          const cplx w = conjugate ? std::conj(omega_[k * stride]) : omega_[k * stride];
          const cplx odd = x[start + half + k] * w, even = x[start + k];
          x[start + k] = even + odd;
          x[start + half + k] = even - odd;
        }
      }
    }
  }
  // slot values -> N * (coefficients of the real polynomial that takes them); the 1/N is folded into encode's scale
  std::vector<cplx> coefficients_times_n(std::vector<cplx> const &values) const {
    const size_t used = std::min(values.size(), nr_slots);
    if (used != values.size()) std::cerr << "WARN: Too many values to encode." << "Just handle the first " << nr_slots << " values.\n";
    std::vector<cplx> at_points(degree, 0.);
    for (size_t j = 0; j < used; ++j) {
      at_points[point_[j]] = values[j];
      at_points[point_[nr_slots + j]] = std::conj(values[j]);
    }
    cyclic_dft(at_points, true);  // sum_k E_k omega^(-k d)
    for (size_t d = 0; d < degree; ++d) at_points[d] *= std::conj(twist_[d]);
    return at_points;
  }

  // trunc(x) mod p in [0,p), what mpz_set_d + mpz_mod_ui give (encoder.cpp:135-136)
  static T double_mod(double x, T p) {
    const bool neg = x < 0;
    int e = 0;
    const double m = std::frexp(std::fabs(x), &e);  // |x| = m * 2^e, m in [0.5, 1)
    T r;
    if (e <= 63) {
      r = (T)std::fabs(x) % p;  // exact truncation
    } else {
      const T mant = (T)std::ldexp(m, 53);  // 53-bit integer mantissa
      r = yell::math::mulmod(mant % p, yell::math::powmod(2, (T)(e - 53), p), p);
    }
    return (neg && r) ? p - r : r;
  }

  // little-endian multi-word unsigned integers for the CRT lift of decode (replaces poly2mpz, encoder.cpp:171)
  using big = std::vector<uint64_t>;
  static void big_muladd(big &a, uint64_t m, uint64_t c) {  // a = a * m + c
    unsigned __int128 carry = c;
    for (auto &w : a) { carry += (unsigned __int128)w * m; w = (uint64_t)carry; carry >>= 64; }
    if (carry) a.push_back((uint64_t)carry);
  }
  static int big_cmp(big const &a, big const &b) {
    size_t n = std::max(a.size(), b.size());
    for (size_t i = n; i-- > 0;) {
      const uint64_t x = i < a.size() ? a[i] : 0, y = i < b.size() ? b[i] : 0;
      if (x != y) return x < y ? -1 : 1;
    }
    return 0;
  }
  static big big_sub(big const &a, big const &b) {  // a - b, a >= b
    big r(a.size());
    uint64_t borrow = 0;
    for (size_t i = 0; i < a.size(); ++i) {
      const uint64_t y = i < b.size() ? b[i] : 0;
      const unsigned __int128 d = (unsigned __int128)a[i] - y - borrow;
      r[i] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1;
    }
    return r;
  }
  static double big_to_double(big const &a) {
    double r = 0;
    for (size_t i = a.size(); i-- > 0;) r = r * 18446744073709551616.0 + (double)a[i];
    return r;
  }

 public:
  explicit Encoder() : point_(degree), omega_(nr_slots), twist_(degree) {
    const size_t two_n = degree << 1u;
    size_t power = 1;  // 3^j mod 2N
    for (size_t j = 0; j < nr_slots; ++j) {
      point_[j] = (uint32_t)((power - 1) >> 1);
      point_[nr_slots + j] = (uint32_t)((two_n - power - 1) >> 1);
      power = (power * 3) & (two_n - 1);
    }
    const double pi = std::acos(-1.0);
    for (size_t k = 0; k < nr_slots; ++k) omega_[k] = std::polar(1.0, 2.0 * pi * (double)k / (double)degree);
    for (size_t d = 0; d < degree; ++d) twist_[d] = std::polar(1.0, pi * (double)d / (double)degree);
  }
  ~Encoder() {}

  // The public encode / decode run on the GPU (fhe_hip_encode / fhe_hip_decode: the reference's butterfly networks in complex
  // double, rounding and CRT lift as device kernels); the plaintext polynomial stays in device memory until somebody reads it on
  // the host.  The host implementations below (a different route to the same numbers: twist + cyclic DFT, multi-word CRT) are
  // kept as an independent cross-check of the device path (fhe_amd/host/test/host_roundtrip.cpp); nothing falls back to them.
  void encode(context::poly_t *msg, std::vector<cplx> const &values, const double scale = context::encoder_scale) const {  // encoder.cpp:116-147
    if (!msg) return;
    const size_t nm = msg->moduli_count();
    if (values.size() > nr_slots) std::cerr << "WARN: Too many values to encode." << "Just handle the first " << nr_slots << " values.\n";
    fhe_hip_ctx *ctx = yell::runtime::get().ctx;
    const size_t nvals = std::min(values.size(), nr_slots);
    auto out = std::make_shared<yell::dev_block>(ctx, nm * degree);
    auto in = std::make_shared<yell::dev_block>(ctx, 2 * std::max<size_t>(nvals, 1));
    if (nvals) yell::check(fhe_hip_mem_copy(ctx, in->p, values.data(), nvals * sizeof(cplx), FHE_HIP_COPY_H2D), "encode: slots -> device");
    yell::check(fhe_hip_encode(ctx, out->p, reinterpret_cast<const double *>(in->p), nvals, (uint32_t)nm, scale, 1, FHE_HIP_MEM_DEVICE), "encode");
    msg->adopt(out, 0, nm);
  }
  void decode(std::vector<cplx> *rop, context::poly_t const &msg, double scale_) const {  // encoder.cpp:159-188
    if (!rop) return;
    fhe_hip_ctx *ctx = yell::runtime::get().ctx;
    auto out = std::make_shared<yell::dev_block>(ctx, 2 * nr_slots);
    yell::check(fhe_hip_decode(ctx, reinterpret_cast<double *>(out->p), msg.cdev(), (uint32_t)msg.moduli_count(), scale_, 1, FHE_HIP_MEM_DEVICE), "decode");
    rop->resize(nr_slots);
    yell::check(fhe_hip_mem_copy(ctx, rop->data(), out->p, nr_slots * sizeof(cplx), FHE_HIP_COPY_D2H), "decode: slots -> host");
  }

  void encode_on_host(context::poly_t *msg, std::vector<cplx> const &values, const double scale = context::encoder_scale) const {
    if (!msg) return;
    const std::vector<cplx> conj_values{coefficients_times_n(values)};
    const double delta = scale / (double)degree;
    for (size_t cm = 0; cm < msg->moduli_count(); ++cm) {
      const T P = yell::params::P[cm];
      T *dst = msg->ptr_at(cm);
      for (size_t i = 0; i < degree; ++i) dst[i] = double_mod(conj_values[i].real() * delta, P);
    }
  }
  void encode(context::poly_t *msg, std::vector<double> const &values, const double scale = context::encoder_scale) const {  // :149-157
    std::vector<cplx> com_values(values.size(), 0.);
    std::transform(values.cbegin(), values.cend(), com_values.begin(), [](double v) { return cplx(v); });
    encode(msg, com_values, scale);
  }

  void decode_on_host(std::vector<cplx> *rop, context::poly_t const &msg, double scale_) const {
    if (!rop) return;
    const size_t nmoduli = msg.moduli_count();
    // Garner mixed radix: x = v0 + v1 q0 + v2 q0 q1 + ...
    std::vector<T> inv(nmoduli, 1);  // (q0...q_{k-1})^-1 mod q_k
    for (size_t k = 1; k < nmoduli; ++k) {
      const T qk = yell::params::P[k];
      T prod = 1;
      for (size_t j = 0; j < k; ++j) prod = yell::math::mulmod(prod, yell::params::P[j] % qk, qk);
      inv[k] = yell::math::powmod(prod, qk - 2, qk);
    }
    big Q{1};
    for (size_t k = 0; k < nmoduli; ++k) big_muladd(Q, yell::params::P[k], 0);
    big Q2(Q);
    { uint64_t carry = 0; for (size_t i = Q2.size(); i-- > 0;) { const uint64_t w = Q2[i]; Q2[i] = (w >> 1) | (carry << 63); carry = w & 1; } }
    const double inv_prec = 1. / scale_;
    std::vector<cplx> res(degree, 0.);
    std::vector<T> v(nmoduli);
    for (size_t i = 0; i < degree; ++i) {
      for (size_t k = 0; k < nmoduli; ++k) {
        const T qk = yell::params::P[k];
        // partial = (v0 + v1 q0 + ... + v_{k-1} q0..q_{k-2}) mod qk, Horner from the top digit
        T partial = 0;
        for (size_t j = k; j-- > 0;) partial = (T)(((yell::params::gt_value_type)partial * (yell::params::P[j] % qk) + v[j] % qk) % qk);
        const T xk = msg.cptr_at(k)[i] % qk;
        v[k] = yell::math::mulmod((xk + qk - partial) % qk, inv[k], qk);
      }
      big x{0};
      for (size_t k = nmoduli; k-- > 0;) big_muladd(x, k + 1 < nmoduli ? yell::params::P[k] : 1, 0), big_muladd(x, 1, v[k]);
      double val;
      if (big_cmp(x, Q2) >= 0) val = -big_to_double(big_sub(Q, x));  // centred lift (:175-176)
      else val = big_to_double(x);
      res[i].real(val * inv_prec);
    }
    for (size_t d = 0; d < degree; ++d) res[d] *= twist_[d];
    cyclic_dft(res, false);  // res[k] = m(zeta^(2k+1))
    rop->resize(nr_slots);
    for (size_t j = 0; j < nr_slots; ++j) (*rop)[j] = res[point_[j]];
  }
  void decode(std::vector<double> *rop, context::poly_t const &msg, double scale_) const {  // :190-199
    if (!rop) return;
    std::vector<cplx> tmp;
    decode(&tmp, msg, scale_);
    rop->resize(tmp.size());
    std::transform(tmp.cbegin(), tmp.cend(), rop->begin(), [](cplx const &v) { return v.real(); });
  }
};
}  // namespace fHE

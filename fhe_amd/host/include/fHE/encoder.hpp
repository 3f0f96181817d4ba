// fHE/encoder.hpp — host mirror of reference include/fHE/encoder.hpp:6-46 / src/encoder.cpp:5-200:
// CKKS canonical-embedding encoder.  Floating point, host only; it exists so that Evaluator::replicate (which encodes a
// slot mask) and full pipelines can run on the mirror.  The reference goes through GMP (mpz_set_d / mpz_mod_ui /
// poly2mpz); here the same integers are formed with 128-bit and small multi-word arithmetic (no GMP dependency).
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
  const size_t generator = 3;
  std::vector<uint32_t> matrix_reps_index_map_;
  std::vector<cplx> roots_, inv_roots_;

  void FFT_inplace(std::vector<cplx> &res) const {  // encoder.cpp:38-58
    size_t tt = degree;
    for (size_t i = 0; i < logn; i++) {
      const size_t mm = size_t(1) << i;
      tt >>= 1;
      for (size_t j = 0; j < mm; j++) {
        const size_t j1 = 2 * j * tt;
        const cplx s = roots_[mm + j];
        for (size_t k = j1; k < j1 + tt; k++) {
          const cplx u = res[k], v = res[k + tt] * s;
          res[k] = u + v;
          res[k + tt] = u - v;
        }
      }
    }
  }
  void iFFT_inplace(std::vector<cplx> &res) const {  // encoder.cpp:60-83 (no 1/N: folded into delta by encode)
    size_t tt = 1;
    for (size_t i = 0; i < logn; i++) {
      const size_t h = (size_t(1) << (logn - i)) / 2;
      size_t k_start = 0;
      for (size_t j = 0; j < h; j++) {
        const cplx s = inv_roots_[h + j];
        for (size_t k = k_start; k < k_start + tt; k++) {
          const cplx u = res[k], v = res[k + tt];
          res[k] = u + v;
          res[k + tt] = (u - v) * s;
        }
        k_start += 2 * tt;
      }
      tt *= 2;
    }
  }
  std::vector<cplx> apply_ifft(std::vector<cplx> const &values) const {  // encoder.cpp:84-100
    const size_t n = std::min(values.size(), nr_slots);
    if (n != values.size()) std::cerr << "WARN: Too many values to encode." << "Just handle the first " << nr_slots << " values.\n";
    std::vector<cplx> conj_values(degree, 0.);
    for (size_t i = 0; i < n; ++i) {
      conj_values[matrix_reps_index_map_[i]] = values[i];
      conj_values[matrix_reps_index_map_[i + nr_slots]] = std::conj(values[i]);
    }
    iFFT_inplace(conj_values);
    return conj_values;
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
  explicit Encoder() : matrix_reps_index_map_(degree), roots_(degree), inv_roots_(degree) {  // encoder.cpp:6-34
    const size_t m = degree << 1u;
    uint64_t pos = 1;
    for (size_t i = 0; i < nr_slots; i++) {
      const uint64_t index1 = (pos - 1) >> 1, index2 = (m - pos - 1) >> 1;
      matrix_reps_index_map_[i] = (uint32_t)yell::math::reverse_bits(index1, logn);
      matrix_reps_index_map_[nr_slots | i] = (uint32_t)yell::math::reverse_bits(index2, logn);
      pos = (pos * generator) & (m - 1);
    }
    const double PI_ = 3.14159265358979323846;
    const cplx psi{std::cos((2 * PI_) / (double)m), std::sin((2 * PI_) / (double)m)};
    for (size_t i = 0; i < degree; i++) {
      roots_[i] = std::pow(psi, (double)yell::math::reverse_bits(i, logn));
      inv_roots_[i] = 1.0 / roots_[i];
    }
  }
  ~Encoder() {}

  void encode(context::poly_t *msg, std::vector<cplx> const &values, const double scale = context::encoder_scale) const {  // encoder.cpp:116-147
    if (!msg) return;
    const std::vector<cplx> conj_values{apply_ifft(values)};
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

  void decode(std::vector<cplx> *rop, context::poly_t const &msg, double scale_) const {  // encoder.cpp:159-188
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
    FFT_inplace(res);
    rop->resize(nr_slots);
    for (size_t i = 0; i < nr_slots; ++i) (*rop)[i] = res[matrix_reps_index_map_[i]];
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

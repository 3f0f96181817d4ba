// fHE/context.hpp — runtime replacement of the reference's compile-time `struct context`
// (reference include/fHE/context.hpp:6-25).  fHE::Context owns the device context; activating it binds the
// yell-style types and the static accessors of fHE::context to it.
#pragma once
#include "yell/poly.hpp"

namespace fHE {
struct context {
  // the reference's constants (context.hpp:7-14) become values of the active Context
  static inline size_t degree = 0;
  static inline size_t nr_ctxt_moduli = 0;
  static inline size_t nr_sp_primes = 0;
  static constexpr double sigma = 3.2;
  static constexpr double encoder_scale = (double)(1ULL << 62);
  static constexpr double plain_scale = (double)(1ULL << 20);
  static constexpr size_t HWT = 32;
  static size_t index_sp_prime(size_t k) { assert(k < nr_sp_primes); return k + nr_ctxt_moduli; }  // context.hpp:16-19
  using poly_t = yell::poly;
  using gauss_struct = yell::gaussian<uint16_t, yell::params::value_type, 2>;
  using gauss_t = yell::FastGaussianNoise<uint16_t, yell::params::value_type, 2>;
  static inline gauss_t fg_prng{sigma, 128, 1 << 14};  // context.cpp:3
};

// north_star "fHE::Context": N = 2^logN, L normal + K special primes, one GPU.
class Context {
  fhe_hip_ctx *ctx_ = nullptr;
  std::vector<uint64_t> primes_, psi_;
  uint32_t logN_, L_, K_;

  static bool is_prime(uint64_t n) {
    static const uint64_t small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    for (uint64_t q : small) { if (n % q == 0) return n == q; }
    uint64_t d = n - 1; int r = 0;
    while (!(d & 1)) { d >>= 1; ++r; }
    for (uint64_t a : small) {
      uint64_t x = yell::math::powmod(a, d, n);
      if (x == 1 || x == n - 1) continue;
      bool comp = true;
      for (int k = 1; k < r; ++k) { x = yell::math::mulmod(x, x, n); if (x == n - 1) { comp = false; break; } }
      if (comp) return false;
    }
    return true;
  }

 public:
  // the repository's prime table (DESIGN.md "Arithmetic contract"): descending primes k*2^32 + 1 below 2^62
  static void default_primes(uint32_t logN, uint32_t count, std::vector<uint64_t> &primes, std::vector<uint64_t> &psi) {
    primes.clear(); psi.clear();
    for (uint64_t k = (1ull << 30) - 1; primes.size() < count; --k) {
      const uint64_t p = (k << 32) + 1;
      if (!is_prime(p)) continue;
      uint64_t x = 2;
      while (yell::math::powmod(x, (p - 1) >> 1, p) != p - 1) ++x;
      primes.push_back(p);
      psi.push_back(yell::math::powmod(x, (p - 1) >> (logN + 1), p));
    }
  }

  Context(uint32_t logN, uint32_t L, uint32_t K, int device = 0) : logN_(logN), L_(L), K_(K) {
    default_primes(logN, L + K, primes_, psi_);
    int rc = fhe_hip_ctx_create(&ctx_, logN, L, K, primes_.data(), psi_.data(), device);
    if (rc != FHE_HIP_OK) throw std::runtime_error(std::string("fhe_hip_ctx_create: ") + fhe_hip_last_error(nullptr));
    yell::block_pool::get().open(ctx_);
    activate();
  }
  Context(Context const &) = delete;
  Context &operator=(Context const &) = delete;
  ~Context() {
    yell::runtime::get().loaded.erase(ctx_);
    if (yell::runtime::get().ctx == ctx_) yell::runtime::get().ctx = nullptr;
    yell::block_pool::get().drain(ctx_);  // polynomials that outlive the context return their blocks to a closed pool: freed at once
    fhe_hip_ctx_destroy(ctx_);
  }
  void activate() {
    auto &rt = yell::runtime::get();
    rt.ctx = ctx_; rt.degree = (size_t)1 << logN_; rt.logn = logN_;
    yell::params::P.tbl = &primes_;
    context::degree = rt.degree; context::nr_ctxt_moduli = L_; context::nr_sp_primes = K_;
  }
  fhe_hip_ctx *handle() const { return ctx_; }
  std::vector<uint64_t> const &primes() const { return primes_; }
  std::vector<uint64_t> const &roots() const { return psi_; }
};
}  // namespace fHE

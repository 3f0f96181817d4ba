// yell/poly.hpp — host-side stand-in for the subset of the `yell` polynomial library that fHE's public API
// exposes (reference include/fHE/context.hpp:2,21-23; usage listed in SURVEY.md §2 "yell poly").  This is part
// of the MI355X drop-in product: the transforms forward to the gfx950 NTT through the C-ABI, the
// pointwise helpers are plain host loops used only by key generation / encryption plumbing.
// Differences from yell, forced by the north star: the ring degree and the prime table are runtime properties
// of the active fHE::Context instead of template / compile-time constants.
#pragma once
#include <algorithm>
#include <array>
#include <cassert>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <iostream>
#include <memory>
#include <numeric>
#include <stdexcept>
#include <unordered_map>
#include <random>
#include <cstdlib>
#include <vector>

#include "fhe_hip.h"

namespace yell {
namespace params {
using value_type = uint64_t;
using gt_value_type = unsigned __int128;
using signed_type = int64_t;
constexpr size_t kModulusBitsize = 62;
constexpr size_t kMaxNbModuli = 255;
// yell::params::P[i]: global prime table of the active context (normal primes first, special primes after)
struct PrimeTable {
  const std::vector<value_type> *tbl = nullptr;
  value_type operator[](size_t i) const { return tbl->at(i); }
};
inline PrimeTable P;
}  // namespace params

// what a device context currently holds: ids of the host key objects last uploaded into it (0 = none).  Kept with the
// context, not with whoever uploaded: every Evaluator that shares the context sees the same truth.
struct device_keys {
  uint64_t mult = 0;
  std::unordered_map<size_t, uint64_t> rot, mixed;  // by galois element
};
// process-wide binding of the yell types to one device context (set by fHE::Context::activate())
struct runtime {
  fhe_hip_ctx *ctx = nullptr;
  size_t degree = 0, logn = 0;
  std::unordered_map<fhe_hip_ctx *, device_keys> loaded;
  device_keys &keys() { return loaded[ctx]; }
  static runtime &get() { static runtime r; return r; }
};
inline void check(int rc, const char *what) {
  if (rc != FHE_HIP_OK) throw std::runtime_error(std::string(what) + ": " + fhe_hip_last_error(runtime::get().ctx));
}

namespace math {
inline uint64_t reverse_bits(uint64_t x, size_t bits) {
  uint64_t r = 0;
  for (size_t i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
  return r;
}
inline params::value_type mulmod(params::value_type a, params::value_type b, params::value_type p) {
  return (params::value_type)(((params::gt_value_type)a * b) % p);
}
inline params::value_type powmod(params::value_type a, params::value_type e, params::value_type p) {
  params::value_type r = 1; a %= p;
  while (e) { if (e & 1) r = mulmod(r, a, p); a = mulmod(a, a, p); e >>= 1; }
  return r;
}
inline params::value_type inv_mod_prime(params::value_type x, size_t cm) { const auto p = params::P[cm]; return powmod(x % p, p - 2, p); }
}  // namespace math

namespace ops {
using T = params::value_type; using gT = params::gt_value_type;
struct mulmod {
  T operator()(T x, T y, size_t cm) const { return math::mulmod(x, y, params::P[cm]); }
  void compute(T &x, T y, size_t cm) const { x = (*this)(x, y, cm); }
};
struct addmod {
  T operator()(T x, T y, size_t cm) const { const T p = params::P[cm]; T s = x % p + y % p; return s >= p ? s - p : s; }
  void compute(T &x, T y, size_t cm) const { x = (*this)(x, y, cm); }
};
struct muladd {  // acc + x*y mod p
  T operator()(T acc, T x, T y, size_t cm) const { const T p = params::P[cm]; return (T)(((gT)x * y + acc) % p); }
};
inline T shoupify(T y, size_t cm) { return (T)(((gT)y << 64) / params::P[cm]); }
struct mulmod_shoup {
  T operator()(T x, T y, T yshoup, size_t cm) const {
    const T p = params::P[cm];
    T r = x * y - (T)(((gT)x * yshoup) >> 64) * p;
    return r >= p ? r - p : r;
  }
  void compute(T &x, T y, T yshoup, size_t cm) const { x = (*this)(x, y, yshoup, cm); }
};
inline void barret_reduction(gT *x, size_t cm) { *x %= params::P[cm]; }
}  // namespace ops

template <size_t n> struct static_log2 { static constexpr size_t value = 1 + static_log2<n / 2>::value; };
template <> struct static_log2<1> { static constexpr size_t value = 0; };

// splitmix64 stream shared by the samplers (the reference's PRNG is not pinned).  The process-wide stream
// prng::global() and the Gaussian sampler's own stream are seeded from the operating system's entropy source, so two processes
// never derive the same secret key or encryption randomness.  Reproducible runs (tests, host_roundtrip) set the environment
// variable FHE_HOST_SEED=<integer> or call prng::reseed(); splitmix64 is a statistical generator, not a cryptographic one:
// the host mirror's key generation is plumbing for the GPU path, and a deployment plugs in its own sampler.
struct prng {
  uint64_t state;
  explicit prng(uint64_t seed) : state(seed) {}
  // 64 bits of OS entropy, or (FHE_HOST_SEED set) a value derived from that seed and the stream's tag
  static uint64_t fresh_seed(uint64_t tag) {
    if (const char *e = std::getenv("FHE_HOST_SEED")) return std::strtoull(e, nullptr, 0) * 0x9E3779B97F4A7C15ull + tag;
    std::random_device rd;
    return ((uint64_t)rd() << 32) ^ (uint64_t)rd() ^ ((uint64_t)rd() << 17);
  }
  static void reseed(uint64_t seed) { global().state = seed * 0x9E3779B97F4A7C15ull + 0x243F6A8885A308D3ull; }
  uint64_t next() {
    uint64_t z = (state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  static prng &global() { static prng g(fresh_seed(0x243F6A8885A308D3ull)); return g; }
};
struct uniform {};
struct hwt_dist { size_t hwt; explicit hwt_dist(size_t h) : hwt(h) {} };
struct ZO_dist {};
// stands for yell::FastGaussianNoise / yell::gaussian: centred binomial of variance 10.5 (sigma ~ 3.24)
template <class In = uint16_t, class Out = params::value_type, unsigned lu = 2> struct FastGaussianNoise {
  prng rng;
  FastGaussianNoise(double /*sigma*/ = 3.2, unsigned /*bits*/ = 128, size_t /*samples*/ = 1 << 14) : rng(prng::fresh_seed(0x13198A2E03707344ull)) {}
  FastGaussianNoise(double, unsigned, size_t, uint64_t seed) : rng(seed) {}
};
template <class In = uint16_t, class Out = params::value_type, unsigned lu = 2> struct gaussian {
  FastGaussianNoise<In, Out, lu> *fg;
  explicit gaussian(FastGaussianNoise<In, Out, lu> *p) : fg(p) {}
};

// A block of device memory shared by the polynomials that view it: one polynomial, the halves of one ciphertext
// ([half][limb][N], the layout the C-ABI takes), or a whole batch of ciphertexts an operation wrote in one call.
// Released blocks go to a per-context pool and are handed out again for the same size: hipMalloc / hipFree cost far more than
// the evaluator calls they would bracket (hipFree also waits for the device).  Reuse needs no synchronisation: every use of a
// block, old or new, is enqueued on the context's one stream.  fHE::Context empties its pool when it is destroyed.
struct block_pool {
  std::unordered_map<fhe_hip_ctx *, std::unordered_map<size_t, std::vector<params::value_type *>>> free_;
  std::unordered_map<fhe_hip_ctx *, size_t> held_words_;
  static constexpr size_t kMaxHeldWords = (size_t)4 << 30;  // 32 GiB of idle blocks per context at most
  static block_pool &get() { static block_pool p; return p; }
  params::value_type *take(fhe_hip_ctx *c, size_t w) {
    auto &v = free_[c][w];
    if (v.empty()) return nullptr;
    params::value_type *p = v.back();
    v.pop_back(); held_words_[c] -= w;
    return p;
  }
  void give(fhe_hip_ctx *c, size_t w, params::value_type *p) {
    if (!free_.count(c)) return;  // the context is gone (a polynomial outlived its fHE::Context): nothing to free the block with
    if (held_words_[c] + w > kMaxHeldWords) { fhe_hip_mem_free(c, p); return; }
    free_[c][w].push_back(p); held_words_[c] += w;
  }
  void open(fhe_hip_ctx *c) { free_[c]; held_words_[c] = 0; }
  void drain(fhe_hip_ctx *c) {  // before the context is destroyed
    auto it = free_.find(c);
    if (it == free_.end()) return;
    for (auto &kv : it->second) for (auto *p : kv.second) fhe_hip_mem_free(c, p);
    free_.erase(it); held_words_.erase(c);
  }
};
struct dev_block {
  fhe_hip_ctx *ctx; params::value_type *p = nullptr; size_t words;
  dev_block(fhe_hip_ctx *c, size_t w) : ctx(c), words(w) {
    p = block_pool::get().take(c, w);
    if (!p) check(fhe_hip_mem_alloc(c, &p, w), "device allocation");
  }
  dev_block(dev_block const &) = delete;
  dev_block &operator=(dev_block const &) = delete;
  ~dev_block() { if (p) block_pool::get().give(ctx, words, p); }
};
using dev_block_ptr = std::shared_ptr<dev_block>;

// limb-major polynomial [nmoduli][degree]; limb k lives under the global modulus index k.
// Storage: a host copy and a device copy with lazy synchronisation (the "host mirror of device buffers" of SURVEY 8(a)
// row a18).  Whoever touches the polynomial through the yell API (ptr_at / cptr_at / the host loops) gets a current host
// copy, faulted back from the GPU when the device copy is the newer one; the device-side users (the transforms here, the
// Evaluator / Encryptor / Decryptor of the host mirror) read and write the device copy through dev() / cdev() / adopt(), so
// a chain of evaluator calls never moves a ciphertext over PCIe.
class poly {
  using T = params::value_type;
  mutable std::vector<T> data_;       // host copy (sized lazily while the device copy is the current one)
  mutable bool host_ok_ = true;       // host copy is current
  mutable dev_block_ptr blk_;         // device copy: blk_->p + off_, nmoduli_ * degree words
  mutable size_t off_ = 0;
  mutable bool dev_ok_ = false;       // device copy is current
  size_t nmoduli_ = 0;

  size_t words() const { return nmoduli_ * degree(); }
  void need_host() const {
    if (host_ok_) return;
    data_.resize(words());
    if (words()) check(fhe_hip_mem_copy(blk_->ctx, data_.data(), blk_->p + off_, words() * sizeof(T), FHE_HIP_COPY_D2H), "device -> host");
    host_ok_ = true;
  }
  void need_dev() const {
    if (dev_ok_) return;
    if (!blk_ || blk_->words < off_ + words() || blk_->ctx != runtime::get().ctx) {
      blk_ = std::make_shared<dev_block>(runtime::get().ctx, words());
      off_ = 0;
    }  // (a view keeps its own region of a shared block: views of one block never overlap, see adopt / relocate)
    if (words()) check(fhe_hip_mem_copy(blk_->ctx, blk_->p + off_, data_.data(), words() * sizeof(T), FHE_HIP_COPY_H2D), "host -> device");
    dev_ok_ = true;
  }
  void host_write() { need_host(); dev_ok_ = false; }   // before any host-side modification

  void set_small(const std::vector<int> &v) {
    host_write();
    const size_t N = degree();
    for (size_t cm = 0; cm < nmoduli_; ++cm) {
      const T p = params::P[cm];
      for (size_t d = 0; d < N; ++d) data_[cm * N + d] = v[d] >= 0 ? (T)v[d] : p - (T)(-v[d]);
    }
  }

 public:
  size_t logn;
  static size_t degree() { return runtime::get().degree; }

  explicit poly(size_t nmoduli = 0) : data_(nmoduli * degree(), 0), nmoduli_(nmoduli), logn(runtime::get().logn) {}
  template <class Dist> poly(size_t nmoduli, Dist const &dist) : poly(nmoduli) { set(dist); }
  // value semantics; a copy takes whichever side is current (device -> device when the host copy is stale)
  poly(poly const &o) : host_ok_(o.host_ok_), nmoduli_(o.nmoduli_), logn(o.logn) {
    if (o.host_ok_) data_ = o.data_;
    else {
      blk_ = std::make_shared<dev_block>(o.blk_->ctx, words());
      if (words()) check(fhe_hip_mem_copy(blk_->ctx, blk_->p, o.blk_->p + o.off_, words() * sizeof(T), FHE_HIP_COPY_D2D), "device copy");
      dev_ok_ = true;
    }
  }
  poly(poly &&) = default;
  poly &operator=(poly const &o) { if (this != &o) { poly t(o); *this = std::move(t); } return *this; }
  poly &operator=(poly &&) = default;

  size_t moduli_count() const { return nmoduli_; }
  T *ptr_at(size_t cm) { host_write(); return data_.data() + cm * degree(); }
  const T *cptr_at(size_t cm) const { need_host(); return data_.data() + cm * degree(); }
  const T *cptr_end(size_t cm) const { return cptr_at(cm) + degree(); }
  void resize_moduli_count(size_t n) { need_host(); data_.resize(n * degree(), 0); nmoduli_ = n; dev_ok_ = false; blk_.reset(); off_ = 0; }
  void clear() { data_.assign(words(), 0); host_ok_ = true; dev_ok_ = false; }

  // ---- device side ------------------------------------------------------------------------------------------------
  bool on_device() const { return dev_ok_; }                       // the device copy is current
  bool host_current() const { return host_ok_; }
  const T *cdev() const { need_dev(); return blk_->p + off_; }     // current device copy, read-only
  T *dev() { need_dev(); host_ok_ = false; return blk_->p + off_; }  // ... to be overwritten / updated by a device call
  dev_block_ptr const &block() const { return blk_; }
  size_t block_offset() const { return off_; }
  // become a view of `n` limbs at word offset `off` of a block a device call has just written (or is about to write)
  void adopt(dev_block_ptr const &b, size_t off, size_t n) { blk_ = b; off_ = off; nmoduli_ = n; dev_ok_ = true; host_ok_ = false; }
  // same content, device copy relocated to (b, off): used to make the halves of a ciphertext contiguous
  void relocate(dev_block_ptr const &b, size_t off) const {
    if (dev_ok_) { if (words()) check(fhe_hip_mem_copy(b->ctx, b->p + off, blk_->p + off_, words() * sizeof(T), FHE_HIP_COPY_D2D), "device copy"); }
    else if (words()) check(fhe_hip_mem_copy(b->ctx, b->p + off, data_.data(), words() * sizeof(T), FHE_HIP_COPY_H2D), "host -> device");
    blk_ = b; off_ = off; dev_ok_ = true;
  }

  void set(uniform const &) {
    host_write();
    const size_t N = degree();
    for (size_t cm = 0; cm < nmoduli_; ++cm) { const T p = params::P[cm]; for (size_t d = 0; d < N; ++d) data_[cm * N + d] = prng::global().next() % p; }
  }
  void set(hwt_dist const &h) {
    const size_t N = degree();
    std::vector<int> v(N, 0);
    for (size_t n = 0; n < h.hwt;) { uint64_t r = prng::global().next(); size_t pos = r % N; if (v[pos]) continue; v[pos] = (r >> 63) ? 1 : -1; ++n; }
    set_small(v);
  }
  void set(ZO_dist const &) {
    std::vector<int> v(degree());
    for (auto &x : v) { uint64_t r = prng::global().next() >> 62; x = r == 0 ? 1 : (r == 1 ? -1 : 0); }
    set_small(v);
  }
  template <class I, class O, unsigned l> void set(gaussian<I, O, l> const &g) {
    std::vector<int> v(degree());
    for (auto &x : v) { uint64_t r = g.fg->rng.next(); x = __builtin_popcountll(r & 0x1FFFFF) - __builtin_popcountll((r >> 21) & 0x1FFFFF); }
    set_small(v);
  }

  // yell::poly::forward/backward: every limb, on the GPU through the C-ABI, on the device copy (no PCIe round trip
  // between consecutive device-side operations)
  void forward() { transform(false); }
  void forward_lazy() { transform(false); }  // yell's lazy variant skips the final correction; canonical here
  void backward() { transform(true); }

  poly &operator+=(poly const &o) {
    assert(o.nmoduli_ == nmoduli_);
    if (dev_ok_ && o.dev_ok_ && !host_ok_) {  // both live on the GPU: add there
      check(fhe_hip_add(runtime::get().ctx, dev(), o.cdev(), (uint32_t)nmoduli_, 1, FHE_HIP_MEM_DEVICE), "add");
      return *this;
    }
    host_write(); o.need_host();
    ops::addmod add; const size_t N = degree();
    for (size_t cm = 0; cm < nmoduli_; ++cm) for (size_t d = 0; d < N; ++d) add.compute(data_[cm * N + d], o.data_[cm * N + d], cm);
    return *this;
  }
  poly &operator*=(poly const &o) {
    assert(o.nmoduli_ >= nmoduli_);
    host_write(); o.need_host();
    ops::mulmod mul; const size_t N = degree();
    for (size_t cm = 0; cm < nmoduli_; ++cm) for (size_t d = 0; d < N; ++d) mul.compute(data_[cm * N + d], o.data_[cm * N + d], cm);
    return *this;
  }
  void add_product_of(poly const &a, poly const &b) {  // *this += a * b (pointwise)
    host_write();
    ops::muladd ma; const size_t N = degree();
    for (size_t cm = 0; cm < nmoduli_; ++cm) for (size_t d = 0; d < N; ++d) data_[cm * N + d] = ma(data_[cm * N + d], a.cptr_at(cm)[d], b.cptr_at(cm)[d], cm);
  }
  void negate() {
    host_write();
    const size_t N = degree();
    for (size_t cm = 0; cm < nmoduli_; ++cm) { const T p = params::P[cm]; for (size_t d = 0; d < N; ++d) { T &x = data_[cm * N + d]; x = x ? p - x % p : 0; } }
  }

 private:
  void transform(bool inverse) {
    if (!nmoduli_) return;
    std::vector<uint32_t> idx(nmoduli_);
    std::iota(idx.begin(), idx.end(), 0u);
    auto &rt = runtime::get();
    T *d = dev();
    check(inverse ? fhe_hip_ntt_backward(rt.ctx, d, idx.data(), nmoduli_, FHE_HIP_MEM_DEVICE)
                  : fhe_hip_ntt_forward(rt.ctx, d, idx.data(), nmoduli_, FHE_HIP_MEM_DEVICE), "ntt");
  }
};

// yell::ntt<N>::forward/backward(T*, moduli_idx) — the raw-pointer seam used all over evaluator.cpp / base_converter.cpp
struct ntt {
  static void forward(params::value_type *limb, size_t moduli_idx) {
    uint32_t m = (uint32_t)moduli_idx;
    check(fhe_hip_ntt_forward(runtime::get().ctx, limb, &m, 1, FHE_HIP_MEM_HOST), "ntt::forward");
  }
  static void backward(params::value_type *limb, size_t moduli_idx) {
    uint32_t m = (uint32_t)moduli_idx;
    check(fhe_hip_ntt_backward(runtime::get().ctx, limb, &m, 1, FHE_HIP_MEM_HOST), "ntt::backward");
  }
};
}  // namespace yell

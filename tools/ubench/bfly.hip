// Microbenchmark: register-resident radix-16 Harvey butterfly throughput on gfx950 for candidate
// Montgomery formulations (checks that they agree bit for bit before timing).
//   hipcc --offload-arch=gfx950 -O3 -o bfly bfly.hip && ./bfly
#include "../../fhe_amd/csrc/modarith.h"
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

#ifndef VARIANTS
#define VARIANTS 7
#endif

template <int V> __device__ __forceinline__ void bf(u64 &X, u64 &Y, u64 W, const PrimeC &P);
template <> __device__ __forceinline__ void bf<0>(u64 &X, u64 &Y, u64 W, const PrimeC &P) { ct_bfly_ref<true>(X, Y, W, P); }
template <> __device__ __forceinline__ void bf<1>(u64 &X, u64 &Y, u64 W, const PrimeC &P) { ct_bfly<true>(X, Y, W, P); }
// variant 2 = the formulation before the "+ 2^32 - 1" addend: q0 by negation, c0 by its borrow, (hlo, 0) by a move
__device__ __forceinline__ u64 mont_diff_v2(u64 a, u64 b, const PrimeC &P) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  const u32 p1 = P.p1;
  const u64 t0 = mad_op0(a0, b0);
  const u32 T0 = (u32)t0;
  const u32 hlo = (u32)(t0 >> 32) + min(T0, 1u);
  const u64 t1 = mad_op(a1, b0, pair64(hlo, 0u));
  const u64 m = mad_op(0u - T0, p1, t1);
  u32 V2, w1, w2;
  const u64 t2 = mad_op_carry(a0, b1, m, V2);
  const u32 V0 = (u32)t2, V1 = (u32)(t2 >> 32);
  const u64 n = mad_op0(V0, p1);
  const u32 elo = __builtin_subc(V1, (u32)n, 0u, &w1);
  const u32 ehi = __builtin_subc(V2, (u32)(n >> 32), w1, &w2);
  return mad_op(a1, b1, pair64(elo, ehi));
}
template <> __device__ __forceinline__ void bf<2>(u64 &X, u64 &Y, u64 W, const PrimeC &P) {
  const u64 D = mont_diff_v2(W, Y, P);
  const u64 xp = (X >= P.twop) ? X - P.p : X + P.p;
  X = xp + D;
  Y = xp - D;
}
// variant 3 = variant 1 with the high word brought down by a multiply-add by 1 instead of a 64-bit shift
__device__ __forceinline__ u64 mad_one(u32 a, u64 c) {
  u64 d, cc;
  asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(d), "=s"(cc) : "v"(a), "v"(c));
  return d;
}
__device__ __forceinline__ u64 mont_diff_v3(u64 a, u64 b, const PrimeC &P) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  const u32 p1 = P.p1;
  const u64 t0 = mad_op_k(a0, b0);
  const u64 t1 = mad_one((u32)(t0 >> 32), mad_op0(a1, b0));
  const u64 m = mad_op(~(u32)t0, p1, t1);
  u32 V2, w1, w2;
  const u64 t2 = mad_op_carry(a0, b1, m, V2);
  const u32 V0 = (u32)t2, V1 = (u32)(t2 >> 32);
  const u64 n = mad_op0(V0, p1);
  const u32 elo = __builtin_subc(V1, (u32)n, 0u, &w1);
  const u32 ehi = __builtin_subc(V2, (u32)(n >> 32), w1, &w2);
  return mad_op(a1, b1, pair64(elo, ehi));
}
template <> __device__ __forceinline__ void bf<3>(u64 &X, u64 &Y, u64 W, const PrimeC &P) {
  const u64 D = mont_diff_v3(W, Y, P);
  const u64 xp = (X >= P.twop) ? X - P.p : X + P.p;
  X = xp + D;
  Y = xp - D;
}

// variant 4 = variant 1 with compiler-generated multiply-adds kept in order by empty asm statements (no inline-asm hazard
// padding after each multiply-add; the carry multiply-add stays an asm block)
__device__ __forceinline__ void keep(u64 &x) { asm("" : "+v"(x)); }
__device__ __forceinline__ u64 mont_diff_v4(u64 a, u64 b, const PrimeC &P) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  const u32 p1 = P.p1;
  u64 t0 = (u64)a0 * b0 + 0xFFFFFFFFull; keep(t0);
  u64 t1 = (u64)a1 * b0 + (t0 >> 32); keep(t1);
  u64 m = (u64)(~(u32)t0) * p1 + t1; keep(m);
  u32 V2, w1, w2;
  const u64 t2 = mad_op_carry(a0, b1, m, V2);
  const u32 V0 = (u32)t2, V1 = (u32)(t2 >> 32);
  u64 n = (u64)V0 * p1; keep(n);
  const u32 elo = __builtin_subc(V1, (u32)n, 0u, &w1);
  const u32 ehi = __builtin_subc(V2, (u32)(n >> 32), w1, &w2);
  u64 e = pair64(elo, ehi); keep(e);
  u64 d = (u64)a1 * b1 + e; keep(d);
  return d;
}
template <> __device__ __forceinline__ void bf<4>(u64 &X, u64 &Y, u64 W, const PrimeC &P) {
  const u64 D = mont_diff_v4(W, Y, P);
  const u64 xp = (X >= P.twop) ? X - P.p : X + P.p;
  X = xp + D;
  Y = xp - D;
}

// variant 5 = variant 1 with the high word brought down by the compiler's own lowering of t0 >> 32 (a move plus a zero)
__device__ __forceinline__ u64 mont_diff_v5(u64 a, u64 b, const PrimeC &P) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  const u32 p1 = P.p1;
  const u64 t0 = mad_op_k(a0, b0);
  const u64 t1 = mad_op(a1, b0, t0 >> 32);
  const u64 m = mad_op(~(u32)t0, p1, t1);
  u32 V2, w1, w2;
  const u64 t2 = mad_op_carry(a0, b1, m, V2);
  const u32 V0 = (u32)t2, V1 = (u32)(t2 >> 32);
  const u64 n = mad_op0(V0, p1);
  const u32 elo = __builtin_subc(V1, (u32)n, 0u, &w1);
  const u32 ehi = __builtin_subc(V2, (u32)(n >> 32), w1, &w2);
  return mad_op(a1, b1, pair64(elo, ehi));
}
template <> __device__ __forceinline__ void bf<5>(u64 &X, u64 &Y, u64 W, const PrimeC &P) {
  const u64 D = mont_diff_v5(W, Y, P);
  const u64 xp = (X >= P.twop) ? X - P.p : X + P.p;
  X = xp + D;
  Y = xp - D;
}

// variant 6 = variant 5 with the 2p reduction of X decided on the high word by plain 32-bit operations
// (values then live in [0, 4p + 2^32))
__device__ __forceinline__ u64 xp_fast(u64 X, const PrimeC &P) {
  const int dd = (int)(2u * P.p1 - (u32)(X >> 32));  // < 0  <=>  X.hi > 2*p1  =>  subtract p, else add p
  const u32 mask = (u32)(dd >> 31);
  return X + pair64(mask | 1u, P.p1 ^ mask);
}
template <> __device__ __forceinline__ void bf<6>(u64 &X, u64 &Y, u64 W, const PrimeC &P) {
  const u64 D = mont_diff_v5(W, Y, P);
  const u64 xp = xp_fast(X, P);
  X = xp + D;
  Y = xp - D;
}

template <int V>
__global__ __launch_bounds__(256) void k(u64 *d, const u64 *w, PrimeC P, int iters) {
  u64 x[16], tw[15];
  const size_t base = (size_t)blockIdx.x * 4096 + threadIdx.x;
  for (int i = 0; i < 16; ++i) x[i] = d[base + 256 * i];
  for (int i = 0; i < 15; ++i) tw[i] = w[(threadIdx.x & 63) * 15 + i];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int h = 8 >> s;
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (!(i & h)) bf<V>(x[i], x[i + h], tw[(1 << s) - 1 + (i >> (4 - s))], P);
    }
  }
  for (int i = 0; i < 16; ++i) d[base + 256 * i] = x[i];
}

static u64 mulmod(u64 a, u64 b, u64 p) { return (u64)((unsigned __int128)a * b % p); }

int main() {
  const u64 p = ((1ull << 30) - 1) * (1ull << 32) + 1;  // modulus shape only; primality is irrelevant here
  PrimeC P{};
  P.p = p; P.twop = 2 * p; P.p1 = (u32)(p >> 32);
  u64 inv = 1; for (int i = 0; i < 6; ++i) inv *= 2 - p * inv; P.pinv = inv;
  const int nblk = 256 * 8, iters = 512;
  const size_t n = (size_t)nblk * 4096;
  std::vector<u64> h(n), w(64 * 15);
  u64 s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
  for (auto &v : h) v = rnd() % (4 * p > p ? 4 * p : p);
  for (auto &v : w) v = rnd() % p;
  // corner cases in the first block
  h[0] = 0; h[1] = 4 * p - 1; h[2] = 2 * p; h[3] = 2 * p - 1; h[4] = p; w[0] = 0; w[1] = p - 1; w[2] = 1ull << 32; w[3] = 0xffffffffull;
  u64 *d[VARIANTS], *dw;
  CHECK(hipMalloc(&dw, w.size() * 8));
  CHECK(hipMemcpy(dw, w.data(), w.size() * 8, hipMemcpyHostToDevice));
  std::vector<std::vector<u64>> out(VARIANTS, std::vector<u64>(n));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int v = 0; v < VARIANTS; ++v) {
    CHECK(hipMalloc(&d[v], n * 8));
    for (int rep = 0; rep < 120; ++rep) {  // ~0.5 s per variant: the clocks need ~0.3 s of load to settle
      CHECK(hipMemcpy(d[v], h.data(), n * 8, hipMemcpyHostToDevice));
      CHECK(hipEventRecord(e0));
      if (v == 0) k<0><<<nblk, 256>>>(d[v], dw, P, iters);
      else if (v == 1) k<1><<<nblk, 256>>>(d[v], dw, P, iters);
      else if (v == 2) k<2><<<nblk, 256>>>(d[v], dw, P, iters);
      else if (v == 3) k<3><<<nblk, 256>>>(d[v], dw, P, iters);
      else if (v == 4) k<4><<<nblk, 256>>>(d[v], dw, P, iters);
      else if (v == 5) k<5><<<nblk, 256>>>(d[v], dw, P, iters);
      else k<6><<<nblk, 256>>>(d[v], dw, P, iters);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      const double bf = (double)nblk * 256 * 32 * iters;
      if (rep == 1 || rep == 119) printf("variant %d: %.3f ms  %.3e bfly/s\n", v, ms, bf / (ms * 1e-3));
    }
    CHECK(hipMemcpy(out[v].data(), d[v], n * 8, hipMemcpyDeviceToHost));
  }
  // both variants are lazy ([0,4p) representatives may differ): compare residues
  size_t bad = 0;
  for (size_t i = 0; i < n; ++i) {
    bool ok = true;
    for (int v = 1; v < VARIANTS; ++v) ok = ok && out[v][i] < 4 * p + (1ull << 32) && out[0][i] % p == out[v][i] % p;
    if (out[0][i] >= 4 * p || !ok) { if (bad < 5) printf("mismatch %zu: %llx %llx\n", i, out[0][i], out[1][i]); ++bad; }
  }
  printf("mismatches: %zu\n", bad);
  (void)mulmod;
  return bad != 0;
}

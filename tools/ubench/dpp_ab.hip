// A/B: exchange between two register phases of the forward row pass through LDS (what ntt_rows8_fwd_kernel does) against a
// quad transpose with DPP (v_mov_b32_dpp quad_perm + select), VERDICT r1 item 1d / north_star "small-stride stages use
// wavefront-level DPP/shuffle".  Register-resident loop of real butterflies (modarith.h), one row of 256 elements per 32 lanes:
//   phase B (spans 16, 8, 4: 12 butterflies per thread on elements {32c + r + 4k})  ->  exchange  ->  phase C (spans 2, 1: 8
//   butterflies on 8 contiguous elements)  ->  exchange back.
// The LDS form needs no vector-ALU instruction for the exchange (addresses are loop invariants); the DPP form needs
// 2 rounds x 8 elements x 2 words of v_mov_dpp/v_cndmask per direction.
//   hipcc --offload-arch=gfx950 -O3 -I../../fhe_amd/csrc -o dpp_ab dpp_ab.hip && ./dpp_ab
#include <hip/hip_runtime.h>
#include <cstdio>
#include "modarith.h"
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
constexpr int ITERS = 256;

__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
template <int LK> __device__ __forceinline__ void phase(u64 (&e)[1 << LK], const u64 *w, const PrimeC &P) {
#pragma unroll
  for (int q = 0; q < LK; ++q) {
    const int half = 1 << (LK - 1 - q);
#pragma unroll
    for (int kk = 0; kk < (1 << q); ++kk)
#pragma unroll
      for (int j = 0; j < half; ++j) ct_bfly<true>(e[kk * 2 * half + j], e[kk * 2 * half + j + half], w[(1 << q) - 1 + kk], P);
  }
}
// swap of register halves between lanes r and r ^ X of a quad: lane with bit X clear keeps lo[], gets partner's lo[] into hi[]
template <int CTRL> __device__ __forceinline__ u64 dpp64(u64 v) {
  const u32 lo = __builtin_amdgcn_mov_dpp((int)(u32)v, CTRL, 0xf, 0xf, true), hi = __builtin_amdgcn_mov_dpp((int)(u32)(v >> 32), CTRL, 0xf, 0xf, true);
  return ((u64)hi << 32) | lo;
}
template <int CTRL> __device__ __forceinline__ void quad_swap(u64 &a, u64 &b, bool upper) {
  // lower lane: (a, b) -> (a, partner.a);  upper lane: (a, b) -> (partner.b, b)
  const u64 pa = dpp64<CTRL>(a), pb = dpp64<CTRL>(b);
  const u64 na = upper ? pb : a, nb = upper ? b : pa;
  a = na; b = nb;
}
template <bool DPP> __global__ __launch_bounds__(256) void k(u64 *out, const u64 *tw, PrimeC P) {
  __shared__ __attribute__((aligned(16))) u64 lds[8 * 8 * 36];
  const u32 tid = threadIdx.x, rloc = tid >> 5, t = tid & 31, c = t >> 2, r = t & 3;
  u64 *L = lds + rloc * 8 * 36;
  u64 e[8], wB[7], wC0[3], wC1[3];
  for (int i = 0; i < 8; ++i) e[i] = tw[(tid * 8 + i) & 1023] >> 2;
  for (int i = 0; i < 7; ++i) wB[i] = tw[(t + i) & 1023];
  for (int i = 0; i < 3; ++i) { wC0[i] = tw[(2 * t + i) & 1023]; wC1[i] = tw[(2 * t + 8 + i) & 1023]; }
  const u32 p0 = (8 * t >> 5) * 36 + ((8 * t) & 31);
  for (int it = 0; it < ITERS; ++it) {
    phase<3>(e, wB, P);
    if constexpr (!DPP) {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) L[c * 36 + r + 4 * kk] = e[kk];
      wave_sync();
#pragma unroll
      for (int kk = 0; kk < 8; kk += 2) { const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(&L[p0 + kk]); e[kk] = v.x; e[kk + 1] = v.y; }
    } else {
      // 8 registers x 4 lanes: two rounds (lane xor 1, lane xor 2); afterwards every lane holds two groups of four consecutive elements
#pragma unroll
      for (int kk = 0; kk < 8; kk += 2) quad_swap<0xB1>(e[kk], e[kk + 1], r & 1);       // quad_perm [1,0,3,2]
#pragma unroll
      for (int kk = 0; kk < 8; kk += 4) { quad_swap<0x4E>(e[kk], e[kk + 2], r & 2); quad_swap<0x4E>(e[kk + 1], e[kk + 3], r & 2); }  // [2,3,0,1]
    }
    u64 g0[4] = {e[0], e[1], e[2], e[3]}, g1[4] = {e[4], e[5], e[6], e[7]};
    phase<2>(g0, wC0, P);
    phase<2>(g1, wC1, P);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) { e[kk] = g0[kk]; e[4 + kk] = g1[kk]; }
    if constexpr (!DPP) {
      wave_sync();
#pragma unroll
      for (int kk = 0; kk < 8; kk += 2) *reinterpret_cast<ulonglong2 *>(&L[p0 + kk]) = make_ulonglong2(e[kk], e[kk + 1]);
      wave_sync();
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) e[kk] = L[c * 36 + r + 4 * kk];
      wave_sync();
    } else {
#pragma unroll
      for (int kk = 0; kk < 8; kk += 4) { quad_swap<0x4E>(e[kk], e[kk + 2], r & 2); quad_swap<0x4E>(e[kk + 1], e[kk + 3], r & 2); }
#pragma unroll
      for (int kk = 0; kk < 8; kk += 2) quad_swap<0xB1>(e[kk], e[kk + 1], r & 1);
    }
  }
  u64 s = 0;
  for (int i = 0; i < 8; ++i) s ^= e[i];
  out[blockIdx.x * blockDim.x + tid] = s;
}

int main() {
  PrimeC P; const u64 p = (((1ull << 30) - 35) << 32) + 1;  // an SP-form modulus (primality is irrelevant to the timing)
  P.p = p; P.twop = 2 * p; P.pinv = 0; P.r2 = 1; P.ninv_m = 1; P.p1 = (u32)(p >> 32); P.n32 = 0u - (u32)P.pinv;
  u64 *out, *tw; CHECK(hipMalloc(&out, 256 * 8 * 256 * 8)); CHECK(hipMalloc(&tw, 1024 * 8));
  u64 h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (0x9E3779B97F4A7C15ull * (i + 1)) % p;
  CHECK(hipMemcpy(tw, h, sizeof(h), hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  printf("exchange A/B, %d iterations of (12 + 8 butterflies + two exchanges) per thread, 256 CUs; butterflies per second over the whole chip\n", ITERS);
  for (int wps : {2, 4, 6}) {
    for (int dpp = 0; dpp < 2; ++dpp) {
      float best = 1e9;
      for (int rep = 0; rep < 40; ++rep) {   // long enough to reach the sustained clock
        CHECK(hipEventRecord(e0));
        if (dpp) k<true><<<256 * wps, 256>>>(out, tw, P); else k<false><<<256 * wps, 256>>>(out, tw, P);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep >= 20 && ms < best) best = ms;
      }
      const double bf = 256.0 * wps * 256 * ITERS * 20;
      printf("waves/SIMD=%d  %-4s  %.3f ms  %.3e butterflies/s\n", wps, dpp ? "DPP" : "LDS", best, bf / (best * 1e-3));
    }
  }
  return 0;
}

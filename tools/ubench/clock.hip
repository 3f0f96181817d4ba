// In-kernel clock under integer-multiply load: delta(s_memtime) / delta(s_memrealtime @100 MHz).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64; typedef unsigned int u32;
#include "../../fhe_amd/csrc/modarith.h"
__global__ void k_bfly(u64* out, u64* stamps, PrimeC P, int iters) {
  u64 x[8], y[8];
  for (int i = 0; i < 8; ++i) { x[i] = (threadIdx.x * 0x9E3779B97F4A7C15ull + i) >> 3; y[i] = x[i] * 3 + 1; }
  u64 c0 = clock64(), r0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    u64 W = P.r2 + it;
#pragma unroll
    for (int i = 0; i < 8; ++i) ct_bfly<true>(x[i], y[i], W, P);
  }
  u64 c1 = clock64(), r1 = wall_clock64();
  u64 s = 0; for (int i = 0; i < 8; ++i) s += x[i] ^ y[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}
__global__ void k_add(u64* out, u64* stamps, int iters) {
  u32 x[8]; for (int i = 0; i < 8; ++i) x[i] = threadIdx.x + i;
  u64 c0 = clock64(), r0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(it));
  }
  u64 c1 = clock64(), r1 = wall_clock64();
  u32 s = 0; for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}
int main() {
  const int grid = 256 * 8, block = 256;
  u64 *out, *st; hipMalloc(&out, (size_t)grid * block * 8); hipMalloc(&st, grid * 16);
  PrimeC P; P.p = ((u64)((1u << 30) - 35) << 32) + 1; P.twop = 2 * P.p; P.p1 = (u32)(P.p >> 32); P.r2 = 12345; P.pinv = 1; P.ninv_m = 1; P.n32 = 0u - (u32)P.pinv;
  std::vector<u64> h(grid * 2);
  for (int rep = 0; rep < 3; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int iters = 40000;
    hipEventRecord(e0); hipLaunchKernelGGL(k_bfly, dim3(grid), dim3(block), 0, 0, out, st, P, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), st, grid * 16, hipMemcpyDeviceToHost);
    double cs = 0, rs = 0; for (int b = 0; b < grid; ++b) { cs += h[2 * b]; rs += h[2 * b + 1]; }
    double ghz = cs / rs * 0.1;
    double nb = (double)grid * block / 64 * iters * 8;   // wave-butterflies
    printf("bfly: %.2f ms, in-kernel clock %.3f GHz, %.1f actual cycles per wave-butterfly (per SIMD), %.3e bfly/s\n", ms, ghz,
           ms * 1e-3 * ghz * 1e9 * 1024 / nb, nb * 64 / (ms * 1e-3));
    iters = 400000;
    hipEventRecord(e0); hipLaunchKernelGGL(k_add, dim3(grid), dim3(block), 0, 0, out, st, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h.data(), st, grid * 16, hipMemcpyDeviceToHost);
    cs = rs = 0; for (int b = 0; b < grid; ++b) { cs += h[2 * b]; rs += h[2 * b + 1]; }
    ghz = cs / rs * 0.1;
    double ni = (double)grid * block / 64 * iters * 8;
    printf("add:  %.2f ms, in-kernel clock %.3f GHz, %.2f actual cycles per v_add_u32\n", ms, ghz, ms * 1e-3 * ghz * 1e9 * 1024 / ni);
  }
  return 0;
}

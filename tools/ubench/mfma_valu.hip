// Microbenchmark: do v_mfma_i32_16x16x64_i8 and the vector-ALU instructions of the base-conversion fold (v_mad_u64_u32, 64-bit
// adds, plain 32-bit ops) overlap on one SIMD of gfx950?  512-thread workgroups (two waves per SIMD), one workgroup per CU.
//   mode 0: every wave MFMA only            mode 1: every wave VALU only
//   mode 2: waves 0-3 MFMA, waves 4-7 VALU (one of each per SIMD)
//   mode 3: every wave, phases: 64 MFMAs then the VALU block (the kernel's shape)
//   mode 4: every wave, interleaved: 1 MFMA + (VALU block / 64) alternating
// VALU block = VB x (v_mad_u64_u32 + v_lshl_add_u64 + v_add_u32 + v_and_b32) on independent chains.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu mfma_valu.hip && ./mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned int u32;
typedef int v4i __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
constexpr int ITERS = 512;
constexpr int VB = 64;  // 4 instructions each: 256 VALU per iteration against 64 MFMAs

#define VALU4(i)                                                                                        \
  asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_lshl_add_u64 %1, %1, 1, %0\n\tv_add_u32 %2, %2, %3\n\tv_and_b32 %3, %3, %2" \
               : "+v"(x[(i) & 7]), "+v"(y[(i) & 7]), "+v"(p[(i) & 7]), "+v"(q[(i) & 7]) : : "vcc")

template <int MODE>
__global__ __launch_bounds__(512) void k(u32 *out, u32 a, u32 b) {
  const u32 wave = threadIdx.x >> 6;
  v4i A = {(int)(a + threadIdx.x), (int)b, (int)a, (int)(b ^ threadIdx.x)}, B = {(int)b, (int)a, (int)(a * 3), (int)threadIdx.x};
  v4i acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = v4i{i, i, i, i};
  u64 x[8], y[8]; u32 p[8], q[8];
  for (int i = 0; i < 8; ++i) { x[i] = a + i; y[i] = b + threadIdx.x; p[i] = a ^ i; q[i] = b + i; }
  const bool do_m = MODE == 0 || MODE >= 3 || (MODE == 2 && wave < 4);
  const bool do_v = MODE == 1 || MODE >= 3 || (MODE == 2 && wave >= 4);
  for (int it = 0; it < ITERS; ++it) {
    if constexpr (MODE == 4) {
#pragma unroll
      for (int m = 0; m < 64; ++m) {
        acc[m & 15] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, acc[m & 15], 0, 0, 0);
#pragma unroll
        for (int v = 0; v < VB / 64; ++v) VALU4(m * (VB / 64) + v);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      if (do_m) {
#pragma unroll
        for (int m = 0; m < 64; ++m) acc[m & 15] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, acc[m & 15], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (do_v) {
#pragma unroll
        for (int v = 0; v < VB; ++v) VALU4(v);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  v4i s = acc[0];
  for (int i = 1; i < 16; ++i) s += acc[i];
  u64 t = 0;
  for (int i = 0; i < 8; ++i) t += x[i] + y[i] + p[i] + q[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)(s[0] + s[1] + s[2] + s[3]) + (u32)t + (u32)(t >> 32);
}

template <int MODE>
int run(u32 *d, const char *what) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, 1u, 2u);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  const int reps = 20;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, d, 1u, 2u);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  printf("mode %d  %-62s %8.1f ns per iteration (per wave: 64 MFMAs and/or %d VALU)\n", MODE, what, ms * 1e6 / reps / ITERS, VB * 4);
  return 0;
}

int main() {
  u32 *d;
  CHECK(hipMalloc(&d, 256 * 512 * 4));
  if (run<0>(d, "MFMA only, 2 waves per SIMD")) return 1;
  if (run<1>(d, "VALU only, 2 waves per SIMD")) return 1;
  if (run<2>(d, "one MFMA wave + one VALU wave per SIMD")) return 1;
  if (run<3>(d, "every wave: 64 MFMAs, then the VALU block")) return 1;
  if (run<4>(d, "every wave: 1 MFMA + 4 VALU interleaved")) return 1;
  return 0;
}

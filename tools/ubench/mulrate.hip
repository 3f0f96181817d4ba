// Microbenchmark: integer multiply issue rates and register-resident Harvey butterfly
// throughput on gfx950.  Decides the butterfly formulation used by the NTT kernels
// ("first action on the GPU box").
//   hipcc --offload-arch=gfx950 -O3 -o mulrate mulrate.hip && ./mulrate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

typedef unsigned long long u64;
typedef unsigned int u32;

constexpr int ITERS = 4096;
constexpr int NACC = 8;

// ---- raw instruction rates (inline asm so the compiler cannot fold) --------------------
__global__ void k_mul_lo(u32* out, u32 a, u32 b) {
  u32 x[NACC];
  for (int i = 0; i < NACC; ++i) x[i] = a + threadIdx.x + i;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
  }
  u32 s = 0; for (int i = 0; i < NACC; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mul_hi(u32* out, u32 a, u32 b) {
  u32 x[NACC];
  for (int i = 0; i < NACC; ++i) x[i] = a + threadIdx.x + i;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
  }
  u32 s = 0; for (int i = 0; i < NACC; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mad64(u32* out, u32 a, u32 b) {
  u64 x[NACC];
  for (int i = 0; i < NACC; ++i) x[i] = a + threadIdx.x + i;
  u32 m = a ^ threadIdx.x;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i)
      asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x[i]) : "v"(m), "v"(b) : "vcc");
  }
  u64 s = 0; for (int i = 0; i < NACC; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)(s ^ (s >> 32));
}
__global__ void k_add32(u32* out, u32 a, u32 b) {
  u32 x[NACC];
  for (int i = 0; i < NACC; ++i) x[i] = a + threadIdx.x + i;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(b));
  }
  u32 s = 0; for (int i = 0; i < NACC; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_add64(u32* out, u32 a, u32 b) {
  u64 x[NACC];
  for (int i = 0; i < NACC; ++i) x[i] = a + threadIdx.x + i;
  u64 bb = ((u64)b << 32) | a;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(x[i]) : "v"(bb));
  }
  u64 s = 0; for (int i = 0; i < NACC; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)(s ^ (s >> 32));
}
__global__ void k_mul24(u32* out, u32 a, u32 b) {
  u32 x[NACC];
  for (int i = 0; i < NACC; ++i) x[i] = a + threadIdx.x + i;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(x[i]) : "v"(b));
  }
  u32 s = 0; for (int i = 0; i < NACC; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_fma64(u32* out, u32 a, u32 b) {
  double x[NACC];
  for (int i = 0; i < NACC; ++i) x[i] = 1.0 + 1e-9 * (threadIdx.x + i);
  double m = 1.0 + 1e-12 * a, c = 1e-15 * b;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(m), "v"(c));
  }
  double s = 0; for (int i = 0; i < NACC; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)s;
}

// ---- butterflies: 8 (x,y) pairs per thread, each iteration one CT butterfly ------------
// V0: textbook Harvey with __umul64hi (compiler-lowered)
__device__ __forceinline__ void bfly_v0(u64& X, u64& Y, u64 W, u64 Wp, u64 p, u64 twop) {
  u64 x = X - (X >= twop ? twop : 0);
  u64 q = __umul64hi(Wp, Y);
  u64 t = W * Y - q * p;           // [0, 2p)
  X = x + t;
  Y = x - t + twop;
}
__global__ void k_bfly_v0(u64* out, u64 W, u64 Wp, u64 p) {
  u64 x[NACC], y[NACC];
  u64 twop = 2 * p;
  for (int i = 0; i < NACC; ++i) { x[i] = (threadIdx.x * 0x9E3779B97F4A7C15ull + i) % p; y[i] = (x[i] * 3 + 1) % p; }
  for (int it = 0; it < ITERS / 4; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) bfly_v0(x[i], y[i], W + it, Wp + it, p, twop);
  }
  u64 s = 0; for (int i = 0; i < NACC; ++i) s += x[i] ^ y[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// V1: hand-decomposed: mulhi via 4 mad_u64_u32 exact; low products via mad_u64_u32 + mul_lo
__device__ __forceinline__ u64 mad_wide(u32 a, u32 b, u64 c) {
  return (u64)a * (u64)b + c;   // compiler emits v_mad_u64_u32
}
__device__ __forceinline__ u64 mulhi64_exact(u64 a, u64 b) {
  u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u64 t0 = (u64)__umulhi(a0, b0);
  u64 t1 = mad_wide(a1, b0, t0);
  u64 t2 = mad_wide(a0, b1, (u64)(u32)t1);
  return mad_wide(a1, b1, (t1 >> 32) + (t2 >> 32));
}
// approximate: drops a0*b0 and the carry out of the two cross-term low halves: result in {q-2..q}
__device__ __forceinline__ u64 mulhi64_approx(u64 a, u64 b) {
  u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u64 t1 = (u64)__umulhi(a1, b0);
  u64 t2 = (u64)__umulhi(a0, b1);
  return mad_wide(a1, b1, t1 + t2);
}
__device__ __forceinline__ u64 mullo64(u64 a, u64 b) {
  u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  u64 r = mad_wide(a0, b0, 0);
  u32 hi = (u32)(r >> 32) + a0 * b1 + a1 * b0;
  return ((u64)hi << 32) | (u32)r;
}
__device__ __forceinline__ void bfly_v1(u64& X, u64& Y, u64 W, u64 Wp, u64 p, u64 twop) {
  u64 x = X - (X >= twop ? twop : 0);
  u64 q = mulhi64_exact(Wp, Y);
  u64 t = mullo64(W, Y) - mullo64(q, p);
  X = x + t;
  Y = x - t + twop;
}
__global__ void k_bfly_v1(u64* out, u64 W, u64 Wp, u64 p) {
  u64 x[NACC], y[NACC];
  u64 twop = 2 * p;
  for (int i = 0; i < NACC; ++i) { x[i] = (threadIdx.x * 0x9E3779B97F4A7C15ull + i) % p; y[i] = (x[i] * 3 + 1) % p; }
  for (int it = 0; it < ITERS / 4; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) bfly_v1(x[i], y[i], W + it, Wp + it, p, twop);
  }
  u64 s = 0; for (int i = 0; i < NACC; ++i) s += x[i] ^ y[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// V2: approximate mulhi (3 multiplies), t in [0,4p) -> needs an extra correction; measures mult savings
__device__ __forceinline__ void bfly_v2(u64& X, u64& Y, u64 W, u64 Wp, u64 p, u64 twop) {
  u64 x = X - (X >= twop ? twop : 0);
  u64 q = mulhi64_approx(Wp, Y);
  u64 t = mullo64(W, Y) - mullo64(q, p);
  t = t - (t >= twop ? twop : 0);
  X = x + t;
  Y = x - t + twop;
}
__global__ void k_bfly_v2(u64* out, u64 W, u64 Wp, u64 p) {
  u64 x[NACC], y[NACC];
  u64 twop = 2 * p;
  for (int i = 0; i < NACC; ++i) { x[i] = (threadIdx.x * 0x9E3779B97F4A7C15ull + i) % p; y[i] = (x[i] * 3 + 1) % p; }
  for (int it = 0; it < ITERS / 4; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) bfly_v2(x[i], y[i], W + it, Wp + it, p, twop);
  }
  u64 s = 0; for (int i = 0; i < NACC; ++i) s += x[i] ^ y[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// V3: t = lo64(W*Y - q*p) computed as one fused low-64 expression: the two low products share
// the final 32-bit high word accumulate (saves adds, same multiplies).
__device__ __forceinline__ void bfly_v3(u64& X, u64& Y, u64 W, u64 Wp, u64 p, u64 twop) {
  u64 x = X - (X >= twop ? twop : 0);
  u64 q = mulhi64_exact(Wp, Y);
  u32 w0 = (u32)W, w1 = (u32)(W >> 32), y0 = (u32)Y, y1 = (u32)(Y >> 32);
  u32 q0 = (u32)q, q1 = (u32)(q >> 32), p0 = (u32)p, p1 = (u32)(p >> 32);
  u64 a = mad_wide(w0, y0, 0);
  u64 b = mad_wide(q0, p0, 0);
  u32 hi = w0 * y1 + w1 * y0 - q0 * p1 - q1 * p0;
  u64 t = a - b + ((u64)hi << 32);
  X = x + t;
  Y = x - t + twop;
}
__global__ void k_bfly_v3(u64* out, u64 W, u64 Wp, u64 p) {
  u64 x[NACC], y[NACC];
  u64 twop = 2 * p;
  for (int i = 0; i < NACC; ++i) { x[i] = (threadIdx.x * 0x9E3779B97F4A7C15ull + i) % p; y[i] = (x[i] * 3 + 1) % p; }
  for (int it = 0; it < ITERS / 4; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) bfly_v3(x[i], y[i], W + it, Wp + it, p, twop);
  }
  u64 s = 0; for (int i = 0; i < NACC; ++i) s += x[i] ^ y[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K, typename... A>
static double time_kernel(K k, int grid, int block, A... args) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, args...);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(grid), dim3(block), 0, 0, args...);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / 5.0 * 1e-3;
}

int main() {
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  int cus = prop.multiProcessorCount; double clk = prop.clockRate * 1e3;
  printf("device %s CUs %d clock %.0f MHz\n", prop.name, cus, clk / 1e6);
  const int block = 256, grid = cus * 8;   // 8 waves per SIMD
  void* out; CHECK(hipMalloc(&out, (size_t)grid * block * 8));
  double nwave = (double)grid * block / 64.0;
  double simds = cus * 4.0;
  auto report = [&](const char* name, double sec, double ops_per_thread) {
    double wave_instr = nwave * ops_per_thread;
    double cyc = sec * clk * simds / wave_instr;   // SIMD-cycles per wave-instruction (at nominal clock)
    printf("%-14s %8.3f ms  %6.2f cyc/wave-op  %8.2f Gop/s(lane)\n", name, sec * 1e3, cyc,
           wave_instr * 64 / sec * 1e-9);
  };
  report("v_add_u32", time_kernel(k_add32, grid, block, (u32*)out, 3u, 5u), (double)ITERS * NACC);
  report("v_lshl_add_u64", time_kernel(k_add64, grid, block, (u32*)out, 3u, 5u), (double)ITERS * NACC);
  report("v_mul_u32_u24", time_kernel(k_mul24, grid, block, (u32*)out, 3u, 5u), (double)ITERS * NACC);
  report("v_mul_lo_u32", time_kernel(k_mul_lo, grid, block, (u32*)out, 3u, 5u), (double)ITERS * NACC);
  report("v_mul_hi_u32", time_kernel(k_mul_hi, grid, block, (u32*)out, 3u, 5u), (double)ITERS * NACC);
  report("v_mad_u64_u32", time_kernel(k_mad64, grid, block, (u32*)out, 3u, 5u), (double)ITERS * NACC);
  report("v_fma_f64", time_kernel(k_fma64, grid, block, (u32*)out, 3u, 5u), (double)ITERS * NACC);
  u64 p = 0x3FFFFFFFFFFC0001ull, W = 0x123456789ABCDEFull, Wp = 0x48D159E26AF37BCull;
  double nb = (double)(ITERS / 4) * NACC;
  report("bfly_v0", time_kernel(k_bfly_v0, grid, block, (u64*)out, W, Wp, p), nb);
  report("bfly_v1", time_kernel(k_bfly_v1, grid, block, (u64*)out, W, Wp, p), nb);
  report("bfly_v2", time_kernel(k_bfly_v2, grid, block, (u64*)out, W, Wp, p), nb);
  report("bfly_v3", time_kernel(k_bfly_v3, grid, block, (u64*)out, W, Wp, p), nb);
  hipFree(out);
  return 0;
}

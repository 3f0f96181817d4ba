// Microbenchmark: per-wave issue cost of the VALU instruction kinds the modular butterflies are made of (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o instrate instrate.hip && ./instrate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned long long u64;
typedef unsigned int u32;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
constexpr int ITERS = 2048, NACC = 8;

#define K32(name, ASM, ...)                                                           \
  __global__ void name(u32 *out, u32 a, u32 b) {                                      \
    u32 x[NACC];                                                                      \
    for (int i = 0; i < NACC; ++i) x[i] = a + threadIdx.x + i;                        \
    u32 bb = b ^ threadIdx.x;                                                         \
    for (int it = 0; it < ITERS; ++it) {                                              \
      _Pragma("unroll") for (int i = 0; i < NACC; ++i) asm volatile(ASM : "+v"(x[i]) : "v"(bb) : __VA_ARGS__); \
    }                                                                                 \
    u32 s = 0; for (int i = 0; i < NACC; ++i) s += x[i];                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                   \
  }
#define K64(name, ASM, ...)                                                           \
  __global__ void name(u32 *out, u32 a, u32 b) {                                      \
    u64 x[NACC];                                                                      \
    for (int i = 0; i < NACC; ++i) x[i] = a + threadIdx.x + i;                        \
    u64 bb = ((u64)b << 32) | (a ^ threadIdx.x);                                      \
    u32 b32 = b ^ threadIdx.x;                                                        \
    for (int it = 0; it < ITERS; ++it) {                                              \
      _Pragma("unroll") for (int i = 0; i < NACC; ++i) asm volatile(ASM : "+v"(x[i]) : "v"(bb), "v"(b32) : __VA_ARGS__); \
    }                                                                                 \
    u64 s = 0; for (int i = 0; i < NACC; ++i) s += x[i];                              \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)(s ^ (s >> 32));                \
  }

K32(k_add32, "v_add_u32 %0, %0, %1", "memory")
K32(k_not32, "v_not_b32 %0, %0", "memory")
K32(k_mov32, "v_mov_b32 %0, %1", "memory")
K32(k_xor3, "v_xad_u32 %0, %0, %1, %1", "memory")
K32(k_addco, "v_add_co_u32 %0, vcc, %0, %1", "vcc")
K32(k_addco_e64, "v_add_co_u32_e64 %0, s[20:21], %0, %1", "s20", "s21")
K32(k_addc, "v_addc_co_u32 %0, vcc, %0, %1, vcc", "vcc")
K32(k_cnd, "v_cndmask_b32 %0, %0, %1, vcc", "memory")
K32(k_cnd_e64, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]", "memory")
K32(k_cmp32, "v_cmp_gt_u32 vcc, %0, %1", "vcc")
K32(k_mullo, "v_mul_lo_u32 %0, %0, %1", "memory")
K32(k_mulhi, "v_mul_hi_u32 %0, %0, %1", "memory")
K32(k_mul24, "v_mul_u32_u24 %0, %0, %1", "memory")
K32(k_add3, "v_add3_u32 %0, %0, %1, %1", "memory")
K64(k_add64, "v_lshl_add_u64 %0, %0, 0, %1", "memory")
K64(k_cmp64, "v_cmp_gt_u64 vcc, %0, %1", "vcc")
K64(k_mov64, "v_mov_b64 %0, %1", "memory")
K64(k_mad, "v_mad_u64_u32 %0, s[20:21], %2, %2, %0", "s20", "s21")
K64(k_mad_lo, "v_mad_u64_u32 %0, s[20:21], %2, %2, 0", "s20", "s21")

typedef void (*kern_t)(u32 *, u32, u32);
struct Ent { const char *name; kern_t k; int per; };

int main() {
  Ent ents[] = {{"v_add_u32", k_add32, 1}, {"v_not_b32", k_not32, 1}, {"v_mov_b32", k_mov32, 1}, {"v_xad_u32", k_xor3, 1},
                {"v_add3_u32", k_add3, 1},
                {"v_add_co_u32 vcc", k_addco, 1}, {"v_add_co_u32_e64", k_addco_e64, 1}, {"v_addc_co_u32 vcc", k_addc, 1},
                {"v_cndmask vcc", k_cnd, 1}, {"v_cndmask_e64 sgpr", k_cnd_e64, 1}, {"v_cmp_gt_u32", k_cmp32, 1},
                {"v_mul_lo_u32", k_mullo, 1}, {"v_mul_hi_u32", k_mulhi, 1}, {"v_mul_u32_u24", k_mul24, 1},
                {"v_lshl_add_u64", k_add64, 1}, {"v_cmp_gt_u64", k_cmp64, 1}, {"v_mov_b64", k_mov64, 1},
                {"v_mad_u64_u32 (+acc)", k_mad, 1}, {"v_mad_u64_u32 (+0)", k_mad_lo, 1}};
  u32 *out;
  CHECK(hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(u32)));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const double clk = 2.3e9;  // tools/ubench/clock.hip: sustained clock under VALU load
  printf("%-24s %10s %10s %10s\n", "instruction", "w/SIMD=1", "w/SIMD=4", "w/SIMD=8");
  for (auto &e : ents) {
    printf("%-24s", e.name);
    for (int wps : {1, 4, 8}) {
      const int blocks = 256 * wps;  // 256 threads = 4 waves = one per SIMD
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        e.k<<<blocks, 256>>>(out, 1, 2);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
      }
      const double instr_per_simd = (double)ITERS * NACC * wps;
      printf(" %10.2f", ms * 1e-3 * clk / instr_per_simd);
    }
    printf("\n");
  }
  return 0;
}

// Microbenchmark 2: issue cost of more VALU instruction kinds on gfx950, alone and mixed with v_mad_u64_u32.
//   hipcc --offload-arch=gfx950 -O3 -o instrate2 instrate2.hip && ./instrate2
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
// mixed: one mad on a 64-bit chain + N cheap ops on a 32-bit chain per slot
#define KMIX(name, ASM64, ASM32, ...)                                                 \
  __global__ void name(u32 *out, u32 a, u32 b) {                                      \
    u64 x[NACC]; u32 y[NACC];                                                         \
    for (int i = 0; i < NACC; ++i) { x[i] = a + threadIdx.x + i; y[i] = b + i; }      \
    u64 bb = ((u64)b << 32) | (a ^ threadIdx.x);                                      \
    u32 b32 = b ^ threadIdx.x;                                                        \
    for (int it = 0; it < ITERS; ++it) {                                              \
      _Pragma("unroll") for (int i = 0; i < NACC; ++i) {                              \
        asm volatile(ASM64 : "+v"(x[i]) : "v"(bb), "v"(b32) : __VA_ARGS__);           \
        asm volatile(ASM32 : "+v"(y[i]) : "v"(b32) : __VA_ARGS__);                    \
      }                                                                               \
    }                                                                                 \
    u64 s = 0; for (int i = 0; i < NACC; ++i) s += x[i] + y[i];                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (u32)(s ^ (s >> 32));                \
  }

K32(k_add32, "v_add_u32 %0, %0, %1", "memory")
K32(k_sub32, "v_sub_u32 %0, %0, %1", "memory")
K32(k_and32, "v_and_b32 %0, %0, %1", "memory")
K32(k_min32, "v_min_u32 %0, %0, %1", "memory")
K32(k_max32, "v_max_u32 %0, %0, %1", "memory")
K32(k_lshl32, "v_lshlrev_b32 %0, 3, %0", "memory")
K32(k_not32, "v_not_b32 %0, %0", "memory")
K32(k_mov32, "v_mov_b32 %0, %1", "memory")
K32(k_cnd_e32, "v_cndmask_b32 %0, %0, %1, vcc", "memory")
K32(k_cnd_e64, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]", "memory")
K32(k_cmp32, "v_cmp_gt_u32 vcc, %0, %1", "vcc")
K32(k_cmp32_e64, "v_cmp_gt_u32_e64 s[20:21], %0, %1", "s20", "s21")
K32(k_addco, "v_add_co_u32 %0, vcc, %0, %1", "vcc")
K32(k_addc, "v_addc_co_u32 %0, vcc, %0, %1, vcc", "vcc")
K32(k_subco, "v_sub_co_u32 %0, vcc, %0, %1", "vcc")
K32(k_subb, "v_subb_co_u32 %0, vcc, %0, %1, vcc", "vcc")
K32(k_add3, "v_add3_u32 %0, %0, %1, %1", "memory")
K32(k_lshladd, "v_lshl_add_u32 %0, %0, 1, %1", "memory")
K32(k_andor, "v_and_or_b32 %0, %0, %1, %1", "memory")
K32(k_bfe, "v_bfe_u32 %0, %0, 3, 8", "memory")
K32(k_alignbit, "v_alignbit_b32 %0, %0, %1, 7", "memory")
K32(k_perm, "v_perm_b32 %0, %0, %1, %1", "memory")
K32(k_mad24, "v_mad_u32_u24 %0, %0, %1, %1", "memory")
K32(k_mul24, "v_mul_u32_u24 %0, %0, %1", "memory")
K32(k_mulhi24, "v_mul_hi_u32_u24 %0, %0, %1", "memory")
K32(k_mullo, "v_mul_lo_u32 %0, %0, %1", "memory")
K32(k_mulhi, "v_mul_hi_u32 %0, %0, %1", "memory")
K32(k_fma32, "v_fma_f32 %0, %0, %1, %1", "memory")
K32(k_movdpp, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "memory")
K32(k_movdpp_ror, "v_mov_b32_dpp %0, %0 row_ror:8 row_mask:0xf bank_mask:0xf", "memory")
K32(k_adddpp, "v_add_u32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "memory")
K32(k_cnddpp, "v_cndmask_b32_dpp %0, %0, %1, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "memory")
K32(k_plswap32, "v_permlane32_swap_b32 %0, %1", "memory")
K32(k_plswap16, "v_permlane16_swap_b32 %0, %1", "memory")
K32(k_sdwa, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD", "memory")
K64(k_add64, "v_lshl_add_u64 %0, %0, 0, %1", "memory")
K64(k_cmp64, "v_cmp_gt_u64 vcc, %0, %1", "vcc")
K64(k_mov64, "v_mov_b64 %0, %1", "memory")
K64(k_lshl64, "v_lshlrev_b64 %0, 3, %0", "memory")
K64(k_lshr64, "v_lshrrev_b64 %0, 3, %0", "memory")
K64(k_mad, "v_mad_u64_u32 %0, s[20:21], %2, %2, %0", "s20", "s21")
K64(k_mad_i, "v_mad_i64_i32 %0, s[20:21], %2, %2, %0", "s20", "s21")
K64(k_mad_lo, "v_mad_u64_u32 %0, s[20:21], %2, %2, 0", "s20", "s21")
K64(k_fma64, "v_fma_f64 %0, %0, %1, %1", "memory")
K64(k_add_f64, "v_add_f64 %0, %0, %1", "memory")
K64(k_pk_fma32, "v_pk_fma_f32 %0, %0, %1, %1", "memory")
K64(k_pk_add32, "v_pk_add_f32 %0, %0, %1", "memory")
KMIX(m_mad_add32, "v_mad_u64_u32 %0, s[20:21], %2, %2, %0", "v_add_u32 %0, %0, %1", "s20", "s21")
KMIX(m_mad_cnd, "v_mad_u64_u32 %0, s[20:21], %2, %2, %0", "v_cndmask_b32 %0, %0, %1, vcc", "s20", "s21")
KMIX(m_mad_addco, "v_mad_u64_u32 %0, s[20:21], %2, %2, %0", "v_add_co_u32 %0, vcc, %0, %1", "s20", "s21", "vcc")
KMIX(m_mad_add2, "v_mad_u64_u32 %0, s[20:21], %2, %2, %0", "v_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1", "s20", "s21")
KMIX(m_add64_add32, "v_lshl_add_u64 %0, %0, 0, %1", "v_add_u32 %0, %0, %1", "s20", "s21")

typedef void (*kern_t)(u32 *, u32, u32);
struct Ent { const char *name; kern_t k; int per; };

int main() {
  Ent ents[] = {{"v_add_u32", k_add32, 1}, {"v_sub_u32", k_sub32, 1}, {"v_and_b32", k_and32, 1}, {"v_min_u32", k_min32, 1}, {"v_max_u32", k_max32, 1},
                {"v_lshlrev_b32", k_lshl32, 1}, {"v_not_b32", k_not32, 1}, {"v_mov_b32", k_mov32, 1},
                {"v_cndmask_e32 vcc", k_cnd_e32, 1}, {"v_cndmask_e64 sgpr", k_cnd_e64, 1}, {"v_cmp_gt_u32 vcc", k_cmp32, 1}, {"v_cmp_gt_u32_e64", k_cmp32_e64, 1},
                {"v_add_co_u32", k_addco, 1}, {"v_addc_co_u32", k_addc, 1}, {"v_sub_co_u32", k_subco, 1}, {"v_subb_co_u32", k_subb, 1},
                {"v_add3_u32", k_add3, 1}, {"v_lshl_add_u32", k_lshladd, 1}, {"v_and_or_b32", k_andor, 1}, {"v_bfe_u32", k_bfe, 1}, {"v_alignbit_b32", k_alignbit, 1},
                {"v_perm_b32", k_perm, 1}, {"v_mad_u32_u24", k_mad24, 1}, {"v_mul_u32_u24", k_mul24, 1}, {"v_mul_hi_u32_u24", k_mulhi24, 1},
                {"v_mul_lo_u32", k_mullo, 1}, {"v_mul_hi_u32", k_mulhi, 1}, {"v_fma_f32", k_fma32, 1},
                {"v_mov_b32_dpp quad", k_movdpp, 1}, {"v_mov_b32_dpp row_ror", k_movdpp_ror, 1}, {"v_add_u32_dpp", k_adddpp, 1}, {"v_cndmask_b32_dpp", k_cnddpp, 1},
                {"v_permlane32_swap", k_plswap32, 1}, {"v_permlane16_swap", k_plswap16, 1}, {"v_add_u32_sdwa", k_sdwa, 1},
                {"v_lshl_add_u64", k_add64, 1}, {"v_cmp_gt_u64", k_cmp64, 1}, {"v_mov_b64", k_mov64, 1}, {"v_lshlrev_b64", k_lshl64, 1}, {"v_lshrrev_b64", k_lshr64, 1},
                {"v_mad_u64_u32 (+acc)", k_mad, 1}, {"v_mad_i64_i32 (+acc)", k_mad_i, 1}, {"v_mad_u64_u32 (+0)", k_mad_lo, 1},
                {"v_fma_f64", k_fma64, 1}, {"v_add_f64", k_add_f64, 1}, {"v_pk_fma_f32", k_pk_fma32, 1}, {"v_pk_add_f32", k_pk_add32, 1},
                {"MIX mad + add32 (pair)", m_mad_add32, 1}, {"MIX mad + cndmask (pair)", m_mad_cnd, 1}, {"MIX mad + add_co (pair)", m_mad_addco, 1},
                {"MIX mad + 2 add32 (triple)", m_mad_add2, 1}, {"MIX add64 + add32 (pair)", m_add64_add32, 1}};
  u32 *out;
  CHECK(hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(u32)));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const double clk = 2.3e9;  // tools/ubench/clock.hip: sustained clock under VALU load
  printf("%-28s %10s %10s %10s %10s   (cycles per wave-instruction or per group, at an assumed 2.3 GHz)\n", "instruction", "w/SIMD=1", "w/SIMD=2", "w/SIMD=4", "w/SIMD=8");
  for (auto &e : ents) {
    printf("%-28s", e.name);
    for (int wps : {1, 2, 4, 8}) {
      const int blocks = 256 * wps;  // 256 threads = 4 waves = one per SIMD
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0));
        e.k<<<blocks, 256>>>(out, 1, 2);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
      }
      const double instr_per_simd = (double)ITERS * NACC * wps;
      printf(" %10.2f", ms * 1e-3 * clk / instr_per_simd);
    }
    printf("\n");
    fflush(stdout);
  }
  return 0;
}

// bconv_mfma_kernels.h — fast basis conversion on the matrix cores of gfx950.
//
// internal::do_approximated_basis_conversion (reference src/base_converter.cpp:313-349) is a matrix product over the integers,
//   T[coefficient][target] = sum_k digit_k[coefficient] * hat[target][k],      digit_k = [x_k * inv_k]_{src_k} < 2^62, hat < 2^62,
// followed by one reduction modulo the target prime.  bconv_kernel (pointwise_kernels.h) runs the product on the vector ALU: six
// v_mad_u64_u32 per term, 180 of the ~250 instructions per (coefficient, target) at 30 source limbs, and is bound by VALU issue at
// 0.22 of the HBM roofline.  Here the product runs on v_mfma_i32_16x16x64_i8 (16 cycles per 16 384 multiply-adds, a unit the
// NTT kernels leave idle) and only the reduction stays on the vector ALU.
//
// Byte decomposition: every 64-bit operand v < 2^63 is written in balanced base 256, v = sum_i s_i 256^i with s_i in [-128, 127]
// (the bytes of (v + C) ^ C, C = 0x8080808080808080).  Then T = sum_s S_s 256^s with the 15 byte-convolution sums
// S_s = sum_k sum_{i+j=s} digit_{k,i} hat_{k,j}, |S_s| <= 8 * 32 * 2^14 = 2^22 for up to 32 source limbs.
//
// MFMA tile (A: 16 rows x 64 k, B: 64 k x 16 columns, C: lane l holds rows 4(l>>4) + r, r = 0..3, of column l & 15):
//   column  = coefficient (16 per tile);
//   k slot  = (source limb, byte i of its digit): lane l of the B operand holds 16 bytes = the digits of limbs 8 st + 2(l>>4) and
//             8 st + 2(l>>4) + 1 of its coefficient, i.e. two plain 64-bit loads (st = k-step, 8 limbs per MFMA);
//   row     = 4 g + r  <->  (target 4 grp + g, byte weight s = 4 a + r), a = 0..3 numbering four tiles per group of four targets.
// The A operand is the Toeplitz image of the hat bytes (hat_{k, s - i}, zero outside 0..7; built once per conversion table by
// bconv_table_kernel, read from LDS).  After the four tiles of a group, lane (column c, g) holds all sixteen S_s of
// (coefficient c, target 4 grp + g): no cross-lane step.  The accumulators start at 2^22, so every S is a non-negative 24-bit
// number and T + Bias (Bias = 2^22 (2^128 - 1) / 255) is put together with 32-bit shifts-and-adds and one multiply-add per four sums,
// taken through two 64-bit Montgomery reductions (the hat tables carry the factor 2^128) and corrected by Bias * 2^-128 mod p.
// ~50 VALU instructions per (coefficient, target) instead of ~250; half of the MFMA rows x k-slots are structural zeros of the
// Toeplitz image, which the otherwise idle unit absorbs.
//
// Used when the digits are ready-made (the rotation pipelines: the producing inverse NTT carries inv_k in its scaling) and the
// conversion has at most 32 source limbs; everything else runs bconv_kernel.
#pragma once
#include "pointwise_kernels.h"

typedef int v4i __attribute__((ext_vector_type(4)));

#define BCONV_MFMA_THREADS 512          // 8 waves: one persistent workgroup per CU (the table fills most of its LDS), 2 waves per SIMD
#define BCONV_MFMA_BLOCK 512            // coefficients per workgroup iteration (64 per wave = four 16-column tiles)
#define BCONV_MFMA_ACC0 (1 << 22)       // accumulator bias: |S_s| <= 2^22

struct BconvMfmaArgs {
  u64 *out; const u64 *src;
  const uint4 *table;     // [ngroups][4][NS][64] A fragments (16 bytes per lane)
  const PrimeC *primes;
  const u64 *bias;        // [M]  Bias * 2^-128 mod p_m, canonical
  u32 nsrc, src_limbs, src_off;
  u32 tgt_first, out_limbs, out_off, ntgt, ngroups;
  u32 nblocks, blocks_per_entry;
  u64 N;
};

__device__ __forceinline__ u64 balanced_bytes(u64 v) {
  const u64 C = 0x8080808080808080ull;
  return (v + C) ^ C;
}

// A fragments of one conversion table: hat_m = [ntgt][hat_stride] (times 2^128 mod the target prime, canonical)
__global__ void bconv_table_kernel(uint4 *table, const u64 *hat_m, u32 hat_stride, u32 nsrc, u32 ntgt, u32 NS, u32 ngroups) {
  const u32 idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= ngroups * 4 * NS * 64) return;
  const u32 lane = idx & 63, f = idx >> 6, st = f % NS, a4 = (f / NS) & 3, grp = f / NS / 4;
  const u32 m = lane & 15, kg = lane >> 4;
  const u32 t = grp * 4 + (m >> 2);
  const int s = (int)(4 * a4 + (m & 3));
  u32 w[4] = {0, 0, 0, 0};
#pragma unroll
  for (int b = 0; b < 16; ++b) {
    const u32 limb = 8 * st + 2 * kg + (b >> 3);
    const int j = s - (b & 7);
    u32 v = 0;
    if (t < ntgt && limb < nsrc && j >= 0 && j <= 7) v = (u32)(balanced_bytes(hat_m[(u64)t * hat_stride + limb]) >> (8 * j)) & 0xFFu;
    w[b >> 2] |= v << (8 * (b & 3));
  }
  table[idx] = make_uint4(w[0], w[1], w[2], w[3]);
}

// sixteen biased byte-convolution sums (four accumulator tiles) of one (coefficient, target) -> the canonical residue
template <bool SP>
__device__ __forceinline__ u64 bconv_mfma_fold(const v4i &c0, const v4i &c1, const v4i &c2, const v4i &c3, const PrimeC &PT, u64 biasc) {
  u32 w[5];
  u32 carry = 0;
  const v4i *cc[4] = {&c0, &c1, &c2, &c3};
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const u32 s0 = (u32)(*cc[a])[0], s1 = (u32)(*cc[a])[1], s2 = (u32)(*cc[a])[2], s3 = (u32)(*cc[a])[3];
    const u32 z = (s1 << 8) + s0 + carry;   // < 2^31 + 2^23 + 2^17
    const u32 y = (s3 << 8) + s2;
    const u64 X = mad_wide(y, 65536u, (u64)z);
    w[a] = (u32)X;
    carry = (u32)(X >> 32);                 // < 2^17
  }
  w[4] = carry;
  const u64 X0 = pair64(w[0], w[1]), X1 = pair64(w[2], w[3]);
  const u64 r1 = mont_reduce<SP>(0, X0, PT);                  // X0 * 2^-64 (mod p), in (0, p]
  const u64 w_lo = X1 + r1;
  const u64 w_hi = (u64)w[4] + (w_lo < r1 ? 1u : 0u);
  const u64 r = csub(mont_reduce<SP>(w_hi, w_lo, PT), PT.p);  // (T + Bias) * 2^-128 mod p, canonical
  return r >= biasc ? r - biasc : r + PT.p - biasc;
}

template <bool SP, int NS>
__global__ __launch_bounds__(BCONV_MFMA_THREADS) void bconv_mfma_kernel(BconvMfmaArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bconv_smem[];
  uint4 *tab = reinterpret_cast<uint4 *>(bconv_smem);
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, g = lane >> 4;
  const u32 nfrag = a.ngroups * 4 * NS * 64;
  for (u32 i = tid; i < nfrag; i += BCONV_MFMA_THREADS) tab[i] = a.table[i];
  __syncthreads();
  const u32 last = a.nsrc - 1;
  for (u32 blk = blockIdx.x; blk < a.nblocks; blk += gridDim.x) {   // bounded: every wave leaves after its last block
    const u32 e = blk / a.blocks_per_entry;
    const u64 d = (u64)(blk % a.blocks_per_entry) * BCONV_MFMA_BLOCK + wave * 64 + col;
    const u64 *sp = a.src + ((u64)e * a.src_limbs + a.src_off) * a.N + d;
    v4i B[4][NS];
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      const u32 k0 = 8 * st + 2 * g, k1 = k0 + 1;     // limbs beyond the source basis meet zero rows of the table: any valid address will do
      const u64 *r0 = sp + (u64)(k0 < last ? k0 : last) * a.N, *r1 = sp + (u64)(k1 < last ? k1 : last) * a.N;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const u64 x0 = balanced_bytes(r0[16 * ct]), x1 = balanced_bytes(r1[16 * ct]);
        B[ct][st] = v4i{(int)(u32)x0, (int)(u32)(x0 >> 32), (int)(u32)x1, (int)(u32)(x1 >> 32)};
      }
    }
    for (u32 grp = 0; grp < a.ngroups; ++grp) {
      v4i acc[4][4];
#pragma unroll
      for (int a4 = 0; a4 < 4; ++a4)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[a4][ct] = v4i{BCONV_MFMA_ACC0, BCONV_MFMA_ACC0, BCONV_MFMA_ACC0, BCONV_MFMA_ACC0};
      const uint4 *tg = tab + (u64)grp * 4 * NS * 64 + lane;
#pragma unroll
      for (int st = 0; st < NS; ++st)
#pragma unroll
        for (int a4 = 0; a4 < 4; ++a4) {
          const uint4 f = tg[(a4 * NS + st) * 64];
          const v4i A = v4i{(int)f.x, (int)f.y, (int)f.z, (int)f.w};
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) acc[a4][ct] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B[ct][st], acc[a4][ct], 0, 0, 0);
        }
      const u32 t = grp * 4 + g, tc = t < a.ntgt ? t : a.ntgt - 1;
      const PrimeC PT = a.primes[a.tgt_first + tc];
      const u64 biasc = a.bias[a.tgt_first + tc];
      u64 *op = a.out + ((u64)e * a.out_limbs + a.out_off + tc) * a.N + d;
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const u64 r = bconv_mfma_fold<SP>(acc[0][ct], acc[1][ct], acc[2][ct], acc[3][ct], PT, biasc);
        DBG_RANGE(r, PT.p, 13);
        if (t < a.ntgt) op[16 * ct] = r;
      }
    }
  }
}

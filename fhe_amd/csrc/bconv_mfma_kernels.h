// bconv_mfma_kernels.h — fast basis conversion on the matrix cores of gfx950.
//
// internal::do_approximated_basis_conversion (reference src/base_converter.cpp:313-349) is a matrix product,
//   T[coefficient][target] = sum_k digit_k[coefficient] * hat[target][k]  (mod p_target),   digit_k = [x_k * inv_k]_{src_k} < 2^62,
// followed by one reduction modulo the target prime.  bconv_kernel (pointwise_kernels.h) runs the product on the vector ALU: six
// v_mad_u64_u32 per term, 180 of the ~250 instructions per (coefficient, target) at 30 source limbs, and is bound by VALU issue at
// 0.22 of the HBM roofline.  Here the product runs on v_mfma_i32_16x16x64_i8 (16 cycles per 16 384 multiply-adds, a unit the
// NTT kernels leave idle) and only the reduction stays on the vector ALU.
//
// Byte decomposition: a digit d < 2^63 is written in balanced base 256, d = sum_i d_i 256^i with d_i in [-128, 127] (the bytes of
// (d + C) ^ C, C = 0x8080808080808080).  Modulo the target prime the weight of byte i moves into the constant:
//   T * 2^64 = sum_k sum_i d_{k,i} * c_{t,k,i}   (mod p_t),      c_{t,k,i} = hat[t][k] * 256^i * 2^64 mod p_t   (< 2^62),
// and with c in balanced bytes too, c = sum_j c_j 256^j, the integer U = sum_j 256^j S_j with the eight sums
//   S_j = sum_{k,i} d_{k,i} c_{t,k,i,j},    |S_j| <= 8 * 32 * 2^14 = 2^22 for up to 32 source limbs,
// is congruent to T * 2^64: a DENSE int8 product with K = 8 bytes x source limbs and 8 rows per target (a convolution over
// byte weights i + j would need 15 rows per target, half of them structural zeros).
//
// MFMA tile (A: 16 rows x 64 k, B: 64 k x 16 columns, C: lane l holds rows 4(l>>4) + r, r = 0..3, of column l & 15):
//   column  = coefficient (16 per tile);
//   k slot  = (source limb, byte i of its digit): lane l of the B operand holds 16 bytes = the digits of limbs 8 st + 2(l>>4) and
//             8 st + 2(l>>4) + 1 of its coefficient, i.e. two plain 64-bit loads (st = k-step, 8 limbs per MFMA);
//   row     = 4 g + r  <->  (target 4 grp + g, byte j = 4 a + r), a = 0, 1 numbering two tiles per group of four targets.
// The A operand (bytes of c, built on the host once per conversion table) is read from LDS.  After the two tiles of a group,
// lane (column c, g) holds all eight S_j of (coefficient c, target 4 grp + g): no cross-lane step.  The accumulators start at
// 2^22 + (a byte of D), so every S is a non-negative 24-bit number, where D = -2^22 (2^64 - 1)/255 mod p is chosen per target so
// that the whole offset is a multiple of the target prime and never has to be taken out again.  U + offset (< 2^81) is put
// together with 32-bit shift-adds, one funnel shift and one carry chain (10 instructions) and taken through ONE 64-bit Montgomery
// reduction: ~25 VALU instructions per (coefficient, target) and 2 NS MFMAs per 16 coefficients x 4 targets instead of ~250 VALU instructions.
// MFMA and VALU work of one SIMD overlap only in part (tools/ubench/mfma_valu.hip: a wave of MFMAs leaves the vector ALU ~40 % of
// its issue slots), so both counts matter.
//
// Used when the conversion has at most 32 source limbs and the launch is large enough for persistent workgroups; everything
// else runs bconv_kernel.  PRE: the digits arrive ready-made (the NTT-domain rotation pipeline: the producing inverse NTT carries
// inv_k in its scaling); otherwise each lane forms the digits of the limbs it loads, [x_k * inv_k] mod src_k, once per block.
#pragma once
#include "pointwise_kernels.h"

typedef int v4i __attribute__((ext_vector_type(4)));

#define BCONV_MFMA_THREADS 512          // 8 waves: one persistent workgroup per CU (the table fills most of its LDS), 2 waves per SIMD
#define BCONV_MFMA_BLOCK 512            // coefficients per workgroup iteration (64 per wave = four 16-column tiles)
#define BCONV_MFMA_ACC0 (1 << 22)       // accumulator bias: |S_s| <= 2^22

// per-prime constants of a conversion target (one 48-byte record, read by the lanes that fold for it)
struct BconvTgt {
  u64 p, pinv;
  u32 init[8];   // 2^22 + byte j of D: accumulator start values
};

static_assert(sizeof(BconvTgt) == 48, "the kernel copies three uint4 per record");

struct BconvMfmaArgs {
  u64 *out; const u64 *src;
  const uint4 *table;     // [ngroups][NS][2][64] A fragments (16 bytes per lane) in the order the kernel consumes them (host-built: fhe_hip.hip)
  const BconvTgt *tgt;    // [M]
  const PrimeC *primes;   // [M]      (digits formed in the kernel: source primes src_first + k)
  const u64 *inv_m;       // [nsrc]   (digits formed in the kernel) Montgomery form w.r.t. the source prime
  u32 src_first;
  u32 nsrc, src_limbs, src_off;
  u32 tgt_first, out_limbs, out_off, ntgt, ngroups;
  u32 nblocks, blocks_per_entry;
  u64 N;
};

__device__ __forceinline__ u64 balanced_bytes(u64 v) {
  const u64 C = 0x8080808080808080ull;
  return (v + C) ^ C;
}

// Fold of the eight offset byte sums of one (coefficient, target) into the canonical residue.  Word a of U collects
// Z_a = S_4a + 2^8 S_4a+1 (one 32-bit shift-add: both < 2^23 + 2^8) and, 16 bits up, Y_a = S_4a+2 + 2^8 S_4a+3, whose upper half
// spills into word a + 1 (one funnel shift); one carry chain adds them.  U * 2^-64 mod p is the conversion's result.
template <bool SP>
__device__ __forceinline__ u64 bconv_mfma_fold(const v4i &c0, const v4i &c1, u64 p, u64 pinv) {
  const u32 z0 = ((u32)c0[1] << 8) + (u32)c0[0], y0 = ((u32)c0[3] << 8) + (u32)c0[2];
  const u32 z1 = ((u32)c1[1] << 8) + (u32)c1[0], y1 = ((u32)c1[3] << 8) + (u32)c1[2];
  u32 cy;
  const u32 w0 = __builtin_addc(z0, y0 << 16, 0u, &cy);
  const u32 w1 = __builtin_addc(z1, __builtin_amdgcn_alignbit(y1, y0, 16), cy, &cy);
  const u32 w2 = (y1 >> 16) + cy;                                  // < 2^17
  PrimeC PT;
  PT.p = p; PT.pinv = pinv; PT.p1 = (u32)(p >> 32);
  return csub(mont_reduce<SP>((u64)w2, pair64(w0, w1), PT), p);    // (w2, p + w2] -> [0, p)
}

template <bool SP, int NS, bool PRE>
__global__ __launch_bounds__(BCONV_MFMA_THREADS) void bconv_mfma_kernel(BconvMfmaArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char bconv_smem[];
  uint4 *tab = reinterpret_cast<uint4 *>(bconv_smem);
  const u32 tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, g = lane >> 4;
  const u32 nfrag = a.ngroups * 2 * NS * 64;
  for (u32 i = tid; i < nfrag; i += BCONV_MFMA_THREADS) tab[i] = a.table[i];
  // the targets' constants behind the table (3 uint4 per record).  Inside the block loop nothing but the digits is LOADED from
  // global memory: vector-memory operations complete in order, so waiting for any younger load would wait for the prefetched
  // digits of the next block as well.
  uint4 *tconst = tab + nfrag;
  for (u32 i = tid; i < a.ntgt * 3; i += BCONV_MFMA_THREADS) tconst[i] = reinterpret_cast<const uint4 *>(a.tgt + a.tgt_first)[i];
  __syncthreads();
  const u32 last = a.nsrc - 1;
  // A fragments: one cyclic stream (group, k-step, tile), PH per group, held in a ring of R register sets: the fragment R
  // positions ahead is requested right after the multiply-adds of the current one were issued, so an LDS read has >= 16 MFMAs
  // to arrive.  The stream wraps from the last group to the first.
  constexpr int PH = 2 * NS, R = (PH % 4 == 0) ? 4 : PH;
  const uint4 *tq = tab + lane;
  const u32 nq = a.ngroups * PH * 64;            // table length in uint4
  u32 qc = 0, qx = PH * 64 < nq ? PH * 64 : 0;   // offsets of the current and the next group
  uint4 fr[R];
#pragma unroll
  for (int n = 0; n < R; ++n) fr[n] = tq[n * 64];
  // digits of the block as B operands: B[ct][st] = limbs 8 st + 2 g, + 1 of coefficient 32 (ct >> 1) + 2 col + (ct & 1) of the wave's
  // 64 (a lane's pair of adjacent coefficients comes from one 16-byte load and leaves in one 16-byte store).  The NEXT block's are
  // requested (raw, into a second register set) as soon as the current block's have been turned into balanced bytes, so the
  // loads have a whole block of MFMAs and folds to arrive: one persistent workgroup per CU has no other waves to hide them.
  // Measured (round 3): with the MFMAs or the fold compiled out the kernel takes the same 0.98 ms per 4 GB at C4's shape, with
  // the LDS reads compiled out too: the kernel moves 4.1 TB/s and is bound by the memory system, not by either unit.
  // (limbs beyond the source basis meet zero rows of the table: any valid address will do)
  u32 kidx[NS][2];
#pragma unroll
  for (int st = 0; st < NS; ++st) {
    const u32 k0 = 8 * st + 2 * g, k1 = k0 + 1;
    kidx[st][0] = k0 < last ? k0 : last; kidx[st][1] = k1 < last ? k1 : last;
  }
  // canonical digit [x * inv_k] mod src_k with lane-dependent constants (each lane owns its own limbs)
  auto digit = [&](u64 x, u32 k) {
    if constexpr (PRE) return x;
    else {
      const PrimeC &PS = a.primes[a.src_first + k];
      PrimeC P;
      P.p = PS.p; P.pinv = PS.pinv; P.p1 = (u32)(P.p >> 32); P.n32 = 0u - (u32)P.pinv; P.twop = 0; P.r2 = 0; P.ninv_m = 0;
      const u64 D = mont_diff<SP>(a.inv_m[k], x, P);   // in (-p, p), mod 2^64
      return (long long)D < 0 ? D + P.p : D;
    }
  };
  auto block_src = [&](u32 blk) {
    const u32 e = blk / a.blocks_per_entry;
    return a.src + ((u64)e * a.src_limbs + a.src_off) * a.N + (u64)(blk % a.blocks_per_entry) * BCONV_MFMA_BLOCK + wave * 64 + 2 * col;
  };
  v4i B[4][NS];
  u64 Bn[4][NS][2];
  auto request = [&](const u64 *sp) {
#pragma unroll
    for (int st = 0; st < NS; ++st)
#pragma unroll
      for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const ulonglong2 v = ld2(sp + (u64)kidx[st][q] * a.N + 32 * pr);
          Bn[2 * pr][st][q] = v.x; Bn[2 * pr + 1][st][q] = v.y;
        }
  };
  if (blockIdx.x < a.nblocks) request(block_src(blockIdx.x));
  const BconvTgt *tp = reinterpret_cast<const BconvTgt *>(tconst);
  v4i acc[2][4];
  // a group's MFMAs: tiles 0, 1 over all k-steps
  auto group_mfma = [&](const v4i &c_lo, const v4i &c_hi) {
#pragma unroll
    for (int st = 0; st < NS; ++st) {
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int n = st * 2 + b;
        const uint4 f = fr[n % R];
        const v4i A = v4i{(int)f.x, (int)f.y, (int)f.z, (int)f.w};
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
          acc[b][ct] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B[ct][st], st == 0 ? (b == 0 ? c_lo : c_hi) : acc[b][ct], 0, 0, 0);
        const int m = n + R;
        fr[n % R] = m < PH ? tq[qc + m * 64] : tq[qx + (m - PH) * 64];
      }
      __builtin_amdgcn_sched_barrier(0);   // (the scheduler otherwise sinks the LDS reads to just before their use)
    }
    qc = qx; qx = qx + PH * 64; qx = qx < nq ? qx : 0;
  };

  for (u32 blk = blockIdx.x; blk < a.nblocks; blk += gridDim.x) {   // bounded: every wave leaves after its last block
    const u32 e = blk / a.blocks_per_entry;
    const u64 d = (u64)(blk % a.blocks_per_entry) * BCONV_MFMA_BLOCK + wave * 64 + 2 * col;
    const u32 nblk = blk + gridDim.x;
#pragma unroll
    for (int st = 0; st < NS; ++st)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const u64 x0 = balanced_bytes(digit(Bn[ct][st][0], kidx[st][0])), x1 = balanced_bytes(digit(Bn[ct][st][1], kidx[st][1]));
        B[ct][st] = v4i{(int)(u32)x0, (int)(u32)(x0 >> 32), (int)(u32)x1, (int)(u32)(x1 >> 32)};
      }
    __builtin_amdgcn_sched_barrier(0);
    if (nblk < a.nblocks) request(block_src(nblk));
    __builtin_amdgcn_sched_barrier(0);
    // target constants one group ahead (LDS)
    u32 tcn = g < a.ntgt ? g : a.ntgt - 1;
    u64 pn = tp[tcn].p, pinvn = tp[tcn].pinv;
    v4i i0n = *reinterpret_cast<const v4i *>(tp[tcn].init), i1n = *reinterpret_cast<const v4i *>(tp[tcn].init + 4);
    for (u32 grp = 0; grp < a.ngroups; ++grp) {
      const u32 t = grp * 4 + g, tc = tcn;
      const u64 p = pn, pinv = pinvn;
      const v4i i0 = i0n, i1 = i1n;
      {
        const u32 t2 = t + 4;
        tcn = t2 < a.ntgt ? t2 : a.ntgt - 1;
        pn = tp[tcn].p; pinvn = tp[tcn].pinv;
        i0n = *reinterpret_cast<const v4i *>(tp[tcn].init); i1n = *reinterpret_cast<const v4i *>(tp[tcn].init + 4);
      }
      group_mfma(i0, i1);
      u64 *op = a.out + ((u64)e * a.out_limbs + a.out_off + tc) * a.N + d;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const u64 r0 = bconv_mfma_fold<SP>(acc[0][2 * pr], acc[1][2 * pr], p, pinv), r1 = bconv_mfma_fold<SP>(acc[0][2 * pr + 1], acc[1][2 * pr + 1], p, pinv);
        if (t < a.ntgt) st2c(op + 32 * pr, r0, r1, p, 13);
      }
    }
  }
}

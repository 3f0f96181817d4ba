// pointwise_kernels.h — coalesced limb-major pointwise / base-conversion kernels for gfx950.
// Integer mat-vec and elementwise work: HBM-bound, no MFMA.  Every thread owns two adjacent
// coefficients (one 16-byte access per operand); grid = (N/512, limb, batch entry).
// Reference loops replaced are cited per kernel (paths relative to the reference root).
#pragma once
#include "modarith.h"

#define PW_THREADS 256

__device__ __forceinline__ ulonglong2 ld2(const u64 *p) { return *reinterpret_cast<const ulonglong2 *>(p); }
__device__ __forceinline__ void st2(u64 *p, u64 x, u64 y) { *reinterpret_cast<ulonglong2 *>(p) = make_ulonglong2(x, y); }
// store of two canonical residues (debug build: range-checked against the modulus, site = kernel)
__device__ __forceinline__ void st2c(u64 *p, u64 x, u64 y, [[maybe_unused]] u64 mod, [[maybe_unused]] unsigned site) {
  DBG_RANGE(x, mod, site); DBG_RANGE(y, mod, site);
  st2(p, x, y);
}
__device__ __forceinline__ u64 addmod_c(u64 a, u64 b, u64 p) { return csub(a + b, p); }

// Mul::cross_mult (src/evaluator.cpp:621-647): c0,c1 = [b][2][Li][N] -> out3 = [b][3][Li][N]
template <bool SP>
__global__ __launch_bounds__(PW_THREADS) void cross_mult_kernel(u64 *out3, const u64 *c0, const u64 *c1, const PrimeC *primes,
                                                                u32 Li, u64 N) {
  const u64 d = ((u64)blockIdx.x * PW_THREADS + threadIdx.x) * 2;
  const u32 l = blockIdx.y, b = blockIdx.z;
  const PrimeC P = primes[l];
  const u64 in = ((u64)b * 2 * Li + l) * N + d, half = (u64)Li * N;
  const ulonglong2 b0 = ld2(c0 + in), a0 = ld2(c0 + in + half), b1 = ld2(c1 + in), a1 = ld2(c1 + in + half);
  u64 r0[2], r1[2], r2[2];
  const u64 vb0[2] = {b0.x, b0.y}, va0[2] = {a0.x, a0.y}, vb1[2] = {b1.x, b1.y}, va1[2] = {a1.x, a1.y};
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    r0[i] = mulmod_full<SP>(vb0[i], vb1[i], P);
    r2[i] = mulmod_full<SP>(va0[i], va1[i], P);
    u64 hi = 0, lo = 0;  // b0*a1 + b1*a0 < 2p^2: one lazy 128-bit sum, one reduction (evaluator.cpp:636-642)
    mac128(hi, lo, vb0[i], va1[i]);
    mac128(hi, lo, vb1[i], va0[i]);
    const u64 t = mont_reduce<SP>(hi, lo, P);
    r1[i] = csub(mont_mul<SP>(P.r2, t, P), P.p);
  }
  const u64 o = ((u64)b * 3 * Li + l) * N + d;
  st2c(out3 + o, r0[0], r0[1], P.p, 14);
  st2c(out3 + o + half, r1[0], r1[1], P.p, 14);
  st2c(out3 + o + 2 * half, r2[0], r2[1], P.p, 14);
}

// internal::do_approximated_basis_conversion (src/base_converter.cpp:313-349), all targets of one
// conversion: out[t][d] = ( sum_k [x_k[d] * inv_k]_{src_k} * hat[t][k] ) mod tgt_t   (uncorrected).
// src = [e][src_limbs][N] (limbs src_off .. src_off+nsrc), out = [e][out_limbs][N] (limbs out_off + t).
struct BconvArgs {
  u64 *out; const u64 *src;
  const u64 *inv_m;   // [nsrc]  Montgomery form w.r.t. the source prime
  const u64 *hat_m;   // [ntgt][hat_stride]  times 2^128 mod the target prime (two Montgomery factors: see bconv_fold)
  const PrimeC *primes;
  u32 nsrc, src_first, src_limbs, src_off;
  u32 tgt_first, out_limbs, out_off, hat_stride, ntgt;
  u64 N;
};
// One thread = one coefficient x BCONV_TC targets.  Per source limb k the canonical digit t0 = [x_k * inv_k]_{p_k}
// is computed once.  Every VALU instruction costs one issue slot on gfx950, so the sum
// sum_k hat[t][k] * t0_k is accumulated WITHOUT carries: hat (< 2^62, uniform, split by the scalar unit into three
// 21-bit limbs) times t0 (two 32-bit words) gives six partial products < 2^53, each added into its own 64-bit
// accumulator by one v_mad_u64_u32 (room for 2^8 terms).  The six accumulators are folded once per target at the end
// (bconv_fold).
#define BCONV_TC 5
// Fold of the six carry-free accumulators A_w (weights 2^w, w = 0, 21, 42 | 32, 53, 74; each < 2^61): the weighted sum
// T = sum_w A_w 2^w = sum_k hat_k * t0_k (< 2^134) is put together with integer shifts and adds, then taken through two
// 64-bit Montgomery reductions, T * 2^-128 mod p.  The hat tables are stored times 2^128 mod p, so this is the plain
// residue.  2 x (2 multiply-adds + 6) + ~30 word operations per target instead of six Montgomery multiplications.
template <bool SP>
__device__ __forceinline__ u64 bconv_fold(const u64 (&A)[6], const PrimeC &PT) {
  typedef unsigned __int128 u128;
  const u128 U = (u128)A[0] + ((u128)A[1] << 21) + ((u128)A[2] << 42);  // < 2^104
  const u128 V = (u128)A[3] + ((u128)A[4] << 21) + ((u128)A[5] << 42);  // < 2^104, weight 2^32
  const u128 S = U + (V << 32);                                         // T mod 2^128
  const u64 X0 = (u64)S, X1 = (u64)(S >> 64);
  const u64 X2 = (u64)(V >> 96) + (S < U ? 1u : 0u);                    // T >> 128, < 2^9
  const u64 r1 = mont_reduce<SP>(0, X0, PT);                            // X0 * 2^-64 (mod p), in (0, p]
  const u64 w_lo = X1 + r1;
  const u64 w_hi = X2 + (w_lo < r1 ? 1u : 0u);                          // (w_hi : w_lo) = (T + m*p) / 2^64 < 2^71
  return csub(mont_reduce<SP>(w_hi, w_lo, PT), PT.p);                   // (0, p + w_hi] -> [0, p)
}
template <bool SP, bool PRE = false>
__global__ __launch_bounds__(PW_THREADS) void bconv_kernel(BconvArgs a) {
  const u64 d = (u64)blockIdx.x * PW_THREADS + threadIdx.x;
  const u32 tbase = blockIdx.y * BCONV_TC, e = blockIdx.z;
  const u32 ntgt = a.ntgt;
  const u64 *sp = a.src + ((u64)e * a.src_limbs + a.src_off) * a.N + d;
  u64 acc[BCONV_TC][6];
  u32 tt[BCONV_TC];
#pragma unroll
  for (int t = 0; t < BCONV_TC; ++t) {
    tt[t] = (tbase + t < ntgt) ? tbase + t : ntgt - 1;
#pragma unroll
    for (int w = 0; w < 6; ++w) acc[t][w] = 0;
  }
  // hat factors of one source limb for the thread's targets: uniform, so they travel through the scalar cache into SGPRs.
  // They are requested one term ahead (hn), like the source limbs: a scalar load waited for right where it is used costs
  // its whole latency ten times per loop iteration.
  u64 hc[BCONV_TC], hn[BCONV_TC];
  auto load_hat = [&](u64 (&h)[BCONV_TC], u32 k) {
#pragma unroll
    for (int t = 0; t < BCONV_TC; ++t) h[t] = a.hat_m[(u64)tt[t] * a.hat_stride + k];
  };
  // one source limb into the accumulators of targets [t_lo, t_hi)
  auto term = [&](u64 t0, const u64 (&hat)[BCONV_TC], int t_lo, int t_hi) {
    const u32 b0 = (u32)t0, b1 = (u32)(t0 >> 32);
#pragma unroll
    for (int t = t_lo; t < t_hi; ++t) {
      const u64 h = hat[t];
      const u32 h0 = (u32)h & 0x1FFFFFu, h1 = (u32)(h >> 21) & 0x1FFFFFu, h2 = (u32)(h >> 42);
      acc[t][0] = mad_wide(h0, b0, acc[t][0]);
      acc[t][1] = mad_wide(h1, b0, acc[t][1]);
      acc[t][2] = mad_wide(h2, b0, acc[t][2]);
      acc[t][3] = mad_wide(h0, b1, acc[t][3]);
      acc[t][4] = mad_wide(h1, b1, acc[t][4]);
      acc[t][5] = mad_wide(h2, b1, acc[t][5]);
    }
  };
  auto digit = [&](u32 k, u64 xk) {  // canonical digit [x_k * inv_k] mod the source prime
    if constexpr (PRE) return xk;
    else { const PrimeC PS = a.primes[a.src_first + k]; return csub(mont_mul<SP>(a.inv_m[k], xk, PS), PS.p); }
  };
  // Source limbs and hat factors are requested one term ahead, ping-pong between two register sets (no copy, so no wait at the
  // loop end); indices are clamped instead of branching, the odd last term is peeled.  Scalar loads return out of order, so
  // waiting for one of them waits for all: the next term's factors are requested only AFTER the first target of the current
  // term has forced that wait, and the remaining 24 multiply-adds of the term cover their latency.
  const u32 last = a.nsrc - 1;
  const u32 l1 = last < 1 ? last : 1, l2 = last < 2 ? last : 2;
  u64 xa = sp[0], xb = sp[(u64)l1 * a.N], xc, xd;  // source limbs run TWO terms ahead (four registers, unrolled by four)
  load_hat(hc, 0);
  u32 k = 0;
  auto clampk = [&](u32 q) { return q < last ? q : last; };
  auto step = [&](u32 kk, u64 xcur, u64 &xfar, u64 (&hcur)[BCONV_TC], u64 (&hnext)[BCONV_TC]) {
    xfar = sp[(u64)clampk(kk + 2) * a.N];
    const u64 dg = digit(kk, xcur);
    term(dg, hcur, 0, 1);
    __builtin_amdgcn_sched_barrier(0);
    load_hat(hnext, clampk(kk + 1));
    __builtin_amdgcn_sched_barrier(0);
    term(dg, hcur, 1, BCONV_TC);
  };
  (void)l2;
  for (; k + 3 < a.nsrc; k += 4) {
    step(k, xa, xc, hc, hn);
    step(k + 1, xb, xd, hn, hc);
    step(k + 2, xc, xa, hc, hn);
    step(k + 3, xd, xb, hn, hc);
  }
  // up to three terms left: xa, xb hold limbs k and k+1 (clamped), hc the factors of term k
  if (k < a.nsrc) { step(k, xa, xc, hc, hn); ++k;
    if (k < a.nsrc) { step(k, xb, xd, hn, hc); ++k;
      if (k < a.nsrc) { term(digit(k, xc), hc, 0, BCONV_TC); ++k; } } }
#pragma unroll
  for (int t = 0; t < BCONV_TC; ++t) {
    const PrimeC PT = a.primes[a.tgt_first + tt[t]];
    const u64 r = bconv_fold<SP>(acc[t], PT);
    DBG_RANGE(r, PT.p, 13);
    if (tbase + t < ntgt) a.out[((u64)e * a.out_limbs + a.out_off + tbase + t) * a.N + d] = r;
  }
}

// fHE::apply_galois (src/yell.cpp:14-24) as a gather through a per-galois index table; out-of-place.
__global__ __launch_bounds__(PW_THREADS) void galois_gather_kernel(u64 *dst, const u64 *src, const u32 *perm, u64 N, u64 src_bstride,
                                                                   u64 dst_bstride) {
  const u64 d = (u64)blockIdx.x * PW_THREADS + threadIdx.x;
  const u64 os = (u64)blockIdx.z * src_bstride + (u64)blockIdx.y * N, od = (u64)blockIdx.z * dst_bstride + (u64)blockIdx.y * N;
  dst[od + d] = src[os + perm[d]];
}
__global__ void galois_perm_kernel(u32 *perm, u32 logN, u64 galois) {
  const u32 d = blockIdx.x * blockDim.x + threadIdx.x;
  const u32 N = 1u << logN;
  if (d >= N) return;
  const u32 r = __brev(d) >> (32 - logN);
  const u64 raw = (galois * (2ull * r + 1)) & (2ull * N - 1);
  perm[d] = __brev((u32)((raw - 1) >> 1)) >> (32 - logN);
}
// fHE::apply_galois_non_ntt (src/yell.cpp:41-52): tmp[(d*g) mod N] = +-t_d with the sign flipped when (d*g) mod 2N >= N.  Run as a
// gather so that the stores are coalesced: output j comes from d' = j * g^-1 mod 2N (g odd): d' < N -> +src[d'], else -src[d' - N]
// (then (d' - N) * g = j + N mod 2N).  Limb k uses modulus k % mper; out-of-place.  ginv = g^-1 mod 2N.
__global__ __launch_bounds__(PW_THREADS) void galois_coeff_kernel(u64 *dst, const u64 *src, const PrimeC *primes, u32 logN,
                                                                  u64 ginv, u32 mper) {
  const u64 N = 1ull << logN;
  const u64 j = (u64)blockIdx.x * PW_THREADS + threadIdx.x;
  const u32 cm = blockIdx.y;
  const u64 o = ((u64)blockIdx.z * gridDim.y + cm) * N;
  const u64 dp = (j * ginv) & (2 * N - 1);
  const u64 t = src[o + (dp & (N - 1))];
  const u64 p = primes[cm % mper].p;  // limb cm of a [..][mper][N] stack lives under prime cm % mper
  dst[o + j] = (dp >> logN) ? (t > 0 ? p - t : 0) : t;
}

// poly operator+= (src/evaluator.cpp:284-285): rop, op = [e][nl][N], limb k uses modulus k
__global__ __launch_bounds__(PW_THREADS) void add_kernel(u64 *rop, const u64 *op, const PrimeC *primes, u64 N) {
  const u64 d = ((u64)blockIdx.x * PW_THREADS + threadIdx.x) * 2;
  const u64 p = primes[blockIdx.y].p;
  const u64 o = ((u64)blockIdx.z * gridDim.y + blockIdx.y) * N + d;
  const ulonglong2 a = ld2(rop + o), b = ld2(op + o);
  st2c(rop + o, addmod_c(a.x, b.x, p), addmod_c(a.y, b.y, p), p, 16);
}

// Evaluator::multiply_plain (src/evaluator.cpp:97-106): ct = [b][2][nl][N] *= pt [nl][N]
template <bool SP>
__global__ __launch_bounds__(PW_THREADS) void mul_plain_kernel(u64 *ct, const u64 *pt, const PrimeC *primes, u32 nl, u64 N) {
  const u64 d = ((u64)blockIdx.x * PW_THREADS + threadIdx.x) * 2;
  const u32 l = blockIdx.y % nl;
  const PrimeC P = primes[l];
  const u64 o = ((u64)blockIdx.z * gridDim.y + blockIdx.y) * N + d;
  const ulonglong2 x = ld2(ct + o), y = ld2(pt + (u64)l * N + d);
  st2c(ct + o, mulmod_full<SP>(y.x, x.x, P), mulmod_full<SP>(y.y, x.y, P), P.p, 16);
}

// key upload: x -> x * 2^64 mod p in place; limb l uses midx[l % mper]
template <bool SP>
__global__ __launch_bounds__(PW_THREADS) void to_mont_kernel(u64 *x, const PrimeC *primes, const u32 *midx, u32 mper, u64 N) {
  const u64 d = ((u64)blockIdx.x * PW_THREADS + threadIdx.x) * 2;
  const PrimeC P = primes[midx[blockIdx.y % mper]];
  const u64 o = (u64)blockIdx.y * N + d;
  const ulonglong2 v = ld2(x + o);
  st2(x + o, to_mont<SP>(v.x, P), to_mont<SP>(v.y, P));
}

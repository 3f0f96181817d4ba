// encoder_kernels.h — CKKS canonical-embedding encoder on the GPU: the loops of the reference's
// Encoder (src/encoder.cpp): slot scatter through the generator-3 index map (:5-25, :84-100), the complex butterfly networks
// iFFT_inplace / FFT_inplace over psi^brv(i) (:38-82; same merged-twiddle Gentleman-Sande / Cooley-Tukey shape as the limb NTT),
// coefficient rounding trunc(x * scale / N) mod p per limb (:117-147: mpz_set_d + mpz_mod_ui) and, for decode, the CRT lift to
// the centred integer and its conversion to double (:159-188: poly2mpz, compare with Q/2, mpz_get_d).
// Floating point: complex double throughout, like the reference.  The networks use the NTT's N = R x 256 decomposition: a row pass
// (spans 1..128 inside 256-element rows) and a column pass (spans 256..N/2 on R x 8 tiles), radix-2 stages in LDS.  This row is
// not on the north star's hot path: the kernels are plain (LDS-bound), correct and measured, not tuned.
#pragma once
#include "modarith.h"
#include "ntt_kernels.h"        // canon4
#include "pointwise_kernels.h"  // bconv_fold

typedef double2 cplx;
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csubc(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cplx cmul(cplx a, cplx b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cplx cconj(cplx a) { return make_double2(a.x, -a.y); }

constexpr int ENC_CT = 8;  // columns per column-pass tile (R x 8 complex doubles: 32 KB at N = 2^16)

// slots -> evaluation vector: w[map[i]] = v_i, w[map[i + N/2]] = conj(v_i), everything else zero (encoder.cpp:84-100)
__global__ void enc_scatter_kernel(cplx *w, const cplx *values, const u32 *index_map, u64 N, u64 nvals) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nvals) return;
  const cplx v = values[(u64)blockIdx.y * nvals + i];
  cplx *wb = w + (u64)blockIdx.y * N;
  wb[index_map[i]] = v;
  wb[index_map[i + (N >> 1)]] = cconj(v);
}

// Row pass: the 8 stages with spans below 256, one 256-element row per 128 threads (two rows per workgroup), in LDS.
// INVERSE: iFFT_inplace stages i = 0..7 (tt = 2^i; twiddle inv_roots[h + j], h = N >> (i+1), j = element >> (i+1)): u + v, (u - v) s.
// forward: FFT_inplace stages s = logR .. logN-1 (span 128 >> q; twiddle roots[2^s + (element >> (logN - s))]): u + v s, u - v s.
template <bool INVERSE>
__global__ __launch_bounds__(256) void enc_fft_rows_kernel(cplx *w, const cplx *roots, u32 logN) {
  __shared__ cplx lds[2 * 256];
  const u32 t = threadIdx.x & 127, rl = threadIdx.x >> 7;
  const u32 R = 1u << (logN - 8);
  const u32 row = blockIdx.x * 2 + rl;
  const u64 N = 1ull << logN;
  if (row >= R) return;  // (R >= 4: never taken; kept for safety, whole half-workgroups only, no barrier divergence inside a row)
  cplx *x = w + (u64)blockIdx.y * N + (u64)row * 256;
  cplx *L = lds + rl * 256;
  L[t] = x[t]; L[t + 128] = x[t + 128];
  __syncthreads();
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    if constexpr (INVERSE) {
      const u32 tt = 1u << q;                              // span
      const u32 k = ((t >> q) << (q + 1)) | (t & (tt - 1));
      const u32 j = (row << (7 - q)) + (t >> q);           // (row * 256 + k) >> (q + 1)
      const cplx s = cconj(roots[(N >> (q + 1)) + j]);     // inv_roots = 1 / roots = conj(roots) (unit modulus)
      const cplx u = L[k], v = L[k + tt];
      L[k] = cadd(u, v);
      L[k + tt] = cmul(csubc(u, v), s);
    } else {
      const u32 tt = 128u >> q;
      const u32 k = ((t / tt) * 2 * tt) + (t % tt);
      const u32 g = (row << q) + (t / tt);                 // (row * 256 + k) >> (8 - q)
      const cplx s = roots[((u64)R << q) + g];
      const cplx u = L[k], v = cmul(L[k + tt], s);
      L[k] = cadd(u, v);
      L[k + tt] = csubc(u, v);
    }
    __syncthreads();
  }
  x[t] = L[t]; x[t + 128] = L[t + 128];
}

// Column pass: the logR stages with spans of 256 elements and more, on a tile of R rows x ENC_CT columns in LDS (dynamic: R * ENC_CT
// complex doubles).  INVERSE: iFFT stages i = 8 .. logN-1; forward: FFT stages s = 0 .. logR-1.
template <bool INVERSE>
__global__ __launch_bounds__(256) void enc_fft_cols_kernel(cplx *w, const cplx *roots, u32 logN) {
  extern __shared__ __attribute__((aligned(16))) unsigned char enc_smem[];
  cplx *L = reinterpret_cast<cplx *>(enc_smem);
  const u32 logR = logN - 8, R = 1u << logR;
  const u64 N = 1ull << logN;
  cplx *x = w + (u64)blockIdx.y * N + (u64)blockIdx.x * ENC_CT;
  for (u32 e = threadIdx.x; e < R * ENC_CT; e += 256) L[e] = x[(u64)(e / ENC_CT) * 256 + (e % ENC_CT)];
  __syncthreads();
  const u32 nb = R * ENC_CT / 2;  // butterflies per stage
  for (u32 q = 0; q < logR; ++q) {
    if constexpr (INVERSE) {
      const u32 rs = 1u << q;  // span in rows (tt = 256 << q elements)
      for (u32 b = threadIdx.x; b < nb; b += 256) {
        const u32 c = b % ENC_CT, pr = b / ENC_CT;             // pair index over rows
        const u32 r = ((pr >> q) << (q + 1)) | (pr & (rs - 1));
        const u32 j = pr >> q;                                  // (r * 256) >> (q + 9)
        const cplx s = cconj(roots[(N >> (q + 9)) + j]);
        const cplx u = L[r * ENC_CT + c], v = L[(r + rs) * ENC_CT + c];
        L[r * ENC_CT + c] = cadd(u, v);
        L[(r + rs) * ENC_CT + c] = cmul(csubc(u, v), s);
      }
    } else {
      const u32 rs = R >> (q + 1);
      for (u32 b = threadIdx.x; b < nb; b += 256) {
        const u32 c = b % ENC_CT, pr = b / ENC_CT;
        const u32 g = pr / rs;
        const u32 r = g * 2 * rs + (pr % rs);
        const cplx s = roots[(1u << q) + g];
        const cplx u = L[r * ENC_CT + c], v = cmul(L[(r + rs) * ENC_CT + c], s);
        L[r * ENC_CT + c] = cadd(u, v);
        L[(r + rs) * ENC_CT + c] = csubc(u, v);
      }
    }
    __syncthreads();
  }
  for (u32 e = threadIdx.x; e < R * ENC_CT; e += 256) x[(u64)(e / ENC_CT) * 256 + (e % ENC_CT)] = L[e];
}

// trunc(x) mod p in [0, p): what mpz_set_d (truncation toward zero) followed by mpz_mod_ui (non-negative remainder) give
// (encoder.cpp:135-140).  |x| < 2^63 is converted exactly; beyond that the 53-bit mantissa is reduced and doubled e times.
__device__ __forceinline__ u64 enc_double_mod(double x, u64 p) {
  const bool neg = x < 0;
  const double a = fabs(x);
  u64 r;
  if (a < 9223372036854775808.0) {
    r = (u64)__double2ull_rz(a);
    if (r >= 4 * p) r %= p;                        // short primes only: a 62-bit prime has 4p > 2^63
    else { r = csub(r, 2 * p); r = csub(r, p); }
  } else {
    int e = 0;
    const double m = frexp(a, &e);                 // a = m * 2^e, m in [0.5, 1)
    r = (u64)__double2ull_rz(ldexp(m, 53)) % p;    // 53-bit integer mantissa
    for (int k = 53; k < e; ++k) { r <<= 1; if (r >= p) r -= p; }   // p < 2^62: no overflow
  }
  return (neg && r) ? p - r : r;
}

// coefficients: msg[b][l][d] = trunc(real(w[b][d]) * delta) mod P[l]   (encoder.cpp:124-144)
__global__ __launch_bounds__(256) void enc_round_kernel(u64 *msg, const cplx *w, const PrimeC *primes, u32 nlimbs, u64 N, double delta) {
  const u64 d = (u64)blockIdx.x * 256 + threadIdx.x;
  const u64 b = blockIdx.y;
  const double scaled = w[b * N + d].x * delta;
  for (u32 l = 0; l < nlimbs; ++l) msg[(b * nlimbs + l) * N + d] = enc_double_mod(scaled, primes[l].p);
}

// Decode, CRT lift (encoder.cpp:165-181): residues x_k under q_0 .. q_{n-1} -> the integer in [0, Q) as mixed-radix digits
//     X = v_0 + v_1 q_0 + v_2 q_0 q_1 + ...   (Garner: v_k = (x_k - (v_0 + ... + v_{k-1} q_0..q_{k-2})) (q_0..q_{k-1})^-1 mod q_k),
// centred (X >= floor(Q/2) -> X - Q, decided exactly by comparing digits with those of floor(Q/2)) and converted to double by
// Horner's rule from the top digit, times 1/scale.  A decrypted message is far below Q, so its top digits are zero and the
// conversion is exact to rounding; mpz_get_d truncates instead (a relative 2^-53 either way).
struct DecLiftArgs {
  cplx *w; const u64 *msg;
  const PrimeC *primes;
  const u64 *prod_m;     // [L][L]: row k: (q_0 .. q_{j-1} mod q_k) * 2^128 mod q_k, j < k
  const u64 *ginv_m;     // [L]: (q_0 .. q_{k-1})^-1 mod q_k, Montgomery form (entry 0 unused)
  const u64 *half;       // [n]: mixed-radix digits of floor(Q_n / 2)
  const double *qd;      // [L]: q_k as double
  u32 n, L; u64 N; double inv_scale;
};
constexpr int DEC_MAX_LIMBS = 64;
constexpr int DEC_THREADS = 128;
// Garner's mixed-radix lift, digit k: v_k = (x_k - sum_{j<k} v_j M_{j,k}) * (q_0 .. q_{k-1})^-1 mod q_k with M_{j,k} = q_0 .. q_{j-1} mod q_k.
// The sum is accumulated without carries like the base conversion's (pointwise_kernels.h: bconv_fold): a digit is kept in LDS as
// three 21-bit limbs, M (uniform, from the scalar cache, stored times 2^128) gives its two words, six v_mad_u64_u32 per term into six
// 64-bit sums (room for 2^6 terms: DEC_MAX_LIMBS), one fold per digit.  Round 3 first had a Horner form with a modular multiplication per
// term and the digits in a dynamically indexed private array (2.64 ms per 64 polynomials of 30 limbs at N = 2^16).
template <bool SP>
__global__ __launch_bounds__(DEC_THREADS) void dec_lift_kernel(DecLiftArgs a) {
  extern __shared__ __attribute__((aligned(16))) u32 dec_vl[];   // [n][n] Garner constants (u64), then [3][n][DEC_THREADS]: 21-bit limbs of the digits
  const u32 tid = threadIdx.x, n = a.n;
  u64 *Ml = reinterpret_cast<u64 *>(dec_vl);   // a scalar load per term, waited for where it is used, costs its latency 435 times per coefficient
  for (u32 i = tid; i < n * n; i += DEC_THREADS) Ml[i] = a.prod_m[(u64)(i / n) * a.L + i % n];
  __syncthreads();
  const u64 d = (u64)blockIdx.x * DEC_THREADS + tid;
  const u64 b = blockIdx.y;
  u32 *l0 = dec_vl + 2 * n * n + tid, *l1 = l0 + n * DEC_THREADS, *l2 = l1 + n * DEC_THREADS;
  auto digit = [&](u32 k) { return (u64)l0[k * DEC_THREADS] | ((u64)l1[k * DEC_THREADS] << 21) | ((u64)l2[k * DEC_THREADS] << 42); };
  auto put = [&](u32 k, u64 v) { l0[k * DEC_THREADS] = (u32)v & 0x1FFFFFu; l1[k * DEC_THREADS] = (u32)(v >> 21) & 0x1FFFFFu; l2[k * DEC_THREADS] = (u32)(v >> 42); };
  for (u32 k = 0; k < n; ++k) {
    const PrimeC P = a.primes[k];
    u64 xk = a.msg[(b * n + k) * a.N + d];
    if (xk >= 2 * P.twop) xk %= P.p;   // (out of contract: residues are expected below 4p; kept exact all the same)
    xk = canon4(xk, P);
    u64 acc[6] = {0, 0, 0, 0, 0, 0};
    const u64 *Mk = Ml + k * n;
#pragma unroll 8
    for (u32 j = 0; j < k; ++j) {
      const u64 M = Mk[j];
      const u32 b0 = (u32)M, b1 = (u32)(M >> 32);
      const u32 h0 = l0[j * DEC_THREADS], h1 = l1[j * DEC_THREADS], h2 = l2[j * DEC_THREADS];
      acc[0] = mad_wide(h0, b0, acc[0]); acc[1] = mad_wide(h1, b0, acc[1]); acc[2] = mad_wide(h2, b0, acc[2]);
      acc[3] = mad_wide(h0, b1, acc[3]); acc[4] = mad_wide(h1, b1, acc[4]); acc[5] = mad_wide(h2, b1, acc[5]);
    }
    const u64 part = k ? bconv_fold<SP>(acc, P) : 0;   // sum_j v_j M_{j,k} mod q_k, canonical
    const u64 diff = xk >= part ? xk - part : xk + P.p - part;
    put(k, k ? csub(mont_mul<SP>(a.ginv_m[k], diff, P), P.p) : diff);
  }
  // X >= floor(Q/2)?  (lexicographic from the top digit; equal counts as >=, like mpz_cmp(..) >= 0)
  bool upper = true;
  for (u32 k = n; k-- > 0;) {
    const u64 h = a.half[k], v = digit(k);
    if (v != h) { upper = v > h; break; }
  }
  // digits of Q - X = (Q - 1 - X) + 1: (q_k - 1 - v_k) needs no borrow, then add one with carry; Horner from the top digit:
  // ((v_{n-1} q_{n-2} + v_{n-2}) q_{n-3} + ...) q_0 + v_0 needs the negated digits from the bottom up first
  if (upper) {
    bool carry = true;
    for (u32 k = 0; k < n; ++k) {
      const u64 q = a.primes[k].p;
      u64 t = q - 1 - digit(k);
      if (carry) { t += 1; carry = (t == q); if (carry) t = 0; }
      put(k, t);
    }
  }
  double val = 0.0;
  for (u32 k = n; k-- > 0;) val = val * a.qd[k] + (double)digit(k);
  // (digit k is multiplied by q_{k-1} .. q_0: the factor applied when stepping from digit k to digit k-1 is q_{k-1})
  a.w[b * a.N + d] = make_double2((upper ? -val : val) * a.inv_scale, 0.0);
}

// The same lift for up to DEC_REG_LIMBS limbs with the digits' limbs in REGISTERS: every loop is unrolled to the bound with uniform guards, so
// the digit arrays are indexed statically and the Garner constants come through merged scalar loads; no LDS, no dependent LDS latency under the
// multiply-adds (the LDS form above runs 1.5 waves per SIMD and waits for its own reads).
constexpr int DEC_REG_LIMBS = 32;
template <bool SP>
__global__ __launch_bounds__(256) void dec_lift_reg_kernel(DecLiftArgs a) {
  const u32 n = a.n;
  const u64 d = (u64)blockIdx.x * 256 + threadIdx.x;
  const u64 b = blockIdx.y;
  u32 h0[DEC_REG_LIMBS], h1[DEC_REG_LIMBS], h2[DEC_REG_LIMBS];
#pragma unroll
  for (int k = 0; k < DEC_REG_LIMBS; ++k) { h0[k] = 0; h1[k] = 0; h2[k] = 0; }
#pragma unroll
  for (int k = 0; k < DEC_REG_LIMBS; ++k) {
    if ((u32)k < n) {
      const PrimeC P = a.primes[k];
      u64 xk = a.msg[(b * n + k) * a.N + d];
      if (xk >= 2 * P.twop) xk %= P.p;   // (out of contract: residues are expected below 4p; kept exact all the same)
      xk = canon4(xk, P);
      u64 acc[6] = {0, 0, 0, 0, 0, 0};
      const u64 *Mk = a.prod_m + (u64)k * a.L;
#pragma unroll
      for (int j = 0; j < k; ++j) {
        const u64 M = Mk[j];
        const u32 b0 = (u32)M, b1 = (u32)(M >> 32);
        acc[0] = mad_wide(h0[j], b0, acc[0]); acc[1] = mad_wide(h1[j], b0, acc[1]); acc[2] = mad_wide(h2[j], b0, acc[2]);
        acc[3] = mad_wide(h0[j], b1, acc[3]); acc[4] = mad_wide(h1[j], b1, acc[4]); acc[5] = mad_wide(h2[j], b1, acc[5]);
      }
      const u64 part = k ? bconv_fold<SP>(acc, P) : 0;
      const u64 diff = xk >= part ? xk - part : xk + P.p - part;
      const u64 v = k ? csub(mont_mul<SP>(a.ginv_m[k], diff, P), P.p) : diff;
      h0[k] = (u32)v & 0x1FFFFFu; h1[k] = (u32)(v >> 21) & 0x1FFFFFu; h2[k] = (u32)(v >> 42);
    }
  }
  auto digit = [&](int k) { return (u64)h0[k] | ((u64)h1[k] << 21) | ((u64)h2[k] << 42); };
  // X >= floor(Q/2)?  (lexicographic from the top digit; equal counts as >=): first differing digit from the top decides
  bool upper = true, decided = false;
#pragma unroll
  for (int k = DEC_REG_LIMBS - 1; k >= 0; --k) {
    if ((u32)k < n) {
      const u64 h = a.half[k], v = digit(k);
      if (!decided && v != h) { upper = v > h; decided = true; }
    }
  }
  // negated digits (Q - X) from the bottom up, then Horner from the top
  u64 vv[DEC_REG_LIMBS];
  bool carry = true;
#pragma unroll
  for (int k = 0; k < DEC_REG_LIMBS; ++k) {
    vv[k] = 0;
    if ((u32)k < n) {
      const u64 v = digit(k), q = a.primes[k].p;
      u64 t = q - 1 - v;
      if (carry) { t += 1; carry = (t == q); if (carry) t = 0; }
      vv[k] = upper ? t : v;
    }
  }
  double val = 0.0;
#pragma unroll
  for (int k = DEC_REG_LIMBS - 1; k >= 0; --k)
    if ((u32)k < n) val = val * a.qd[k] + (double)vv[k];
  a.w[b * a.N + d] = make_double2((upper ? -val : val) * a.inv_scale, 0.0);
}

// slots[b][i] = w[b][map[i]], i < N/2 (encoder.cpp:183-187)
__global__ void dec_pick_kernel(cplx *slots, const cplx *w, const u32 *index_map, u64 N) {
  const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (N >> 1)) return;
  slots[(u64)blockIdx.y * (N >> 1) + i] = w[(u64)blockIdx.y * N + index_map[i]];
}

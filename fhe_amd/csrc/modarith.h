// modarith.h — 64-bit modular arithmetic for gfx950 (device side).
//
// Replaces the inner bodies of yell::ops::{mulmod, mulmod_shoup, addmod, muladd, barret_reduction}
// (absent third-party library; usage: reference include/fHE/yell.hpp:14-18, src/base_converter.cpp:227-238,
// src/evaluator.cpp:636-642).  Everything multiplicative is Montgomery (R = 2^64) with constants stored
// in Montgomery form, so data never leaves the plain domain: mont(w~, x) = w*x mod p for w~ = w*R mod p.
// gfx950 has no 64-bit multiplier: products are built from v_mad_u64_u32 (half rate, measured in
// tools/ubench/mulrate.hip).  A prime with p = 1 (mod 2^32) ("SP" form, the repo's prime table) makes the
// Montgomery quotient multiplication-free and cuts a modular multiply from 11 to 6 32-bit multiplies.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;
typedef unsigned int u32;

struct PrimeC {
  u64 p;       // modulus, < 2^62
  u64 twop;    // 2p
  u64 pinv;    // p^-1 mod 2^64
  u64 r2;      // 2^128 mod p  (to-Montgomery factor)
  u64 ninv_m;  // N^-1 mod p in Montgomery form
  u32 p1;      // p >> 32
  u32 n32;     // -p^-1 mod 2^32 (quotient digit factor of the generic word-serial Montgomery steps)
};

__device__ __forceinline__ u64 mad_wide(u32 a, u32 b, u64 c) { return (u64)a * (u64)b + c; }

// Debug build (make -C fhe_amd/csrc DEBUG=1 -> libfhe_hip_debug.so): residue-range checks compiled into the kernels, the
// analogue of the reference's non-release -DCHECK_STRICTMOD (CMakeCompilers.txt:26).  A violation is COUNTED (and the first
// site remembered), never trapped: a trapping kernel faults the GPU.  fhe_hip_debug_violations() reads the counters.
// Sites: 1 fwd rows in (<4p), 2 fwd rows out (<p), 3 fwd rows epilogue out (<p), 4 inv rows in (<2p), 5 inv rows out (<2p),
// 6 fwd cols in (<4p), 7 fwd cols out (<4p), 8 inv cols in (<2p), 9 inv cols out (<p), 10 relin out (<p), 11 relin digit in (<4p),
// 12 rotation special-limb rows out (<2p), 13 bconv out (<p), 14 cross_mult out (<p), 15 (unused since the key products moved into the row passes), 16 pointwise out (<p).
#ifdef FHE_HIP_DEBUG
__device__ unsigned long long fhe_hip_dbg_count = 0;
__device__ unsigned int fhe_hip_dbg_first_site = 0;
__device__ __forceinline__ void dbg_range(u64 v, u64 bound, unsigned site) {
  if (v >= bound) {
    if (atomicAdd(&fhe_hip_dbg_count, 1ull) == 0) fhe_hip_dbg_first_site = site;
  }
}
#define DBG_RANGE(v, bound, site) dbg_range((v), (bound), (site))
#else
#define DBG_RANGE(v, bound, site) ((void)0)
#endif

// (thi:tlo) * 2^-64 mod p, in (0, 2p) provided thi < p.  With thi < c*p the result is < (c+1)*p.
template <bool SP>
__device__ __forceinline__ u64 mont_reduce(u64 thi, u64 tlo, const PrimeC &P) {
  if constexpr (SP) {
    // p = p1*2^32 + 1  =>  p^-1 = 1 - p1*2^32 (mod 2^64):  m = tlo * p^-1 has m0 = T0, m1 = T1 - lo32(T0*p1)
    // hi64(m*p) = m1*p1 + ((T0*p1 + m1) >> 32)
    const u32 T0 = (u32)tlo, T1 = (u32)(tlo >> 32);
    const u64 v = mad_wide(T0, P.p1, 0);
    const u32 m1 = T1 - (u32)v;
    const u64 s = v + m1;
    const u64 w = mad_wide(m1, P.p1, s >> 32);
    return thi + P.p - w;
  } else {
    const u64 m = tlo * P.pinv;
    const u64 w = __umul64hi(m, P.p);
    return thi + P.p - w;
  }
}

template <bool SP>
__device__ __forceinline__ u64 mont_mul(u64 a, u64 b, const PrimeC &P);

__device__ __forceinline__ u64 csub(u64 x, u64 m) { return x >= m ? x - m : x; }
// same for a modulus the compiler holds in scalar registers (a field of the workgroup's PrimeC): the subtraction becomes one
// 64-bit add of -m, kept opaque so that it is not turned back into a two-instruction borrow chain
__device__ __forceinline__ u64 csub_u(u64 x, u64 m) {
  u64 negm = 0 - m;
  // (readfirstlane: the modulus is wave-uniform, but where its words also feed vector operands the compiler may keep it in VGPRs)
  negm = ((u64)(u32)__builtin_amdgcn_readfirstlane((u32)(negm >> 32)) << 32) | (u64)(u32)__builtin_amdgcn_readfirstlane((u32)negm);  // (the builtin returns int)
  asm("" : "+s"(negm));
  return x >= m ? x + negm : x;
}

// exact a*b mod p in [0,p); a < p, b < 2^64 (two Montgomery steps: a*b/R, then * R^2/R)
template <bool SP>
__device__ __forceinline__ u64 mulmod_full(u64 a, u64 b, const PrimeC &P) {
  const u64 t = mont_mul<SP>(a, b, P);
  return csub(mont_mul<SP>(P.r2, t, P), P.p);
}

// x -> x*R mod p, canonical
template <bool SP>
__device__ __forceinline__ u64 to_mont(u64 x, const PrimeC &P) { return csub(mont_mul<SP>(P.r2, x, P), P.p); }

// 128-bit accumulate helper: (hi:lo) += a*b
__device__ __forceinline__ void mac128(u64 &hi, u64 &lo, u64 a, u64 b) {
  const u64 l = a * b;
  const u64 h = __umul64hi(a, b);
  lo += l;
  hi += h + (lo < l ? 1 : 0);
}

// v_mad_u64_u32 behind opaque asm: keeps the 64-bit addend inside the multiply-add (the compiler otherwise splits
// mad(a, b, c) into mul + add and re-associates the carry words through extra v_mov pairs).  The carry-out SGPR pair
// is a dead dummy, never read.
__device__ __forceinline__ u64 mad_op0(u32 a, u32 b) {
  u64 d, cc;
  asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(cc) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ u64 mad_op(u32 a, u32 b, u64 c) {
  u64 d, cc;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cc) : "v"(a), "v"(b), "v"(c));
  return d;
}
// a*b + k with the 64-bit addend in a scalar pair (wave-uniform constant)
__device__ __forceinline__ u64 mad_op_s(u32 a, u32 b, u64 k) {
  u64 d, cc;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cc) : "v"(a), "v"(b), "s"(k));
  return d;
}
__device__ __forceinline__ u64 pair64(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }
// a*b + (2^32 - 1), the addend in a scalar pair: the high word is then h0 + (T0 != 0) and the low word T0 - 1
__device__ __forceinline__ u64 mad_op_k(u32 a, u32 b) {
  u64 d, cc;
  const u64 k = 0xFFFFFFFFull;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cc) : "v"(a), "v"(b), "s"(k));
  return d;
}

// Reference formulation of mont_diff (kept for tools/ubench/bfly.hip and the generic-prime path).
template <bool SP>
__device__ __forceinline__ u64 mont_diff_ref(u64 a, u64 b, const PrimeC &P) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  const u64 t0 = mad_wide(a0, b0, 0);
  const u64 t1 = mad_wide(a0, b1, t0 >> 32);
  const u64 t2 = mad_wide(a1, b0, (u64)(u32)t1);
  const u64 thi = mad_wide(a1, b1, (t1 >> 32) + (t2 >> 32));
  if constexpr (SP) {
    const u32 T0 = (u32)t0, T1 = (u32)t2;
    const u64 v = mad_wide(T0, P.p1, 0);
    const u32 vlo = (u32)v;
    const u32 m1 = T1 - vlo;
    const u32 u = (u32)(v >> 32) + (T1 < vlo ? 1u : 0u);  // carry of vlo + m1 == borrow of T1 - vlo
    return thi - mad_wide(m1, P.p1, (u64)u);
  } else {
    const u64 tlo = (t2 << 32) | (u64)(u32)t0;
    return thi - __umul64hi(tlo * P.pinv, P.p);
  }
}

// t2 = a*b + c and the carry out of that 64-bit sum as 0/1 (one asm statement so that the two wait states gfx950
// needs between a VALU SGPR write and a VALU read of it are inside the statement).
__device__ __forceinline__ u64 mad_op_carry(u32 a, u32 b, u64 c, u32 &carry) {
  u64 d, cc;
  asm("v_mad_u64_u32 %0, %2, %3, %4, %5\n\ts_nop 1\n\tv_cndmask_b32_e64 %1, 0, 1, %2"
      : "=&v"(d), "=v"(carry), "=&s"(cc)
      : "v"(a), "v"(b), "v"(c));
  return d;
}

// D = a*b*2^-64 (mod p) as a signed value in (-p, p), returned mod 2^64; a < p (Montgomery-form constant or
// canonical data, so a1 < 2^30), b <= 2^64 - 2^34 (any lazy value < 4p + 2^32).  For b up to 2^64-1 the upper bound
// weakens to p + 2^30.   mont_mul(a, b) == mont_diff(a, b) + p.
//
// SP form (p = p1*2^32 + 1), word-serial; on gfx950 every VALU instruction except plain 32-bit VOP1/VOP2 ones issues
// at the rate of v_mad_u64_u32 (tools/ubench/instrate.hip), so the 64-bit additions ride in multiply-add addends:
//   step 1 (additive):  q0 = -T0 mod 2^32, c0 = (T0 != 0):
//        V = (T + q0*p) / 2^32 - a1*b1*2^32 = (h0 + c0) + a1*b0 + q0*p1 + a0*b1              (65 bits: V2:V1:V0)
//   step 2 (subtractive): V - V0*p has a zero low word:  D = a1*b1 + (V2:V1) - V0*p1.
// c0 comes for free as the carry of adding 2^32 - 1 inside the first multiply-add.
// 6 multiply-adds + 2 carry-word operations + 4 plain ones (the product-then-reduce form compiles to 6 + 18).
//
// Generic primes (any odd p < 2^62; the table a drop-in caller passes, e.g. yell's params::P): the same word-serial scheme
// with the quotient digits q = T0 * (-+p^-1 mod 2^32) (one v_mul_lo_u32 each) and the products q*p0 (p0 = p mod 2^32):
//   step 1 (additive):  q0 = T0 * n32:  (a0*b0 + q0*p0) / 2^32 = hi(u) + 2^32 cu with u = q0*p0 + a0*b0 (low word zero,
//        65 bits: carry cu), which is the addend of a1*b0;  V = that + q0*p1 + a0*b1                       (65 bits: V2:V1:V0)
//   step 2 (subtractive, signed result): q1 = V0 * pinv32;  D = a1*b1 + (V2:V1) - q1*p1 - hi(q1*p0)         in (-p, p)
//        (additive, mont_mul): q1 = V0 * n32;  M = a1*b1 + (V2:V1) + q1*p1 + hi(q1*p0 + 2^32 - 1)            in [0, 2p)
// 8 multiply-adds + 2 low multiplies + 6 word operations = 16 (the product-then-reduce form it replaces compiled to 11 + ~20).
template <bool ADDITIVE>
__device__ __forceinline__ u64 mont_generic(u64 a, u64 b, const PrimeC &P) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  const u32 p0 = (u32)P.p, p1 = P.p1, pinv32 = (u32)P.pinv, n32 = P.n32;  // (n32 from the table: computed here, the compiler negates every quotient instead)
  const u64 t0 = mad_op0(a0, b0);
  const u32 q0 = (u32)t0 * n32;
  u32 cu, V2;
  const u64 u = mad_op_carry(q0, p0, t0, cu);             // low word 0
  const u64 t1 = mad_op(a1, b0, pair64((u32)(u >> 32), cu));  // < 2^62 + 2^33
  const u64 m = mad_op(q0, p1, t1);                       // < 2^63 + 2^33
  const u64 t2 = mad_op_carry(a0, b1, m, V2);
  const u32 V0 = (u32)t2, V1 = (u32)(t2 >> 32);
  if constexpr (ADDITIVE) {
    const u32 q1 = V0 * n32;
    const u64 w = mad_op_k(q1, p0);                       // high word = hi(q1*p0) + (V0 != 0)
    const u64 n = mad_op(q1, p1, w >> 32);
    u32 w1, w2;
    const u32 elo = __builtin_addc(V1, (u32)n, 0u, &w1);
    const u32 ehi = __builtin_addc(V2, (u32)(n >> 32), w1, &w2);
    return mad_op(a1, b1, pair64(elo, ehi));
  } else {
    const u32 q1 = V0 * pinv32;
    const u64 w = mad_op0(q1, p0);                        // low word = V0
    const u64 n = mad_op(q1, p1, w >> 32);
    u32 w1, w2;
    const u32 elo = __builtin_subc(V1, (u32)n, 0u, &w1);
    const u32 ehi = __builtin_subc(V2, (u32)(n >> 32), w1, &w2);
    return mad_op(a1, b1, pair64(elo, ehi));
  }
}

// SP core with a constant offset: returns D + off (mod 2^64), off wave-uniform.  The offset rides in the addend of the step-2
// multiply-add (n - off instead of n; everything downstream is arithmetic mod 2^64), so "+ p" costs no instruction.
template <bool OFF>
__device__ __forceinline__ u64 mont_sp(u64 a, u64 b, const PrimeC &P, u64 negoff) {
  const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  const u32 p1 = P.p1;
  const u64 t0 = mad_op_k(a0, b0);                      // a0*b0 + 2^32 - 1 <= 2^64 - 2^32: (T0 - 1, h0 + c0)
  const u64 t1 = mad_op(a1, b0, t0 >> 32);              // < 2^62 + 2^32 (a move and a shared zero: cheaper than a 64-bit shift)
  const u64 m = mad_op(~(u32)t0, p1, t1);               // q0 = -T0 = ~(T0 - 1);  < 2^63 + 2^32
  u32 V2, w1, w2;
  const u64 t2 = mad_op_carry(a0, b1, m, V2);
  const u32 V0 = (u32)t2, V1 = (u32)(t2 >> 32);
  u64 n;
  if constexpr (OFF) n = mad_op_s(V0, p1, negoff); else n = mad_op0(V0, p1);
  const u32 elo = __builtin_subc(V1, (u32)n, 0u, &w1);
  const u32 ehi = __builtin_subc(V2, (u32)(n >> 32), w1, &w2);
  return mad_op(a1, b1, pair64(elo, ehi));
}

template <bool SP>
__device__ __forceinline__ u64 mont_diff(u64 a, u64 b, const PrimeC &P) {
  if constexpr (SP) return mont_sp<false>(a, b, P, 0);
  else return mont_generic<false>(a, b, P);
}

// mont_diff with a wave-uniform first factor (a twiddle that every lane of the workgroup shares, held in scalar registers):
// its two words go into the multiply-adds as scalar operands, which saves the two per-twiddle copies into vector registers.
// (One scalar source per instruction: the 2^32 - 1 addend of the first multiply-add therefore sits in a vector pair.)
__device__ __forceinline__ u64 mont_diff_ua(u64 a_uniform, u64 b, const PrimeC &P) {
  const u32 a0 = (u32)a_uniform, a1 = (u32)(a_uniform >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
  const u32 p1 = P.p1;
  const u64 k = 0xFFFFFFFFull;
  u64 t0, t1, t2, d, cc;
  u32 V2, w1, w2;
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(t0), "=s"(cc) : "s"(a0), "v"(b0), "v"(k));
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(t1), "=s"(cc) : "s"(a1), "v"(b0), "v"(t0 >> 32));
  const u64 m = mad_op(~(u32)t0, p1, t1);
  asm("v_mad_u64_u32 %0, %2, %3, %4, %5\n\ts_nop 1\n\tv_cndmask_b32_e64 %1, 0, 1, %2"
      : "=&v"(t2), "=v"(V2), "=&s"(cc)
      : "s"(a0), "v"(b1), "v"(m));
  const u32 V0 = (u32)t2, V1 = (u32)(t2 >> 32);
  const u64 n = mad_op0(V0, p1);
  const u32 elo = __builtin_subc(V1, (u32)n, 0u, &w1);
  const u32 ehi = __builtin_subc(V2, (u32)(n >> 32), w1, &w2);
  asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cc) : "s"(a1), "v"(b1), "v"(pair64(elo, ehi)));
  return d;
}
template <bool SP>
__device__ __forceinline__ void ct_bfly_ua(u64 &X, u64 &Y, u64 W_uniform, const PrimeC &P) {
  u64 D;
  if constexpr (SP) D = mont_diff_ua(W_uniform, Y, P); else D = mont_diff<false>(W_uniform, Y, P);
  const u64 xp = (X >= P.twop) ? X - P.p : X + P.p;
  X = xp + D;
  Y = xp - D;
}

// a * b * 2^-64 mod p in (0, 2p); a < 2p (Montgomery-form constant or data), b < 4p + 2^32.
template <bool SP>
__device__ __forceinline__ u64 mont_mul(u64 a, u64 b, const PrimeC &P) {
  if constexpr (SP) {
    // D + p with the offset folded (no instruction).  The modulus is wave-uniform in every kernel; the readfirstlane pair only
    // tells the compiler so where it holds the constant in vector registers (it folds away where P already sits in SGPRs).
    const u64 negp = 0 - P.p;
    return mont_sp<true>(a, b, P, pair64(__builtin_amdgcn_readfirstlane((u32)negp), __builtin_amdgcn_readfirstlane((u32)(negp >> 32))));
  }
  else return mont_generic<true>(a, b, P);
}

template <bool SP>
__device__ __forceinline__ void ct_bfly_ref(u64 &X, u64 &Y, u64 W, const PrimeC &P) {
  const u64 D = mont_diff_ref<SP>(W, Y, P);
  const u64 xp = (X >= P.twop) ? X - P.p : X + P.p;
  X = xp + D;
  Y = xp - D;
}

// Harvey lazy Cooley-Tukey butterfly: X, Y in [0,4p) -> (0,4p);  W in Montgomery form.
// With D = mont_diff(W, Y) in (-p, p) and xp = (X mod 2p) + p in [p, 3p):  X' = xp + D, Y' = xp - D.
template <bool SP>
__device__ __forceinline__ void ct_bfly(u64 &X, u64 &Y, u64 W, const PrimeC &P) {
  const u64 D = mont_diff<SP>(W, Y, P);
  const u64 xp = (X >= P.twop) ? X - P.p : X + P.p;
  X = xp + D;
  Y = xp - D;
}

// lazy Gentleman-Sande butterfly: X, Y in [0,2p) -> [0,2p)
template <bool SP>
__device__ __forceinline__ void gs_bfly(u64 &X, u64 &Y, u64 W, const PrimeC &P) {
  const u64 s = X + Y;
  const u64 d = X - Y + P.twop;
  X = csub_u(s, P.twop);
  Y = mont_mul<SP>(W, d, P);
}

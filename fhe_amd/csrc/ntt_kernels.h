// ntt_kernels.h — batched negacyclic NTT for gfx950: replaces yell::ntt<N>::forward/backward
// (absent library; call sites reference src/evaluator.cpp:365,390,668,707, src/base_converter.cpp:224,260).
//
// Ordering contract (reference src/yell.cpp:16-21): natural-order coefficients in, evaluations at
// psi^(2*brv(d)+1) out (bit-reversed), i.e. the merged-twiddle Cooley-Tukey forward / Gentleman-Sande
// inverse with twiddle table tw[k] = psi^brv(k) (Montgomery form).  Stage s (m = 2^s groups) uses
// tw[2^s + (element_index >> (logN - s))].
//
// Decomposition: N = R x 256.  Forward = column pass (stages 0..logR-1, stride-256 butterflies over an
// R x CT tile held in LDS) then row pass (stages logR..logN-1 inside contiguous 256-element rows).
// Inverse = row pass then column pass (+ N^-1).  Column pass: every thread keeps 16 elements in registers for
// 4 stages (radix-16), exchanging through LDS once; row pass: 8 elements per thread, register phases of 3 + 3 + 2
// stages.  Values stay lazy ([0,4p) forward, [0,2p) inverse) and are made canonical only by the last pass.
// For N <= 2^14 a limb fits one workgroup's LDS and plain transforms run as ONE kernel (ntt_onepass_*, end of this file).
#pragma once
#include "modarith.h"
#include <type_traits>

struct NttArgs {
  const u64 *src;       // source limbs
  u64 *dst;             // destination limbs (may equal src)
  const u64 *tw;        // [M][N] twiddles (forward or inverse table), Montgomery form
  const PrimeC *primes; // [M]
  const u32 *midx;      // modulus map: limb l of a batch entry uses midx[(l / mdiv) % mper]
  u32 mdiv, mper;
  u32 smod;             // source limb = l % smod (broadcast of a source limb over several outputs)
  u32 mdiv_inv, mper_inv, smod_inv;  // ceil(2^32 / d) of the three divisors (limb ids stay below 2^16: exact, see limb_div)
  u32 logN, logR;
  u32 l0;               // limb offset: limb id = blockIdx.y + l0
  u32 skip_diag;        // if nonzero (= mdiv): skip limbs with (l / mdiv) == (l % mdiv)  (relinearisation, j == i)
  u64 src_bstride, dst_bstride;  // elements between batch entries
  // optional epilogue fused into the forward row pass (the NTT result t never reaches HBM):
  //   epi == 1: out = (src + 2p - t) * c mod p          BaseConverter::rescale inner loop (src/base_converter.cpp:264-271)
  //   epi == 2: out = (src + t) * c [+ add] mod p       tail of approximated_mod_down (:227-240) [+ evaluator.cpp:398]
  // src/add/out = [b](bstride)[limb][N] addressed by the same (b, limb) as the NTT; c[modulus index] in Montgomery form.
  const u64 *inv_scale;  // inverse column pass only: [M] Montgomery-form factors used instead of N^-1 (N^-1 times a constant the consumer needs)
  const u32 *perm;   // inverse row pass only: element d of a limb is read from src[perm[d]] (fHE::apply_galois fused into the load)
  const u32 *epi_add_perm;  // same for the epilogue addend
  // inverse row pass only (8-element kernel): the element read is src + pre_k * pre_a  (Decryptor::decrypt, decryptor.cpp:24-35:
  // bx + sx * ax); pre_a = [b](src_bstride)[limb][N] like src, pre_k = [limb][N]; null = plain load
  const u64 *pre_a, *pre_k;
  // epi == 4 (hybrid rotation, normal limbs): batch entry = 2*ciphertext + half; out = (galois(ax)_j * key_half_j + t) * c
  // [+ galois(bx)_j on half 0] with epi_src = the INPUT ciphertexts [ct](epi_src_bstride)[2][limb][N] (bx at 0, ax at epi_half),
  // read through epi_add_perm (null = identity), epi_key0 / epi_key1 = beta / alpha [modulus][N] Montgomery form
  const u64 *epi_key0, *epi_key1;
  u64 epi_half;
  u32 epi;
  const u64 *epi_src, *epi_c, *epi_add;
  u64 *epi_out;
  u64 epi_src_bstride, epi_add_bstride, epi_out_bstride;
};

// Limb ids are wave-uniform and below 2^16, the divisors below 2^16: x / d = hi32(x * ceil(2^32 / d)) exactly (d = 1 apart), two
// scalar instructions where the compiler's generic 32-bit division costs every thread ~20 vector ones.
__device__ __forceinline__ u32 limb_div(u32 x, u32 d, u32 inv) { return d == 1 ? x : __umulhi(x, inv); }
__device__ __forceinline__ u32 limb_mod(u32 x, u32 d, u32 inv) { return x - limb_div(x, d, inv) * d; }
__device__ __forceinline__ u32 limb_modulus(const NttArgs &a, u32 l) { return a.midx[limb_mod(limb_div(l, a.mdiv, a.mdiv_inv), a.mper, a.mper_inv)]; }
__device__ __forceinline__ u32 limb_source(const NttArgs &a, u32 l) { return limb_mod(l, a.smod, a.smod_inv); }

// Gather of the element pair (d, d+1), d even, through a galois index table.  In the bit-reversed order the automorphism maps
// d and d+1 to the two halves of ONE aligned pair (they differ in the top bit of the exponent index, and multiplying by an
// odd g keeps that a top-bit difference), so a single 16-byte load fetches both; the table entry's low bit says whether
// they are swapped.
__device__ __forceinline__ ulonglong2 gather_pair(const u64 *limb, u32 pd) {
  const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(&limb[pd & ~1u]);
  return (pd & 1u) ? make_ulonglong2(v.y, v.x) : v;
}

// LDS hand-off inside one wave (a 256-element row is owned by 32 consecutive lanes of one wave)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ u64 canon4(u64 v, const PrimeC &P) { return csub_u(csub_u(v, P.twop), P.p); }

// acc + k*x*2^-64 (mod p) with acc kept lazy in [0,4p); k < p in Montgomery form, x < 4p + 2^32
template <bool SP>
__device__ __forceinline__ u64 lazy_acc(u64 acc, u64 k, u64 x, const PrimeC &P) {
  const u64 xp = (acc >= P.twop) ? acc - P.p : acc + P.p;
  return xp + mont_diff<SP>(k, x, P);
}

template <bool SP, int LK>
__device__ __forceinline__ void fwd_phase(u64 (&e)[1 << LK], const u64 *__restrict__ tw, u32 base, const PrimeC &P) {
#pragma unroll
  for (int q = 0; q < LK; ++q) {
    const int half = 1 << (LK - 1 - q);
#pragma unroll
    for (int kk = 0; kk < (1 << q); ++kk) {
      const u64 W = tw[(base << q) + kk];
#pragma unroll
      for (int j = 0; j < half; ++j) {
        const int a = kk * 2 * half + j;
        ct_bfly<SP>(e[a], e[a + half], W, P);
      }
    }
  }
}

// Twiddles of one register phase fetched ahead of time: w[(1 << q) - 1 + kk] = tw[(base << q) + kk].  The stand-alone
// passes issue these loads together with the data loads at the top of the kernel, so that a wave waits for memory once
// instead of once per phase (forward passes: 4-5 % faster; the same change made the inverse row pass slower and is not used there).
template <int LK>
__device__ __forceinline__ void load_tw(u64 (&w)[(1 << LK) - 1], const u64 *__restrict__ tw, u32 base) {
#pragma unroll
  for (int q = 0; q < LK; ++q)
#pragma unroll
    for (int kk = 0; kk < (1 << q); ++kk) w[(1 << q) - 1 + kk] = tw[(base << q) + kk];
}
template <bool SP, int LK>
__device__ __forceinline__ void fwd_phase_w(u64 (&e)[1 << LK], const u64 (&w)[(1 << LK) - 1], const PrimeC &P) {
#pragma unroll
  for (int q = 0; q < LK; ++q) {
    const int half = 1 << (LK - 1 - q);
#pragma unroll
    for (int kk = 0; kk < (1 << q); ++kk)
#pragma unroll
      for (int j = 0; j < half; ++j) ct_bfly<SP>(e[kk * 2 * half + j], e[kk * 2 * half + j + half], w[(1 << q) - 1 + kk], P);
  }
}

// forward phase whose twiddles are the same for every lane (first phase of the column pass: heap indices 1 .. 2^LK - 1)
template <bool SP, int LK, typename TwPtr>
__device__ __forceinline__ void fwd_phase_ua(u64 (&e)[1 << LK], TwPtr tw, const PrimeC &P) {
#pragma unroll
  for (int q = 0; q < LK; ++q) {
    const int half = 1 << (LK - 1 - q);
#pragma unroll
    for (int kk = 0; kk < (1 << q); ++kk) {
      const u64 W = tw[(1 << q) + kk];
#pragma unroll
      for (int j = 0; j < half; ++j) ct_bfly_ua<SP>(e[kk * 2 * half + j], e[kk * 2 * half + j + half], W, P);
    }
  }
}

// Last stage of an inverse transform with the scaling folded in (N^-1, or the factor a consumer wants in its place): the stage's single
// twiddle W becomes W*s once per thread, X' = (X + Y)*s and Y' = (X - Y)*(W*s), canonical.  One modular multiplication per element
// instead of the butterfly's one per pair plus one per element for the scaling: half a multiplication per element saved.
template <bool SP>
__device__ __forceinline__ void gs_bfly_scaled(u64 &X, u64 &Y, u64 s_m, u64 Ws_m, const PrimeC &P) {
  const u64 sum = X + Y;               // < 4p
  const u64 dif = X - Y + P.twop;      // (0, 4p)
  X = csub(mont_mul<SP>(s_m, sum, P), P.p);
  Y = csub(mont_mul<SP>(Ws_m, dif, P), P.p);
}
template <bool SP, int LK, typename TwPtr>
__device__ __forceinline__ void inv_phase(u64 (&e)[1 << LK], TwPtr tw, u32 base, const PrimeC &P) {
#pragma unroll
  for (int q = LK - 1; q >= 0; --q) {
    const int half = 1 << (LK - 1 - q);
#pragma unroll
    for (int kk = 0; kk < (1 << q); ++kk) {
      const u64 W = tw[(base << q) + kk];
#pragma unroll
      for (int j = 0; j < half; ++j) {
        const int a = kk * 2 * half + j;
        gs_bfly<SP>(e[a], e[a + half], W, P);
      }
    }
  }
}

// inverse phase with the twiddles of load_tw already in registers
template <bool SP, int LK>
__device__ __forceinline__ void inv_phase_w(u64 (&e)[1 << LK], const u64 (&w)[(1 << LK) - 1], const PrimeC &P) {
#pragma unroll
  for (int q = LK - 1; q >= 0; --q) {
    const int half = 1 << (LK - 1 - q);
#pragma unroll
    for (int kk = 0; kk < (1 << q); ++kk)
#pragma unroll
      for (int j = 0; j < half; ++j) gs_bfly<SP>(e[kk * 2 * half + j], e[kk * 2 * half + j + half], w[(1 << q) - 1 + kk], P);
  }
}
// inv_phase / inv_phase_w whose last stage (q == 0, twiddle tw[base] / w[0]) is the transform's last: outputs scaled and canonical
template <bool SP, int LK, typename TwPtr>
__device__ __forceinline__ void inv_phase_final(u64 (&e)[1 << LK], TwPtr tw, u32 base, const PrimeC &P, u64 s_m) {
#pragma unroll
  for (int q = LK - 1; q >= 1; --q) {
    const int half = 1 << (LK - 1 - q);
#pragma unroll
    for (int kk = 0; kk < (1 << q); ++kk) {
      const u64 W = tw[(base << q) + kk];
#pragma unroll
      for (int j = 0; j < half; ++j) gs_bfly<SP>(e[kk * 2 * half + j], e[kk * 2 * half + j + half], W, P);
    }
  }
  const u64 Ws = csub(mont_mul<SP>(tw[base], s_m, P), P.p);
#pragma unroll
  for (int j = 0; j < (1 << (LK - 1)); ++j) gs_bfly_scaled<SP>(e[j], e[j + (1 << (LK - 1))], s_m, Ws, P);
}
template <bool SP, int LK>
__device__ __forceinline__ void inv_phase_w_final(u64 (&e)[1 << LK], const u64 (&w)[(1 << LK) - 1], const PrimeC &P, u64 s_m) {
#pragma unroll
  for (int q = LK - 1; q >= 1; --q) {
    const int half = 1 << (LK - 1 - q);
#pragma unroll
    for (int kk = 0; kk < (1 << q); ++kk)
#pragma unroll
      for (int j = 0; j < half; ++j) gs_bfly<SP>(e[kk * 2 * half + j], e[kk * 2 * half + j + half], w[(1 << q) - 1 + kk], P);
  }
  const u64 Ws = csub(mont_mul<SP>(w[0], s_m, P), P.p);
#pragma unroll
  for (int j = 0; j < (1 << (LK - 1)); ++j) gs_bfly_scaled<SP>(e[j], e[j + (1 << (LK - 1))], s_m, Ws, P);
}

// ---------------------------------------------------------------------------------------------
// Row pass (forward): 8 stages inside contiguous 256-element rows; 32 threads per row with 8 elements each, three
// register phases of 3 + 3 + 2 stages, up to 8 rows per workgroup.  A row is owned by 32 consecutive lanes of one wave,
// so the exchanges need only wave-level ordering (no s_barrier).  LDS image per row: 8 chunks of 32 elements at chunk
// stride 36 (conflict-free strided reads).
// ---------------------------------------------------------------------------------------------
constexpr int ROW8_LDS = 8 * 36;
__device__ __forceinline__ u32 row8_pos(u32 e) { return (e >> 5) * 36 + (e & 31); }

template <bool SP, int EPI = 0>
__global__ __launch_bounds__(256) void ntt_rows8_fwd_kernel(NttArgs a) {
  __shared__ __attribute__((aligned(16))) u64 lds[8 * ROW8_LDS];
  const u32 tid = threadIdx.x, rloc = tid >> 5, t = tid & 31;
  const u32 row = blockIdx.x * (blockDim.x >> 5) + rloc;  // 8 rows per workgroup (R < 8: all R rows, 32 * R threads)
  // EPI == 4: grid = (row blocks, batch entries, limbs), the limb slowest: every entry of the batch reads the same two key limbs and
  // the same galois index table, which then stay in L2 across the batch, and the two halves of a ciphertext (adjacent entries, same
  // XCD) share the gathered galois(ax) (profiles/r03: 11.0 GB crossed the L2 boundary per launch with the limb in grid y)
  const u32 l = (EPI == 4 ? blockIdx.z : blockIdx.y) + a.l0, b = EPI == 4 ? blockIdx.y : blockIdx.z;
  const u32 m = limb_modulus(a, l);
  const PrimeC P = a.primes[m];
  const u64 N = 1ull << a.logN;
  const u64 *__restrict__ tw = a.tw + (u64)m * N;
  const u64 *s = a.src + (u64)b * a.src_bstride + (u64)limb_source(a, l) * N + (u64)row * 256;
  u64 *d = a.dst + (u64)b * a.dst_bstride + (u64)l * N + (u64)row * 256;
  u64 *L = lds + rloc * ROW8_LDS;
  const u32 baseA = (1u << a.logR) + row;
  const u32 c = t >> 2, r = t & 3;
  u64 e[8], wA[7], wB[7], wC0[3], wC1[3];
  // phase A: spans 128, 64, 32; thread holds {t + 32k}.  All twiddles of the three phases are requested here, behind the data.
#pragma unroll
  for (int k = 0; k < 8; ++k) { e[k] = s[t + 32 * k]; DBG_RANGE(e[k], 2 * P.twop, 1); }
  load_tw<3>(wA, tw, baseA);
  load_tw<3>(wB, tw, (baseA << 3) + c);
  load_tw<2>(wC0, tw, (baseA << 6) + 2 * t);
  load_tw<2>(wC1, tw, (baseA << 6) + 2 * t + 1);
  __builtin_amdgcn_sched_barrier(0);
  fwd_phase_w<SP, 3>(e, wA, P);
#pragma unroll
  for (int k = 0; k < 8; ++k) L[k * 36 + t] = e[k];
  wave_lds_sync();
  // phase B: spans 16, 8, 4 inside 32-element chunks; thread (c = t >> 2, r = t & 3) holds {32c + r + 4k}
#pragma unroll
  for (int k = 0; k < 8; ++k) e[k] = L[c * 36 + r + 4 * k];
  fwd_phase_w<SP, 3>(e, wB, P);
#pragma unroll
  for (int k = 0; k < 8; ++k) L[c * 36 + r + 4 * k] = e[k];
  wave_lds_sync();
  // phase C: spans 2, 1; thread holds the 8 contiguous elements {8t + k} = two groups of four
  {
    const u32 p0 = row8_pos(8 * t);
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(&L[p0 + k]);
      e[k] = v.x; e[k + 1] = v.y;
    }
    u64 g0[4] = {e[0], e[1], e[2], e[3]}, g1[4] = {e[4], e[5], e[6], e[7]};
    fwd_phase_w<SP, 2>(g0, wC0, P);
    fwd_phase_w<SP, 2>(g1, wC1, P);
#pragma unroll
    for (int k = 0; k < 4; ++k) { e[k] = g0[k]; e[4 + k] = g1[k]; }
#pragma unroll
    for (int k = 0; k < 8; ++k) e[k] = (EPI == 0) ? canon4(e[k], P) : csub(e[k], P.twop);
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < 8; k += 2) *reinterpret_cast<ulonglong2 *>(&L[p0 + k]) = make_ulonglong2(e[k], e[k + 1]);
  }
  wave_lds_sync();
  // coalesced 16-byte stores: instruction i covers elements [64i, 64i + 64) of the row
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const u32 el = 64 * i + 2 * t;
    const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(&L[row8_pos(el)]);
    if constexpr (EPI == 0) {
      DBG_RANGE(v.x, P.p, 2); DBG_RANGE(v.y, P.p, 2);
      *reinterpret_cast<ulonglong2 *>(&d[el]) = v;
    } else if constexpr (EPI == 4) {
      // tail of approximated_mod_down (src/base_converter.cpp:227-240) for both halves of a rotation, with the key product of the
      // normal limbs (src/evaluator.cpp:373-384) and "+ galois(bx)" (:394-398) formed here from the input ciphertext
      const u32 h = b & 1;
      const u64 cm = a.epi_c[m];
      const u64 *cb = a.epi_src + (u64)(b >> 1) * a.epi_src_bstride + (u64)l * N;
      const u32 idx = row * 256 + el;
      const u32 pd = a.epi_add_perm ? a.epi_add_perm[idx] : idx;
      const ulonglong2 xx = gather_pair(cb + a.epi_half, pd);
      const u64 x0 = xx.x, x1 = xx.y;
      const ulonglong2 kv = *reinterpret_cast<const ulonglong2 *>(&(h ? a.epi_key1 : a.epi_key0)[(u64)m * N + idx]);
      u64 r0 = csub(mont_mul<SP>(cm, mont_mul<SP>(kv.x, x0, P) + v.x, P), P.p);
      u64 r1 = csub(mont_mul<SP>(cm, mont_mul<SP>(kv.y, x1, P) + v.y, P), P.p);
      if (!h) { const ulonglong2 bb = gather_pair(cb, pd); r0 = csub(r0 + bb.x, P.p); r1 = csub(r1 + bb.y, P.p); }
      DBG_RANGE(r0, P.p, 3); DBG_RANGE(r1, P.p, 3);
      *reinterpret_cast<ulonglong2 *>(&a.epi_out[(u64)b * a.epi_out_bstride + (u64)l * N + idx]) = make_ulonglong2(r0, r1);
    } else {
      const u64 off = (u64)l * N + (u64)row * 256 + el;
      const ulonglong2 h = *reinterpret_cast<const ulonglong2 *>(&a.epi_src[(u64)b * a.epi_src_bstride + off]);
      u64 r0, r1;
      [[maybe_unused]] u64 cm = 0;
      if constexpr (EPI != 3) cm = a.epi_c[m];
      if constexpr (EPI == 3) {
        // Encryptor::encrypt add_product_of (src/encryptor.cpp:31,35): out = NTT(e) + u^ * pk, u^ = epi_src, pk = epi_add
        const ulonglong2 pk = *reinterpret_cast<const ulonglong2 *>(&a.epi_add[(u64)b * a.epi_add_bstride + off]);
        r0 = csub(mulmod_full<SP>(pk.x, h.x, P) + csub(v.x, P.p), P.p);
        r1 = csub(mulmod_full<SP>(pk.y, h.y, P) + csub(v.y, P.p), P.p);
      } else if constexpr (EPI == 1) {
        r0 = csub(mont_mul<SP>(cm, h.x + P.twop - v.x, P), P.p);
        r1 = csub(mont_mul<SP>(cm, h.y + P.twop - v.y, P), P.p);
        if (a.epi_add) {  // MixedRotKey tail: rop[0] = ct[0] + ... (src/evaluator.cpp:503-514)
          const ulonglong2 ad = *reinterpret_cast<const ulonglong2 *>(&a.epi_add[(u64)b * a.epi_add_bstride + off]);
          r0 = csub(r0 + ad.x, P.p); r1 = csub(r1 + ad.y, P.p);
        }
      } else {
        r0 = csub(mont_mul<SP>(cm, h.x + v.x, P), P.p);
        r1 = csub(mont_mul<SP>(cm, h.y + v.y, P), P.p);
        if (a.epi_add) {
          ulonglong2 ad;
          if (a.epi_add_perm) {
            const u64 *al = a.epi_add + (u64)b * a.epi_add_bstride + (u64)l * N;
            ad = gather_pair(al, a.epi_add_perm[row * 256 + el]);
          } else {
            ad = *reinterpret_cast<const ulonglong2 *>(&a.epi_add[(u64)b * a.epi_add_bstride + off]);
          }
          r0 = csub(r0 + ad.x, P.p); r1 = csub(r1 + ad.y, P.p);
        }
      }
      DBG_RANGE(r0, P.p, 3); DBG_RANGE(r1, P.p, 3);
      *reinterpret_cast<ulonglong2 *>(&a.epi_out[(u64)b * a.epi_out_bstride + off]) = make_ulonglong2(r0, r1);
    }
  }
}

// Row pass of the inverse NTT, 8 elements per thread (mirror of ntt_rows8_fwd_kernel): spans 1, 2 | 4, 8, 16 | 32, 64, 128;
// coalesced 16-byte loads (optionally gathered through a galois index table), output lazy in [0,2p) for the column pass.
// PERM: the source is read through NttArgs::perm (a template parameter, not a run-time test: the four loads of a thread then
// issue back to back instead of each waiting for the one before it behind a branch).
template <bool SP, bool PRE = false, bool PERM = false>
__global__ __launch_bounds__(256) void ntt_rows8_inv_kernel(NttArgs a) {
  __shared__ __attribute__((aligned(16))) u64 lds[8 * ROW8_LDS];
  const u32 tid = threadIdx.x, rloc = tid >> 5, t = tid & 31;
  const u32 row = blockIdx.x * (blockDim.x >> 5) + rloc;  // 8 rows per workgroup (R < 8: all R rows, 32 * R threads)
  const u32 l = blockIdx.y + a.l0, b = blockIdx.z;
  const u32 m = limb_modulus(a, l);
  const PrimeC P = a.primes[m];
  const u64 N = 1ull << a.logN;
  const u64 *__restrict__ tw = a.tw + (u64)m * N;
  const u64 *s = a.src + (u64)b * a.src_bstride + (u64)limb_source(a, l) * N + (u64)row * 256;
  u64 *d = a.dst + (u64)b * a.dst_bstride + (u64)l * N + (u64)row * 256;
  u64 *L = lds + rloc * ROW8_LDS;
  const u32 baseA = (1u << a.logR) + row;
  const u32 c = t >> 2, r = t & 3, p0 = row8_pos(8 * t);
  {
    ulonglong2 v[4];
    if constexpr (PERM) {
      const u64 *sl = s - (u64)row * 256;  // limb base
      u32 pd[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) pd[i] = a.perm[row * 256 + 64 * i + 2 * t];
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = gather_pair(sl, pd[i]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const ulonglong2 *>(&s[64 * i + 2 * t]);
    }
    if constexpr (PRE) {  // Decryptor::decrypt: the element is bx + sx * ax (NttArgs::pre_a)
      ulonglong2 x[4], k[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u64 lo = (u64)limb_source(a, l) * N + (u64)row * 256 + 64 * i + 2 * t;
        x[i] = *reinterpret_cast<const ulonglong2 *>(&a.pre_a[(u64)b * a.src_bstride + lo]);
        k[i] = *reinterpret_cast<const ulonglong2 *>(&a.pre_k[lo]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i].x = csub(mulmod_full<SP>(k[i].x, x[i].x, P) + csub(v[i].x, P.p), P.p);
        v[i].y = csub(mulmod_full<SP>(k[i].y, x[i].y, P) + csub(v[i].y, P.p), P.p);
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      DBG_RANGE(v[i].x, P.twop, 4); DBG_RANGE(v[i].y, P.twop, 4);
      *reinterpret_cast<ulonglong2 *>(&L[row8_pos(64 * i + 2 * t)]) = v[i];
    }
  }
  wave_lds_sync();
  u64 e[8];
#pragma unroll
  for (int k = 0; k < 8; k += 2) {
    const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(&L[p0 + k]);
    e[k] = v.x; e[k + 1] = v.y;
  }
  {
    u64 g0[4] = {e[0], e[1], e[2], e[3]}, g1[4] = {e[4], e[5], e[6], e[7]};
    inv_phase<SP, 2>(g0, tw, (baseA << 6) + 2 * t, P);
    inv_phase<SP, 2>(g1, tw, (baseA << 6) + 2 * t + 1, P);
#pragma unroll
    for (int k = 0; k < 4; ++k) { e[k] = g0[k]; e[4 + k] = g1[k]; }
  }
  wave_lds_sync();
#pragma unroll
  for (int k = 0; k < 8; k += 2) *reinterpret_cast<ulonglong2 *>(&L[p0 + k]) = make_ulonglong2(e[k], e[k + 1]);
  wave_lds_sync();
#pragma unroll
  for (int k = 0; k < 8; ++k) e[k] = L[c * 36 + r + 4 * k];
  inv_phase<SP, 3>(e, tw, (baseA << 3) + c, P);
#pragma unroll
  for (int k = 0; k < 8; ++k) L[c * 36 + r + 4 * k] = e[k];
  wave_lds_sync();
#pragma unroll
  for (int k = 0; k < 8; ++k) e[k] = L[k * 36 + t];
  inv_phase<SP, 3>(e, tw, baseA, P);
#pragma unroll
  for (int k = 0; k < 8; ++k) { DBG_RANGE(e[k], P.twop, 5); d[t + 32 * k] = e[k]; }
}

// ---------------------------------------------------------------------------------------------
// Special-limb middle of the hybrid rotation key switch (reference src/evaluator.cpp:363-392), fused: finish the forward NTT of
// an extended limb (row pass), multiply by the two key limbs, and run the row pass of both inverse NTTs, all on the same
// 256-element rows in registers/LDS.  The thread that ends the forward pass with the 8 contiguous values {8t + k} is the one
// that starts the inverse pass from them, so the two key products and the switch of direction need no exchange.
// Replaces: forward row pass + key product + two inverse row passes (three round trips through HBM) for the K special limbs.
// ---------------------------------------------------------------------------------------------
struct RotSpArgs {
  u64 *e0, *e1;                  // [b](estride)[first_limb + k][N]: e0 holds the column-pass output of the special limbs on entry
  u64 estride;                   // elements between batch entries
  const u64 *beta_m, *alpha_m;   // [L+K][N] Montgomery form
  const u64 *tw_fwd, *tw_inv;
  const PrimeC *primes;
  u32 first_mod;                 // modulus index of the first special limb (= L)
  const u32 *mmap;               // optional: limb -> modulus index (all Li + K limbs of the power-basis rotation); null = first_mod + limb
  u32 first_limb;                // its limb position inside an entry
  u32 logN, logR;
};

template <bool SP>
__global__ __launch_bounds__(256) void rot_special_rows8_kernel(RotSpArgs a) {
  __shared__ __attribute__((aligned(16))) u64 lds[8 * ROW8_LDS];
  const u32 tid = threadIdx.x, rloc = tid >> 5, t = tid & 31;
  const u32 row = blockIdx.x * (blockDim.x >> 5) + rloc;  // 8 rows per workgroup (R < 8: all R rows, 32 * R threads)
  // grid = (row blocks, batch entries, limbs): the limb slowest, so that its two key limbs stay in L2 across the batch
  const u32 ly = blockIdx.z, b = blockIdx.y;
  const u32 m = a.mmap ? a.mmap[ly] : a.first_mod + ly;
  const PrimeC P = a.primes[m];
  const u64 N = 1ull << a.logN;
  const u64 *__restrict__ twf = a.tw_fwd + (u64)m * N;
  const u64 *__restrict__ twi = a.tw_inv + (u64)m * N;
  const u64 off = (u64)b * a.estride + (u64)(a.first_limb + ly) * N + (u64)row * 256;
  u64 *L = lds + rloc * ROW8_LDS;
  const u32 baseA = (1u << a.logR) + row;
  const u32 c = t >> 2, r = t & 3, p0 = row8_pos(8 * t);
  u64 x[8];
  // forward row pass (as ntt_rows8_fwd_kernel), result lazy in [0,4p), thread holds {8t + k}
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = a.e0[off + t + 32 * k];
  fwd_phase<SP, 3>(x, twf, baseA, P);
#pragma unroll
  for (int k = 0; k < 8; ++k) L[k * 36 + t] = x[k];
  wave_lds_sync();
#pragma unroll
  for (int k = 0; k < 8; ++k) x[k] = L[c * 36 + r + 4 * k];
  fwd_phase<SP, 3>(x, twf, (baseA << 3) + c, P);
#pragma unroll
  for (int k = 0; k < 8; ++k) L[c * 36 + r + 4 * k] = x[k];
  wave_lds_sync();
#pragma unroll
  for (int k = 0; k < 8; k += 2) {
    const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(&L[p0 + k]);
    x[k] = v.x; x[k + 1] = v.y;
  }
  {
    u64 g0[4] = {x[0], x[1], x[2], x[3]}, g1[4] = {x[4], x[5], x[6], x[7]};
    fwd_phase<SP, 2>(g0, twf, (baseA << 6) + 2 * t, P);
    fwd_phase<SP, 2>(g1, twf, (baseA << 6) + 2 * t + 1, P);
#pragma unroll
    for (int k = 0; k < 4; ++k) { x[k] = g0[k]; x[4 + k] = g1[k]; }
  }
  // two key products and two inverse row passes (spans 1..128), outputs lazy in [0,2p) as ntt_rows_kernel<INVERSE> leaves them
  const u64 koff = (u64)m * N + (u64)row * 256 + 8 * t;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const u64 *key = (h == 0 ? a.beta_m : a.alpha_m) + koff;
    u64 y[8];
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      const ulonglong2 kv = *reinterpret_cast<const ulonglong2 *>(&key[k]);
      y[k] = mont_mul<SP>(kv.x, x[k], P); y[k + 1] = mont_mul<SP>(kv.y, x[k + 1], P);  // (0,2p)
    }
    {
      u64 g0[4] = {y[0], y[1], y[2], y[3]}, g1[4] = {y[4], y[5], y[6], y[7]};
      inv_phase<SP, 2>(g0, twi, (baseA << 6) + 2 * t, P);
      inv_phase<SP, 2>(g1, twi, (baseA << 6) + 2 * t + 1, P);
#pragma unroll
      for (int k = 0; k < 4; ++k) { y[k] = g0[k]; y[4 + k] = g1[k]; }
    }
    wave_lds_sync();  // the row image may still be read by the previous exchange
#pragma unroll
    for (int k = 0; k < 8; k += 2) *reinterpret_cast<ulonglong2 *>(&L[p0 + k]) = make_ulonglong2(y[k], y[k + 1]);
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < 8; ++k) y[k] = L[c * 36 + r + 4 * k];
    inv_phase<SP, 3>(y, twi, (baseA << 3) + c, P);
#pragma unroll
    for (int k = 0; k < 8; ++k) L[c * 36 + r + 4 * k] = y[k];
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < 8; ++k) y[k] = L[k * 36 + t];
    inv_phase<SP, 3>(y, twi, baseA, P);
    u64 *d = (h == 0 ? a.e0 : a.e1) + off;
#pragma unroll
    for (int k = 0; k < 8; ++k) { DBG_RANGE(y[k], P.twop, 12); d[t + 32 * k] = y[k]; }
  }
}

// ---------------------------------------------------------------------------------------------
// Column pass: LOGR stages over rows (stride 256 elements).  Tile = R rows x CT columns in LDS
// ([R][CT+1]); threads = R*CT/16.  Two register phases: LA = max(LOGR-4,0) stages on rows
// {g + (R>>LA) k}, then LB = min(LOGR,4) stages on rows {(u << LB) + k}; inverse runs them mirrored
// and multiplies by N^-1 (output canonical).
// ---------------------------------------------------------------------------------------------
template <int LOGR> struct ColCfg {
  static constexpr int R = 1 << LOGR;
  static constexpr int LA = LOGR > 4 ? LOGR - 4 : 0;
  static constexpr int LB = LOGR > 4 ? 4 : LOGR;
  // 16 columns for R = 256 (32 KiB tile, 256 threads); fewer rows take proportionally wider tiles, so that a workgroup keeps
  // four waves and row segments grow to 256 B .. 2 KiB (N = 2^14 / 2^15: +5 % / +2.5 % over one-wave workgroups)
  static constexpr int CT = (16 * (256 / R)) > 256 ? 256 : 16 * (256 / R);
  static constexpr int CS = CT + 1;
  static constexpr int THREADS = R * CT / 16;
};

template <bool SP, bool INVERSE, int LOGR>
__global__ __launch_bounds__(ColCfg<LOGR>::THREADS) void ntt_cols_kernel(NttArgs a) {
  using Cfg = ColCfg<LOGR>;
  constexpr int R = Cfg::R, LA = Cfg::LA, LB = Cfg::LB, CT = Cfg::CT, CS = Cfg::CS, T = Cfg::THREADS;
  __shared__ u64 lds[R * CS];
  const u32 tid = threadIdx.x;
  const u32 l = blockIdx.y + a.l0, b = blockIdx.z;
  if (a.skip_diag) { const u32 q = limb_div(l, a.mdiv, a.mdiv_inv); if (q == l - q * a.mdiv) return; }  // skip_diag == mdiv (host-checked)
  const u32 m = limb_modulus(a, l);
  const PrimeC P = a.primes[m];
  const u64 N = 1ull << a.logN;
  const u64 *__restrict__ tw = a.tw + (u64)m * N;
  const u32 col0 = blockIdx.x * CT;
  const u64 *s = a.src + (u64)b * a.src_bstride + (u64)limb_source(a, l) * N + col0;
  u64 *d = a.dst + (u64)b * a.dst_bstride + (u64)l * N + col0;
  const u64 nscale = (INVERSE && a.inv_scale) ? a.inv_scale[m] : P.ninv_m;

  if constexpr (!INVERSE) {
    // twiddles of the second register phase (per thread) requested up front, with the data loads of the first
    u64 wB[16 >> LB][(1 << LB) - 1];
    if constexpr (LA > 0) {
#pragma unroll
      for (int rep = 0; rep < (16 >> LB); ++rep) load_tw<LB>(wB[rep], tw, (1u << LA) + (tid + rep * T) / CT);
    }
    if constexpr (LA > 0) {
      constexpr int G = R >> LA;  // row groups in phase A
#pragma unroll
      for (int rep = 0; rep < (16 >> LA); ++rep) {
        const u32 slot = tid + rep * T, g = slot / CT, c = slot % CT;
        u64 e[1 << LA];
#pragma unroll
        for (int k = 0; k < (1 << LA); ++k) { e[k] = s[(u64)(g + G * k) * 256 + c]; DBG_RANGE(e[k], 2 * P.twop, 6); }
        fwd_phase_ua<SP, LA>(e, tw, P);
#pragma unroll
        for (int k = 0; k < (1 << LA); ++k) lds[(g + G * k) * CS + c] = e[k];
      }
      __syncthreads();
    }
#pragma unroll
    for (int rep = 0; rep < (16 >> LB); ++rep) {
      const u32 slot = tid + rep * T, u = slot / CT, c = slot % CT;
      u64 e[1 << LB];
#pragma unroll
      for (int k = 0; k < (1 << LB); ++k) {
        const u32 row = (u << LB) + k;
        e[k] = (LA > 0) ? lds[row * CS + c] : s[(u64)row * 256 + c];
        DBG_RANGE(e[k], 2 * P.twop, 6);
      }
      if constexpr (LA > 0) fwd_phase_w<SP, LB>(e, wB[rep], P);
      else fwd_phase<SP, LB>(e, tw, (1u << LA) + u, P);
#pragma unroll
      for (int k = 0; k < (1 << LB); ++k) { DBG_RANGE(e[k], 2 * P.twop, 7); d[(u64)((u << LB) + k) * 256 + c] = e[k]; }  // lazy [0,4p)
    }
  } else {
#pragma unroll
    for (int rep = 0; rep < (16 >> LB); ++rep) {
      const u32 slot = tid + rep * T, u = slot / CT, c = slot % CT;
      u64 e[1 << LB];
#pragma unroll
      for (int k = 0; k < (1 << LB); ++k) { e[k] = s[(u64)((u << LB) + k) * 256 + c]; DBG_RANGE(e[k], P.twop, 8); }
      if constexpr (LA > 0) inv_phase<SP, LB>(e, tw, (1u << LA) + u, P);
      else inv_phase_final<SP, LB>(e, tw, 1u + u, P, nscale);   // (u == 0: R <= 16 rows are one register phase)
#pragma unroll
      for (int k = 0; k < (1 << LB); ++k) {
        const u32 row = (u << LB) + k;
        if constexpr (LA > 0) lds[row * CS + c] = e[k];
        else { DBG_RANGE(e[k], P.p, 9); d[(u64)row * 256 + c] = e[k]; }
      }
    }
    if constexpr (LA > 0) {
      constexpr int G = R >> LA;
      __syncthreads();
#pragma unroll
      for (int rep = 0; rep < (16 >> LA); ++rep) {
        const u32 slot = tid + rep * T, g = slot / CT, c = slot % CT;
        u64 e[1 << LA];
#pragma unroll
        for (int k = 0; k < (1 << LA); ++k) e[k] = lds[(g + G * k) * CS + c];
        inv_phase_final<SP, LA>(e, tw, 1u, P, nscale);
#pragma unroll
        for (int k = 0; k < (1 << LA); ++k) { DBG_RANGE(e[k], P.p, 9); d[(u64)(g + G * k) * 256 + c] = e[k]; }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Fused relinearisation row pass (reference src/evaluator.cpp:663-691): for output limb i and a block of
// rows, loop over the digits j: finish NTT_i(ct2_power[j]) (stages logR..logN-1 of the column-pass output
// t1[i][j]), multiply by evk.beta_j[i] / evk.alpha_j[i] and accumulate; then add ct[0]/ct[1] and store.
// The Li^2 NTT outputs never reach HBM and the twiddles of (i, row) are reused for all j.  For j == i the
// NTT is the identity on ct[2] limb i (NTT_i(INTT_i(x)) = x), so it is read directly.
// t1 = [b][Li(i)][Li(j)][N]; ct3 = [b][3][Li][N]; out = [b][2][Li][N]; keys [L][L][N] Montgomery form.
// ---------------------------------------------------------------------------------------------
// The same kernel serves the MixedRotKey key switch (src/evaluator.cpp:424-473): there the output limbs are the Li
// normal primes plus the special prime P[L] (omap), the digits are the limbs of galois(ax), the diagonal digit is ct[1]
// itself (:436-437) and nothing is added.
struct RelinArgs {
  const u64 *t1;           // [b][gi][nd][N] column-pass output of the output limbs i0 .. i0+gi-1
  const u64 *diag;         // digit j == output limb i: [b](diag_bstride)[limb][N], read directly
  const u64 *add0, *add1;  // optional addends per half: [b](add_bstride)[limb][N]; null = none
  u64 *out;                // [b][2][nout][N]
  const u64 *beta_m, *alpha_m;  // [key j][klimbs][N] Montgomery form
  const u64 *tw; const PrimeC *primes;
  const u32 *omap;         // output limb -> global modulus index (also the key limb index)
  u64 diag_bstride, add_bstride;
  u32 nd, nout, klimbs, logN, logR;
  u32 nb, rb_count;        // batch entries and row blocks per limb (grid decode)
  u32 i0, gi;              // this launch covers output limbs i0 .. i0+gi-1 (a group: bounds the size of t1)
};

// 32 threads per row, 8 elements per thread, up to 8 rows per workgroup; grid = gi * rb_count * nb workgroups.
// Vector-memory loads return in issue order (one vmcnt
// counter), so a prefetch only hides latency if everything issued before it is also consumed before it.  Here issue order ==
// consumption order: inside the key MAC of digit j the kernel requests digit j+1 (when x has been parked in the row image in
// LDS, from which the MAC reads its operands), then keys_{j+1}[groups 0,1] once MAC groups 0,1 have used keys_j[0,1], then
// keys_{j+1}[2,3]; the NTT of a digit issues no vector-memory loads at all (its 255 twiddles per row are read from an LDS copy
// made once per workgroup, heap order: entry (1 << s) + g = stage s, group g).  16 + 32 VGPRs of loads stay in flight across
// an NTT or a MAC; measured 6 % faster than the same kernel without the ordering, which waits for the newest load.
template <bool SP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void relin_rows8p_mac_kernel(RelinArgs a) {
  __shared__ __attribute__((aligned(16))) u64 lds[8 * ROW8_LDS];
  __shared__ __attribute__((aligned(16))) u64 twl[8 * 256];
  const u32 tid = threadIdx.x, rloc = tid >> 5, t = tid & 31;
  u32 rb, b, irel;
  {
    const u32 id = blockIdx.x, nb = a.nb, RB = a.rb_count;
    if ((RB & 7) == 0) {
      const u32 xcd = id & 7, q = id >> 3;
      b = q % nb;
      const u32 rest = q / nb;
      rb = xcd + 8 * (rest % (RB >> 3));
      irel = rest / (RB >> 3);
    } else {
      b = id % nb;
      rb = (id / nb) % RB;
      irel = id / (nb * RB);
    }
  }
  const u32 i = a.i0 + irel;
  const u32 row = rb * (blockDim.x >> 5) + rloc;  // 8 rows per workgroup (R < 8: all R rows)
  const u32 nd = a.nd;
  const u32 m = a.omap[i];
  const PrimeC P = a.primes[m];
  const u64 N = 1ull << a.logN;
  u64 *L = lds + rloc * ROW8_LDS;
  u64 *T = twl + rloc * 256;
  const u32 baseA = (1u << a.logR) + row;
  const u64 row_off = (u64)row * 256;
  const u32 c = t >> 2, r = t & 3, p0 = row8_pos(8 * t);
  {
    const u64 *tw = a.tw + (u64)m * N;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const u32 idx = t + 32 * k;
      if (idx) { const u32 st = 31 - __clz(idx); T[idx] = tw[((u64)(baseA - 1) << st) + idx]; }
    }
  }
  wave_lds_sync();
  u64 acc0[8], acc1[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { acc0[k] = 0; acc1[k] = 0; }
  u64 x[8];
  // both kinds of digit are requested with the same eight strided loads {t + 32k} (one code path: the wait counters of the
  // prefetch stay exact); the diagonal digit is ct[2] limb i itself and skips the NTT
  auto load_digit = [&](u32 j) {
    const u64 *s = (j == i) ? a.diag + (u64)b * a.diag_bstride + (u64)i * N + row_off
                            : a.t1 + (((u64)b * a.gi + irel) * nd + j) * N + row_off;
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = s[t + 32 * k];
  };
  [[maybe_unused]] auto check_digit = [&]() {
#pragma unroll
    for (int k = 0; k < 8; ++k) DBG_RANGE(x[k], 2 * P.twop, 11);
  };
  ulonglong2 kb[4], ka[4];
  auto load_keys = [&](u32 j, int q0) {  // two MAC groups (q0, q0 + 1) of digit j
    const u64 koff = ((u64)j * a.klimbs + m) * N + row_off;
#pragma unroll
    for (int q = q0; q < q0 + 2; ++q) {
      kb[q] = *reinterpret_cast<const ulonglong2 *>(&a.beta_m[koff + 64 * q + 2 * t]);
      ka[q] = *reinterpret_cast<const ulonglong2 *>(&a.alpha_m[koff + 64 * q + 2 * t]);
    }
  };
  auto mac = [&](int q) {
    const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(&L[row8_pos(64 * q + 2 * t)]);
    // accumulators stay lazy in [0,4p) like the X side of a butterfly: acc' = ((acc mod 2p) + p) + D, D in (-p, p)
    acc0[2 * q] = lazy_acc<SP>(acc0[2 * q], kb[q].x, v.x, P);
    acc0[2 * q + 1] = lazy_acc<SP>(acc0[2 * q + 1], kb[q].y, v.y, P);
    acc1[2 * q] = lazy_acc<SP>(acc1[2 * q], ka[q].x, v.x, P);
    acc1[2 * q + 1] = lazy_acc<SP>(acc1[2 * q + 1], ka[q].y, v.y, P);
  };
  // issue order == consumption order everywhere: digit_0, keys_0[0..1], keys_0[2..3], then inside MAC_j: digit_{j+1},
  // keys_{j+1}[0..1] (after MAC groups 0,1 have consumed keys_j[0..1]), keys_{j+1}[2..3] (after groups 2,3)
  if (nd) {
    __builtin_amdgcn_sched_barrier(0);
    load_digit(0);
    __builtin_amdgcn_sched_barrier(0);
    load_keys(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    load_keys(0, 2);
    __builtin_amdgcn_sched_barrier(0);
  }
  for (u32 j = 0; j < nd; ++j) {
    const u32 jn = j + 1 < nd ? j + 1 : j;  // the last digit re-requests itself (keeps one code path; results unused)
#ifdef FHE_HIP_DEBUG
    check_digit();
#endif
    if (j == i) {
#pragma unroll
      for (int k = 0; k < 8; ++k) L[k * 36 + t] = x[k];
    } else {
      fwd_phase<SP, 3>(x, T, 1u, P);
#pragma unroll
      for (int k = 0; k < 8; ++k) L[k * 36 + t] = x[k];
      wave_lds_sync();
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] = L[c * 36 + r + 4 * k];
      fwd_phase<SP, 3>(x, T, 8u + c, P);
#pragma unroll
      for (int k = 0; k < 8; ++k) L[c * 36 + r + 4 * k] = x[k];
      wave_lds_sync();
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(&L[p0 + k]);
        x[k] = v.x; x[k + 1] = v.y;
      }
      {
        u64 g0[4] = {x[0], x[1], x[2], x[3]}, g1[4] = {x[4], x[5], x[6], x[7]};
        fwd_phase<SP, 2>(g0, T, 64u + 2 * t, P);
        fwd_phase<SP, 2>(g1, T, 64u + 2 * t + 1, P);
#pragma unroll
        for (int k = 0; k < 4; ++k) { x[k] = g0[k]; x[4 + k] = g1[k]; }
      }
      wave_lds_sync();
#pragma unroll
      for (int k = 0; k < 8; k += 2) *reinterpret_cast<ulonglong2 *>(&L[p0 + k]) = make_ulonglong2(x[k], x[k + 1]);
    }
    wave_lds_sync();
    __builtin_amdgcn_sched_barrier(0);
    load_digit(jn);
    __builtin_amdgcn_sched_barrier(0);
    mac(0); mac(1);
    __builtin_amdgcn_sched_barrier(0);
    load_keys(jn, 0);
    __builtin_amdgcn_sched_barrier(0);
    mac(2); mac(3);
    __builtin_amdgcn_sched_barrier(0);
    load_keys(jn, 2);
    __builtin_amdgcn_sched_barrier(0);
    wave_lds_sync();
  }
  const u64 half = (u64)a.nout * N;
  u64 *o = a.out + ((u64)b * 2 * a.nout + i) * N + row_off;
  const u64 *c0 = a.add0 ? a.add0 + (u64)b * a.add_bstride + (u64)i * N + row_off : nullptr;
  const u64 *c1 = a.add1 ? a.add1 + (u64)b * a.add_bstride + (u64)i * N + row_off : nullptr;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const u32 e = 64 * q + 2 * t;
    u64 r00 = canon4(acc0[2 * q], P), r01 = canon4(acc0[2 * q + 1], P), r10 = canon4(acc1[2 * q], P), r11 = canon4(acc1[2 * q + 1], P);
    if (c0) { const ulonglong2 d0 = *reinterpret_cast<const ulonglong2 *>(&c0[e]); r00 = csub(r00 + d0.x, P.p); r01 = csub(r01 + d0.y, P.p); }
    if (c1) { const ulonglong2 d1 = *reinterpret_cast<const ulonglong2 *>(&c1[e]); r10 = csub(r10 + d1.x, P.p); r11 = csub(r11 + d1.y, P.p); }
    DBG_RANGE(r00, P.p, 10); DBG_RANGE(r01, P.p, 10); DBG_RANGE(r10, P.p, 10); DBG_RANGE(r11, P.p, 10);
    *reinterpret_cast<ulonglong2 *>(&o[e]) = make_ulonglong2(r00, r01);
    *reinterpret_cast<ulonglong2 *>(&o[half + e]) = make_ulonglong2(r10, r11);
  }
}

// ---------------------------------------------------------------------------------------------
// Same fused relinearisation row pass with the key products accumulated WITHOUT reduction (round 3).  A Montgomery key MAC costs
// 16 issue slots (11 for the product, 5 to keep the accumulator below 4p); here a MAC is its 6 multiply-adds and nothing else:
// the lazy NTT output x (< 2^64) is cut once into three 22-bit limbs (shared by the beta and the alpha product), the key word
// pair (k0, k1) multiplies each limb, and every one of the six partial products (< 2^54) is added into a 64-bit accumulator of
// its own by the multiply-add that forms it (room for 2^8 digits).  The six sums are folded once per output coefficient, after
// the digit loop (relin_fold: integer shifts and adds to the 134-bit sum, then two 64-bit Montgomery reductions, i.e. * 2^-128;
// the keys of this kernel are stored times 2^128 mod p, so the result is the plain residue).
// 96 accumulator registers per thread allow 4 coefficients per thread: 64 threads (one wave) per 256-element row, register
// phases of 2 + 2 + 2 + 2 stages, three wave-local LDS exchanges; a thread ends the row pass with the 4 contiguous coefficients
// {4t + k} and takes its key words with two 16-byte loads per key, so the MAC needs no exchange at all.
// ---------------------------------------------------------------------------------------------
constexpr int ROW4_LDS = 288;
// element e of a row sits at e + 2 (e >> 4): two pad words per 16 elements (groups of 4 stay 16-byte aligned)

template <bool SP>
__device__ __forceinline__ u64 relin_fold(const u64 (&A)[6], const PrimeC &PT) {
  typedef unsigned __int128 u128;
  const u128 U = (u128)A[0] + ((u128)A[1] << 22) + ((u128)A[2] << 44);  // k0 * x          < 2^104
  const u128 V = (u128)A[3] + ((u128)A[4] << 22) + ((u128)A[5] << 44);  // k1 * x, weight 2^32
  const u128 S = U + (V << 32);
  const u64 X0 = (u64)S, X1 = (u64)(S >> 64);
  const u64 X2 = (u64)(V >> 96) + (S < U ? 1u : 0u);
  const u64 r1 = mont_reduce<SP>(0, X0, PT);
  const u64 w_lo = X1 + r1;
  const u64 w_hi = X2 + (w_lo < r1 ? 1u : 0u);
  return csub(mont_reduce<SP>(w_hi, w_lo, PT), PT.p);
}

template <bool SP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void relin_rows4_mac_kernel(RelinArgs a) {
  // per wave: two row images (two digits are transformed side by side: four independent butterflies per stage and thread, which
  // is what keeps the multiply-add pipeline full at two waves per SIMD)
  __shared__ __attribute__((aligned(16))) u64 lds[4 * 2 * ROW4_LDS];
  // the wave index is wave-uniform: through readfirstlane, so that everything that depends on the row (addresses of the digit, the
  // keys, the twiddles) is scalar arithmetic and the loads take a scalar base with a 32-bit lane offset
  const u32 tid = threadIdx.x, rloc = __builtin_amdgcn_readfirstlane(tid >> 6), t = tid & 63;
  u32 rb, b, irel;
  {
    const u32 id = blockIdx.x, nb = a.nb, RB = a.rb_count;
    if ((RB & 7) == 0) {
      const u32 xcd = id & 7, q = id >> 3;
      b = q % nb;
      const u32 rest = q / nb;
      rb = xcd + 8 * (rest % (RB >> 3));
      irel = rest / (RB >> 3);
    } else {
      b = id % nb;
      rb = (id / nb) % RB;
      irel = id / (nb * RB);
    }
  }
  const u32 i = a.i0 + irel;
  const u32 row = rb * 4 + rloc;  // 4 rows per workgroup, one wave each
  const u32 nd = a.nd;
  const u32 m = a.omap[i];
  const PrimeC P = a.primes[m];
  const u64 N = 1ull << a.logN;
  u64 *L = lds + rloc * 2 * ROW4_LDS;
  const u32 baseA = (1u << a.logR) + row;
  const u64 row_off = (u64)row * 256;
  const u32 ga = t >> 4, ra = t & 15, gc = t >> 2, rc = t & 3;
  // the thread's twelve twiddles (three per register phase) are the same for every digit: loaded once, kept in registers
  u64 wA[3], wB[3], wC[3], wD[3];
  {
    const u64 *tw = a.tw + (u64)m * N;
    load_tw<2>(wA, tw, baseA);
    load_tw<2>(wB, tw, (baseA << 2) + ga);
    load_tw<2>(wC, tw, (baseA << 4) + gc);
    load_tw<2>(wD, tw, (baseA << 6) + t);
  }
  u64 acc[4][2][6];
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int w = 0; w < 6; ++w) acc[e][h][w] = 0;
  // LDS positions of the four ownership patterns, row4_pos spelled out as base + constant (immediate offsets):
  //   {t + 64k} -> pA + 72k;  {64 ga + ra + 16k} -> pB + 18k;  {16 gc + rc + 4k} -> pC + 4k;  {4t + k} -> pD + k
  u64 *const pA = L + t + 2 * (t >> 4), *const pB = L + 72 * ga + ra, *const pC = L + 18 * gc + rc, *const pD = L + 4 * t + 2 * (t >> 2);
  struct Keys { ulonglong2 b0, b1, a0, a1; };
  auto load_keys = [&](u32 j) {
    const u64 koff = ((u64)j * a.klimbs + m) * N + row_off;   // wave-uniform
    const u64 *kbp = a.beta_m + koff, *kap = a.alpha_m + koff;
    Keys k;
    k.b0 = *reinterpret_cast<const ulonglong2 *>(&kbp[4 * t]); k.b1 = *reinterpret_cast<const ulonglong2 *>(&kbp[4 * t + 2]);
    k.a0 = *reinterpret_cast<const ulonglong2 *>(&kap[4 * t]); k.a1 = *reinterpret_cast<const ulonglong2 *>(&kap[4 * t + 2]);
    return k;
  };
  auto load_digit = [&](u64 (&v)[4], u32 j) {   // column-pass output of digit j: the thread's {t + 64k}
    const u64 *s = a.t1 + (((u64)b * a.gi + irel) * nd + j) * N + row_off;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = s[t + 64 * k];
  };
  auto mac = [&](const u64 (&x)[4], const Keys &key) {
    const u64 kb[4] = {key.b0.x, key.b0.y, key.b1.x, key.b1.y}, ka[4] = {key.a0.x, key.a0.y, key.a1.x, key.a1.y};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const u32 xl = (u32)x[e], xh = (u32)(x[e] >> 32);
      const u32 x0 = xl & 0x3FFFFFu, x1 = ((xl >> 22) | (xh << 10)) & 0x3FFFFFu, x2 = xh >> 12;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const u64 k = h ? ka[e] : kb[e];
        const u32 k0 = (u32)k, k1 = (u32)(k >> 32);
        acc[e][h][0] = mad_wide(k0, x0, acc[e][h][0]);
        acc[e][h][1] = mad_wide(k0, x1, acc[e][h][1]);
        acc[e][h][2] = mad_wide(k0, x2, acc[e][h][2]);
        acc[e][h][3] = mad_wide(k1, x0, acc[e][h][3]);
        acc[e][h][4] = mad_wide(k1, x1, acc[e][h][4]);
        acc[e][h][5] = mad_wide(k1, x2, acc[e][h][5]);
      }
    }
  };
  // the row pass of one digit (TWO = false) or of two digits side by side (second row image at + ROW4_LDS)
  auto row_pass = [&](u64 (&xa)[4], u64 (&xb)[4], auto two_tag) {
    constexpr bool TWO = decltype(two_tag)::value;
#ifdef FHE_HIP_DEBUG
#pragma unroll
    for (int k = 0; k < 4; ++k) { DBG_RANGE(xa[k], 2 * P.twop, 11); if constexpr (TWO) DBG_RANGE(xb[k], 2 * P.twop, 11); }
#endif
    fwd_phase_w<SP, 2>(xa, wA, P);                          // spans 128, 64: {t + 64k}
#pragma unroll
    for (int k = 0; k < 4; ++k) pA[72 * k] = xa[k];
    if constexpr (TWO) {
      fwd_phase_w<SP, 2>(xb, wA, P);
#pragma unroll
      for (int k = 0; k < 4; ++k) pA[ROW4_LDS + 72 * k] = xb[k];
    }
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < 4; ++k) xa[k] = pB[18 * k];
    if constexpr (TWO) {
#pragma unroll
      for (int k = 0; k < 4; ++k) xb[k] = pB[ROW4_LDS + 18 * k];
    }
    wave_lds_sync();  // (all reads issued before the first write below: same wave, LDS operations complete in order)
    fwd_phase_w<SP, 2>(xa, wB, P);                          // spans 32, 16
#pragma unroll
    for (int k = 0; k < 4; ++k) pB[18 * k] = xa[k];
    if constexpr (TWO) {
      fwd_phase_w<SP, 2>(xb, wB, P);
#pragma unroll
      for (int k = 0; k < 4; ++k) pB[ROW4_LDS + 18 * k] = xb[k];
    }
    wave_lds_sync();
#pragma unroll
    for (int k = 0; k < 4; ++k) xa[k] = pC[4 * k];
    if constexpr (TWO) {
#pragma unroll
      for (int k = 0; k < 4; ++k) xb[k] = pC[ROW4_LDS + 4 * k];
    }
    wave_lds_sync();
    fwd_phase_w<SP, 2>(xa, wC, P);                          // spans 8, 4
#pragma unroll
    for (int k = 0; k < 4; ++k) pC[4 * k] = xa[k];
    if constexpr (TWO) {
      fwd_phase_w<SP, 2>(xb, wC, P);
#pragma unroll
      for (int k = 0; k < 4; ++k) pC[ROW4_LDS + 4 * k] = xb[k];
    }
    wave_lds_sync();
    {
      const ulonglong2 v0 = *reinterpret_cast<const ulonglong2 *>(pD), v1 = *reinterpret_cast<const ulonglong2 *>(pD + 2);
      xa[0] = v0.x; xa[1] = v0.y; xa[2] = v1.x; xa[3] = v1.y;
      if constexpr (TWO) {
        const ulonglong2 w0 = *reinterpret_cast<const ulonglong2 *>(pD + ROW4_LDS), w1 = *reinterpret_cast<const ulonglong2 *>(pD + ROW4_LDS + 2);
        xb[0] = w0.x; xb[1] = w0.y; xb[2] = w1.x; xb[3] = w1.y;
      }
    }
    wave_lds_sync();  // the row images are rewritten by the next digits' first exchange
    fwd_phase_w<SP, 2>(xa, wD, P);                          // spans 2, 1: {4t + k}, lazy in [0, 4p)
    if constexpr (TWO) fwd_phase_w<SP, 2>(xb, wD, P);
  };
  // the digits other than the diagonal one, in order: q -> q (q < i) or q + 1
  const bool has_diag = i < nd;
  const u32 nnd = has_diag ? nd - 1 : nd;
  auto dig = [&](u32 q) { const u32 qq = q < nnd ? q : nnd - 1; return (has_diag && qq >= i) ? qq + 1 : qq; };  // clamped: the tail re-requests the last digit
  u64 xa[4], xb[4], na[4], nb_[4];
  if (nnd) { load_digit(xa, dig(0)); load_digit(xb, dig(1)); }
  if (has_diag) {  // digit j == i is ct[2] limb i itself (NTT_i(INTT_i(.)) = identity): no row pass, taken as the row pass would leave it
    const u64 *s = a.diag + (u64)b * a.diag_bstride + (u64)i * N + row_off;   // wave-uniform base
    const ulonglong2 v0 = *reinterpret_cast<const ulonglong2 *>(&s[4 * t]), v1 = *reinterpret_cast<const ulonglong2 *>(&s[4 * t + 2]);
    const u64 xd[4] = {v0.x, v0.y, v1.x, v1.y};
    const Keys kd = load_keys(i);
    mac(xd, kd);
  }
  u32 q = 0;
  for (; q + 1 < nnd; q += 2) {
    // this pair's key words and the next pair's digits: requested before the row passes, which cover them
    const Keys ka_ = load_keys(dig(q)), kb_ = load_keys(dig(q + 1));
    load_digit(na, dig(q + 2));
    load_digit(nb_, dig(q + 3));
    row_pass(xa, xb, std::true_type{});
    mac(xa, ka_);
    mac(xb, kb_);
#pragma unroll
    for (int k = 0; k < 4; ++k) { xa[k] = na[k]; xb[k] = nb_[k]; }
  }
  if (q < nnd) {  // odd count: the last digit alone
    const Keys ka_ = load_keys(dig(q));
    row_pass(xa, xb, std::false_type{});
    mac(xa, ka_);
  }
  const u64 half = (u64)a.nout * N;
  u64 *o = a.out + ((u64)b * 2 * a.nout + i) * N + row_off + 4 * t;
  u64 r[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < 4; ++e) r[h][e] = relin_fold<SP>(acc[e][h], P);
  const u64 *c0 = a.add0 ? a.add0 + (u64)b * a.add_bstride + (u64)i * N + row_off + 4 * t : nullptr;
  const u64 *c1 = a.add1 ? a.add1 + (u64)b * a.add_bstride + (u64)i * N + row_off + 4 * t : nullptr;
  if (c0) {
    const ulonglong2 d0 = *reinterpret_cast<const ulonglong2 *>(c0), d1 = *reinterpret_cast<const ulonglong2 *>(c0 + 2);
    r[0][0] = csub(r[0][0] + d0.x, P.p); r[0][1] = csub(r[0][1] + d0.y, P.p); r[0][2] = csub(r[0][2] + d1.x, P.p); r[0][3] = csub(r[0][3] + d1.y, P.p);
  }
  if (c1) {
    const ulonglong2 d0 = *reinterpret_cast<const ulonglong2 *>(c1), d1 = *reinterpret_cast<const ulonglong2 *>(c1 + 2);
    r[1][0] = csub(r[1][0] + d0.x, P.p); r[1][1] = csub(r[1][1] + d0.y, P.p); r[1][2] = csub(r[1][2] + d1.x, P.p); r[1][3] = csub(r[1][3] + d1.y, P.p);
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int e = 0; e < 4; ++e) DBG_RANGE(r[h][e], P.p, 10);
    *reinterpret_cast<ulonglong2 *>(o + h * half) = make_ulonglong2(r[h][0], r[h][1]);
    *reinterpret_cast<ulonglong2 *>(o + h * half + 2) = make_ulonglong2(r[h][2], r[h][3]);
  }
}

// ---------------------------------------------------------------------------------------------
// One-pass NTT for N = 2^12 .. 2^14: a whole limb (R = N/256 <= 64 rows of 256) sits in one workgroup's LDS, so the limb is read
// from and written to HBM once (16 N bytes instead of the 32 N of the pass pair).  The kernel is the column pass on an
// R x 256 tile followed by the row pass on the same tile: the tile keeps each row in the row pass's own image (8 chunks of 32
// elements at stride 36), so the column pass's last write is the row pass's first read.  16 R threads, 16 elements each; in the
// row pass a half-wave owns a row (two rows one after the other), and only the column/row hand-over needs a workgroup barrier.
// Plain NTTs only (no epilogue, no gather): everything else keeps the pass pair.
// ---------------------------------------------------------------------------------------------
// Read-only tables read through the constant address space: inside a persistent loop (global stores before the next limb's
// loads) the compiler can no longer prove a plain uniform load unclobbered and would turn it into a vector load; twiddles that go
// into multiply-adds as scalar operands, the prime constants and the modulus map must stay scalar loads.
typedef const u64 __attribute__((address_space(4))) *cu64p;
typedef const u32 __attribute__((address_space(4))) *cu32p;
__device__ __forceinline__ cu64p as_const(const u64 *p) { return (cu64p)(uintptr_t)p; }
__device__ __forceinline__ cu32p as_const(const u32 *p) { return (cu32p)(uintptr_t)p; }
__device__ __forceinline__ PrimeC load_prime_const(const PrimeC *p) {
  const cu64p q = (cu64p)(uintptr_t)p;
  PrimeC P;
  P.p = q[0]; P.twop = q[1]; P.pinv = q[2]; P.r2 = q[3]; P.ninv_m = q[4];
  const u64 w = q[5];
  P.p1 = (u32)w; P.n32 = (u32)(w >> 32);
  return P;
}
static_assert(sizeof(PrimeC) == 48, "load_prime_const reads six words");

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter (its fence covers global
// memory), which in the persistent kernels would wait for the prefetched limb and for the row stores at every hand-over.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int LOGR> struct OneCfg {
  static constexpr int R = 1 << LOGR;
  static constexpr int LA = LOGR - 4;  // LOGR >= 4
  static constexpr int LB = 4;
  static constexpr int T = 16 * R;
  static constexpr int LDS_WORDS = R * ROW8_LDS;
};
__device__ __forceinline__ u32 tile_pos(u32 row, u32 c) { return row * ROW8_LDS + row8_pos(c); }

// Persistent form: workgroup w transforms the limbs w, w + gridDim.x, ... of the nl * batch limbs of the call (limb id = entry * nl
// + limb).  With a 2^14 limb only one workgroup fits a CU, so nothing else would cover its memory latency: the 16 elements per
// thread of the NEXT limb are requested in the middle of the current limb's row pass and land while it finishes.  Vector-memory
// operations retire in issue order, so the order below is the point: twiddles of the second row, then the next limb, then the
// stores of the first row - every wait then covers only what is needed next.
template <bool SP, int LOGR>
__global__ __launch_bounds__(OneCfg<LOGR>::T) void ntt_onepass_fwd_kernel(NttArgs a, u32 nl, u32 total) {
  using Cfg = OneCfg<LOGR>;
  constexpr int R = Cfg::R, LA = Cfg::LA, LB = Cfg::LB, T = Cfg::T;
  __shared__ __attribute__((aligned(16))) u64 lds[Cfg::LDS_WORDS];
  const u32 tid = threadIdx.x;
  const u64 N = 1ull << a.logN;
  const u32 u = tid >> 8, c = tid & 255;
  u64 pre[16];
  auto request = [&](u32 id) {
    const u32 b = __builtin_amdgcn_readfirstlane(id / nl), l = id - b * nl + a.l0;  // wave-uniform (the quotient comes out of the vector unit)
    const u64 *s = a.src + (u64)b * a.src_bstride + (u64)limb_source(a, l) * N;
    // element (row g + G k, column cc) of slot tid + rep T sits at tid + 256 (rep T / 256 + G k): one per-thread offset, the rest
    // is a wave-uniform base (scalar address arithmetic; sixteen per-thread addresses would not fit the register budget)
    if constexpr (LA > 0) {
      constexpr int G = R >> LA;
#pragma unroll
      for (int rep = 0; rep < (16 >> LA); ++rep) {
#pragma unroll
        for (int k = 0; k < (1 << LA); ++k) pre[rep * (1 << LA) + k] = (s + (u64)(rep * (T / 256) + G * k) * 256)[tid];
      }
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) pre[k] = (s + (u64)k * 256)[tid];
    }
  };
  u32 id = blockIdx.x;
  if (id >= total) return;
  request(id);
  while (true) {
    const u32 b = __builtin_amdgcn_readfirstlane(id / nl), l = id - b * nl + a.l0;  // wave-uniform (the quotient comes out of the vector unit)
    const u32 m = as_const(a.midx)[limb_mod(limb_div(l, a.mdiv, a.mdiv_inv), a.mper, a.mper_inv)];
    const PrimeC P = load_prime_const(a.primes + m);
    const u64 *__restrict__ tw = a.tw + (u64)m * N;
    const cu64p twc = as_const(tw);
    u64 *d = a.dst + (u64)b * a.dst_bstride + (u64)l * N;
    {
      // column pass, stages 0 .. LOGR-1 (as ntt_cols_kernel with a 256-column tile)
      u64 e[16];
      if constexpr (LA > 0) {
        u64 wB[15];
        load_tw<LB>(wB, tw, (1u << LA) + u);
        constexpr int G = R >> LA;
#pragma unroll
        for (int rep = 0; rep < (16 >> LA); ++rep) {
          const u32 slot = tid + rep * T, g = slot >> 8, cc = slot & 255;
          u64 f[1 << LA];
#pragma unroll
          for (int k = 0; k < (1 << LA); ++k) { f[k] = pre[rep * (1 << LA) + k]; DBG_RANGE(f[k], 2 * P.twop, 6); }
          fwd_phase_ua<SP, LA>(f, twc, P);
#pragma unroll
          for (int k = 0; k < (1 << LA); ++k) lds[tile_pos(g + G * k, cc)] = f[k];
        }
        lds_barrier();
#pragma unroll
        for (int k = 0; k < 16; ++k) e[k] = lds[tile_pos((u << LB) + k, c)];
        fwd_phase_w<SP, LB>(e, wB, P);
      } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) { e[k] = pre[k]; DBG_RANGE(e[k], 2 * P.twop, 6); }
        fwd_phase_ua<SP, LB>(e, twc, P);
      }
#pragma unroll
      for (int k = 0; k < 16; ++k) lds[tile_pos((u << LB) + k, c)] = e[k];
    }
    lds_barrier();
    // row pass, stages LOGR .. logN-1 (as ntt_rows8_fwd_kernel, rows read from the tile): a half-wave owns a row, two rows each.
    // The thread id goes through an opaque move once per limb: the row pass's ~40 per-thread offsets (twiddle indices, LDS and
    // store addresses) are the same for every limb, and hoisted out of the loop they would be spilled to scratch and reloaded
    // between the prefetch loads, each reload draining the memory counter; recomputing them costs ~2 % of a limb's instructions.
    u32 tid_ = tid;
    asm volatile("" : "+v"(tid_));
    const u32 t = tid_ & 31, cq = t >> 2, r = t & 3, p0 = row8_pos(8 * t);
    const u32 row0 = tid_ >> 5, row1 = row0 + T / 32;
    u64 wA[7], wB[7], wC0[3], wC1[3];
    auto row_twiddles = [&](u32 row) {
      const u32 baseA = (1u << LOGR) + row;
      load_tw<3>(wA, tw, baseA);
      load_tw<3>(wB, tw, (baseA << 3) + cq);
      load_tw<2>(wC0, tw, (baseA << 6) + 2 * t);
      load_tw<2>(wC1, tw, (baseA << 6) + 2 * t + 1);
    };
    auto row_compute = [&](u32 row) {
      u64 *L = lds + row * ROW8_LDS;
      u64 e[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) e[k] = L[k * 36 + t];
      fwd_phase_w<SP, 3>(e, wA, P);
#pragma unroll
      for (int k = 0; k < 8; ++k) L[k * 36 + t] = e[k];
      wave_lds_sync();
#pragma unroll
      for (int k = 0; k < 8; ++k) e[k] = L[cq * 36 + r + 4 * k];
      fwd_phase_w<SP, 3>(e, wB, P);
#pragma unroll
      for (int k = 0; k < 8; ++k) L[cq * 36 + r + 4 * k] = e[k];
      wave_lds_sync();
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(&L[p0 + k]);
        e[k] = v.x; e[k + 1] = v.y;
      }
      {
        u64 g0[4] = {e[0], e[1], e[2], e[3]}, g1[4] = {e[4], e[5], e[6], e[7]};
        fwd_phase_w<SP, 2>(g0, wC0, P);
        fwd_phase_w<SP, 2>(g1, wC1, P);
#pragma unroll
        for (int k = 0; k < 4; ++k) { e[k] = canon4(g0[k], P); e[4 + k] = canon4(g1[k], P); }
      }
      wave_lds_sync();
#pragma unroll
      for (int k = 0; k < 8; k += 2) *reinterpret_cast<ulonglong2 *>(&L[p0 + k]) = make_ulonglong2(e[k], e[k + 1]);
      wave_lds_sync();
    };
    auto row_store = [&](u32 row, u32 it) {
      const u64 *L = lds + row * ROW8_LDS;
      u64 *dr = d + (u64)it * (T / 32) * 256;  // uniform part of the row address
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const u32 el = 64 * i + 2 * t;
        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(&L[row8_pos(el)]);
        DBG_RANGE(v.x, P.p, 2); DBG_RANGE(v.y, P.p, 2);
        *reinterpret_cast<ulonglong2 *>(&dr[row0 * 256 + el]) = v;
      }
    };
    const u32 next = id + gridDim.x;
    row_twiddles(row0);
    __builtin_amdgcn_sched_barrier(0);
    row_compute(row0);
    __builtin_amdgcn_sched_barrier(0);
    row_twiddles(row1);
    __builtin_amdgcn_sched_barrier(0);
    request(next < total ? next : id);  // the last limb re-requests itself (one code path: the loads stay where they are put)
    __builtin_amdgcn_sched_barrier(0);
    row_store(row0, 0);
    __builtin_amdgcn_sched_barrier(0);
    row_compute(row1);
    row_store(row1, 1);
    if (next >= total) break;
    id = next;
    lds_barrier();  // the tile is rewritten by the next limb's column pass
  }
}

// Inverse: row pass into the tile, then the column pass + N^-1.  Persistent like the forward kernel; the next limb (both rows of
// every half-wave, 16 elements per thread) is requested at the start of the column pass, behind that pass's own twiddles.
template <bool SP, int LOGR>
__global__ __launch_bounds__(OneCfg<LOGR>::T) void ntt_onepass_inv_kernel(NttArgs a, u32 nl, u32 total) {
  using Cfg = OneCfg<LOGR>;
  constexpr int R = Cfg::R, LA = Cfg::LA, LB = Cfg::LB, T = Cfg::T;
  __shared__ __attribute__((aligned(16))) u64 lds[Cfg::LDS_WORDS];
  const u32 tid = threadIdx.x;
  const u64 N = 1ull << a.logN;
  const u32 u = tid >> 8, c = tid & 255;
  ulonglong2 pre[8];
  auto request = [&](u32 id) {
    const u32 b = __builtin_amdgcn_readfirstlane(id / nl), l = id - b * nl + a.l0;  // wave-uniform (the quotient comes out of the vector unit)
    const u64 *s = a.src + (u64)b * a.src_bstride + (u64)limb_source(a, l) * N;
    const u32 off = (tid >> 5) * 256 + 2 * (tid & 31);
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i) pre[4 * it + i] = *reinterpret_cast<const ulonglong2 *>(&(s + (u64)it * (T / 32) * 256 + 64 * i)[off]);
  };
  u32 id = blockIdx.x;
  if (id >= total) return;
  request(id);
  while (true) {
    const u32 b = __builtin_amdgcn_readfirstlane(id / nl), l = id - b * nl + a.l0;
    const u32 m = as_const(a.midx)[limb_mod(limb_div(l, a.mdiv, a.mdiv_inv), a.mper, a.mper_inv)];
    const PrimeC P = load_prime_const(a.primes + m);
    const u64 *__restrict__ tw = a.tw + (u64)m * N;
    const cu64p twc = as_const(tw);
    u64 *d = a.dst + (u64)b * a.dst_bstride + (u64)l * N;
    const u64 nscale = a.inv_scale ? as_const(a.inv_scale)[m] : P.ninv_m;
    {
      // row pass, spans 1, 2 | 4, 8, 16 | 32, 64, 128 (as ntt_rows8_inv_kernel), left in the tile.  Opaque thread id: see the forward kernel.
      u32 tid_ = tid;
      asm volatile("" : "+v"(tid_));
      const u32 t = tid_ & 31, cq = t >> 2, r = t & 3, p0 = row8_pos(8 * t);
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const u32 row = (tid_ >> 5) + it * (T / 32);
        u64 *L = lds + row * ROW8_LDS;
        const u32 baseA = (1u << LOGR) + row;
        u64 wA[7], wB[7], wC0[3], wC1[3];  // all twiddles of the row requested up front, in the order they are used
        load_tw<2>(wC0, tw, (baseA << 6) + 2 * t);
        load_tw<2>(wC1, tw, (baseA << 6) + 2 * t + 1);
        load_tw<3>(wB, tw, (baseA << 3) + cq);
        load_tw<3>(wA, tw, baseA);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const ulonglong2 v = pre[4 * it + i];
          DBG_RANGE(v.x, P.twop, 4); DBG_RANGE(v.y, P.twop, 4);
          *reinterpret_cast<ulonglong2 *>(&L[row8_pos(64 * i + 2 * t)]) = v;
        }
        wave_lds_sync();
        u64 e[8];
#pragma unroll
        for (int k = 0; k < 8; k += 2) {
          const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(&L[p0 + k]);
          e[k] = v.x; e[k + 1] = v.y;
        }
        {
          u64 g0[4] = {e[0], e[1], e[2], e[3]}, g1[4] = {e[4], e[5], e[6], e[7]};
          inv_phase_w<SP, 2>(g0, wC0, P);
          inv_phase_w<SP, 2>(g1, wC1, P);
#pragma unroll
          for (int k = 0; k < 4; ++k) { e[k] = g0[k]; e[4 + k] = g1[k]; }
        }
        wave_lds_sync();
#pragma unroll
        for (int k = 0; k < 8; k += 2) *reinterpret_cast<ulonglong2 *>(&L[p0 + k]) = make_ulonglong2(e[k], e[k + 1]);
        wave_lds_sync();
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = L[cq * 36 + r + 4 * k];
        inv_phase_w<SP, 3>(e, wB, P);
#pragma unroll
        for (int k = 0; k < 8; ++k) L[cq * 36 + r + 4 * k] = e[k];
        wave_lds_sync();
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = L[k * 36 + t];
        inv_phase_w<SP, 3>(e, wA, P);
#pragma unroll
        for (int k = 0; k < 8; ++k) { DBG_RANGE(e[k], P.twop, 5); L[k * 36 + t] = e[k]; }
      }
    }
    lds_barrier();
    const u32 next = id + gridDim.x;
    {
      // column pass (as ntt_cols_kernel<INVERSE> with a 256-column tile) + N^-1; its twiddles first, then the next limb
      u64 wcol[15];
      load_tw<LB>(wcol, tw, (1u << LA) + u);
      __builtin_amdgcn_sched_barrier(0);
      request(next < total ? next : id);  // the last limb re-requests itself (one code path)
      __builtin_amdgcn_sched_barrier(0);
      u64 e[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) e[k] = lds[tile_pos((u << LB) + k, c)];
      if constexpr (LA > 0) inv_phase_w<SP, LB>(e, wcol, P);
      else inv_phase_w_final<SP, LB>(e, wcol, P, nscale);
      if constexpr (LA > 0) {
#pragma unroll
        for (int k = 0; k < 16; ++k) lds[tile_pos((u << LB) + k, c)] = e[k];
        lds_barrier();
        constexpr int G = R >> LA;
#pragma unroll
        for (int rep = 0; rep < (16 >> LA); ++rep) {
          const u32 slot = tid + rep * T, g = slot >> 8, cc = slot & 255;
          u64 f[1 << LA];
#pragma unroll
          for (int k = 0; k < (1 << LA); ++k) f[k] = lds[tile_pos(g + G * k, cc)];
          inv_phase_final<SP, LA>(f, twc, 1u, P, nscale);
#pragma unroll
          for (int k = 0; k < (1 << LA); ++k) {
            const u64 v = f[k];
            DBG_RANGE(v, P.p, 9);
            (d + (u64)(rep * (T / 256) + G * k) * 256)[tid] = v;
          }
        }
      } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) { DBG_RANGE(e[k], P.p, 9); (d + (u64)k * 256)[tid] = e[k]; }
      }
    }
    if (next >= total) break;
    id = next;
    lds_barrier();  // the tile is rewritten by the next limb's row pass
  }
}

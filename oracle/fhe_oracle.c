/*
 * fhe_oracle.c — CPU restatement of fHE's RNS-CKKS polynomial hot path.  See fhe_oracle.h.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED at the yell boundary (header comment of fhe_oracle.h).
 * All file:line citations are relative to /root/reference.
 */
#include "fhe_oracle.h"
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef unsigned __int128 u128;

struct orc_ctx {
  uint32_t logN, L, K;
  size_t N;
  u64 *P;            /* [L+K] */
  u64 *psi;          /* [L+K] */
  u64 **w, **wp;     /* forward twiddles psi^brv(k), and Shoup companions, [L+K][N] */
  u64 **wi, **wip;   /* inverse twiddles psi^-brv(k) */
  u64 *ninv, *ninvp; /* N^-1 mod p and Shoup companion */
  u64 *R64, *R64p;   /* 2^64 mod p and Shoup companion (128-bit reduction) */
  /* BaseConverter tables, base_converter.cpp:49-59 */
  u64 *hat_pi_invmod_pi;  /* [K] */
  u64 *neg_hat_pi_mod_qj; /* [L][K] */
  u64 *P_invmod_qj;       /* [L] */
  u64 *P_mod_qj;          /* [L] */
  u64 *hat_qj_invmod_qj;  /* [L-1][L] */
  u64 *hat_qj_mod_pi;     /* [L-1][K][L] */
  u64 *Q_mod_pi;          /* [L-1][K] */
};

/* ------------------------------------------------------------------------------------------ */
/* yell::ops restated                                                         */
/* ------------------------------------------------------------------------------------------ */
u64 orc_mulmod(u64 a, u64 b, u64 p) { return (u64)(((u128)a * b) % p); }
u64 orc_addmod(u64 a, u64 b, u64 p) { u64 s = (a % p) + (b % p); return s >= p ? s - p : s; }
u64 orc_powmod(u64 a, u64 e, u64 p) {
  u64 r = 1; a %= p;
  while (e) { if (e & 1) r = orc_mulmod(r, a, p); a = orc_mulmod(a, a, p); e >>= 1; }
  return r;
}
u64 orc_invmod(u64 a, u64 p) { return orc_powmod(a, p - 2, p); } /* math::inv_mod_prime */
u64 orc_shoupify(u64 y, u64 p) { return (u64)(((u128)y << 64) / p); }
static inline u64 mulhi(u64 a, u64 b) { return (u64)(((u128)a * b) >> 64); }
/* x arbitrary 64-bit, y < p: result in [0, 2p) */
static inline u64 shoup_lazy(u64 x, u64 y, u64 yp, u64 p) { return x * y - mulhi(x, yp) * p; }
u64 orc_mulmod_shoup(u64 x, u64 y, u64 yp, u64 p) { u64 r = shoup_lazy(x, y, yp, p); return r >= p ? r - p : r; }

/* barret_reduction(gT*, cm): 128-bit lazy sum -> [0,p) (evaluator.cpp:640, base_converter.cpp:346) */
static inline u64 reduce128(const orc_ctx *c, u128 x, uint32_t idx) {
  u64 p = c->P[idx];
  u64 hi = (u64)(x >> 64), lo = (u64)x;
  u64 t = shoup_lazy(hi, c->R64[idx], c->R64p[idx], p); /* hi * 2^64 mod p, in [0,2p) */
  u128 s = (u128)t + (lo % p);
  return (u64)(s % p);
}

int orc_is_prime(u64 n) {
  if (n < 2) return 0;
  static const u64 small[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
  for (size_t i = 0; i < 12; ++i) { if (n == small[i]) return 1; if (n % small[i] == 0) return 0; }
  u64 d = n - 1; int r = 0;
  while (!(d & 1)) { d >>= 1; ++r; }
  for (size_t i = 0; i < 12; ++i) { /* deterministic for n < 2^64 */
    u64 x = orc_powmod(small[i], d, n);
    if (x == 1 || x == n - 1) continue;
    int comp = 1;
    for (int k = 1; k < r; ++k) { x = orc_mulmod(x, x, n); if (x == n - 1) { comp = 0; break; } }
    if (comp) return 0;
  }
  return 1;
}

void orc_gen_primes(uint32_t logN, uint32_t count, u64 *primes, u64 *psi) {
  u64 k = (1ull << 30) - 1;
  for (uint32_t n = 0; n < count; --k) {
    u64 p = (k << 32) + 1;
    if (!orc_is_prime(p)) continue;
    u64 x = 2;
    while (orc_powmod(x, (p - 1) >> 1, p) != p - 1) ++x; /* smallest quadratic non-residue */
    primes[n] = p;
    psi[n] = orc_powmod(x, (p - 1) >> (logN + 1), p);
    ++n;
  }
}

static inline size_t brv(size_t x, uint32_t bits) {
  size_t r = 0;
  for (uint32_t i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
  return r;
}

/* fHE::product_of (include/fHE/yell.hpp:9-21): product of P[i] for i in idx[0..n) skipping `skip`, mod P[m] */
static u64 product_of(const orc_ctx *c, uint32_t first, uint32_t n, int skip, uint32_t m) {
  u64 prod = 1;
  for (uint32_t i = 0; i < n; ++i) {
    if ((int)i == skip) continue;
    prod = orc_mulmod(prod, c->P[first + i] % c->P[m], c->P[m]);
  }
  return prod;
}

orc_ctx *orc_ctx_create(uint32_t logN, uint32_t L, uint32_t K, const u64 *primes, const u64 *psi) {
  orc_ctx *c = (orc_ctx *)calloc(1, sizeof(orc_ctx));
  c->logN = logN; c->L = L; c->K = K; c->N = (size_t)1 << logN;
  const uint32_t M = L + K; const size_t N = c->N;
  c->P = (u64 *)malloc(M * 8); c->psi = (u64 *)malloc(M * 8);
  memcpy(c->P, primes, M * 8); memcpy(c->psi, psi, M * 8);
  c->w = (u64 **)calloc(M, sizeof(u64 *)); c->wp = (u64 **)calloc(M, sizeof(u64 *));
  c->wi = (u64 **)calloc(M, sizeof(u64 *)); c->wip = (u64 **)calloc(M, sizeof(u64 *));
  c->ninv = (u64 *)malloc(M * 8); c->ninvp = (u64 *)malloc(M * 8);
  c->R64 = (u64 *)malloc(M * 8); c->R64p = (u64 *)malloc(M * 8);
  for (uint32_t m = 0; m < M; ++m) {
    u64 p = c->P[m], ps = c->psi[m], psinv = orc_invmod(ps, p);
    c->w[m] = (u64 *)malloc(N * 8); c->wp[m] = (u64 *)malloc(N * 8);
    c->wi[m] = (u64 *)malloc(N * 8); c->wip[m] = (u64 *)malloc(N * 8);
    u64 f = 1, g = 1;
    for (size_t k = 0; k < N; ++k) { /* power k stored at bit-reversed position */
      size_t r = brv(k, logN);
      c->w[m][r] = f; c->wi[m][r] = g;
      f = orc_mulmod(f, ps, p); g = orc_mulmod(g, psinv, p);
    }
    for (size_t k = 0; k < N; ++k) { c->wp[m][k] = orc_shoupify(c->w[m][k], p); c->wip[m][k] = orc_shoupify(c->wi[m][k], p); }
    c->ninv[m] = orc_invmod((u64)N % p, p); c->ninvp[m] = orc_shoupify(c->ninv[m], p);
    c->R64[m] = (u64)((((u128)1) << 64) % p); c->R64p[m] = orc_shoupify(c->R64[m], p);
  }
  /* BaseConverter::Impl::Impl(), base_converter.cpp:122-187 */
  c->hat_pi_invmod_pi = (u64 *)calloc(K ? K : 1, 8);
  c->neg_hat_pi_mod_qj = (u64 *)calloc((size_t)L * K + 1, 8);
  c->P_invmod_qj = (u64 *)calloc(L, 8);
  c->P_mod_qj = (u64 *)calloc(L, 8);
  size_t Lm1 = L > 1 ? L - 1 : 0;
  c->hat_qj_invmod_qj = (u64 *)calloc(Lm1 * L + 1, 8);
  c->hat_qj_mod_pi = (u64 *)calloc(Lm1 * K * L + 1, 8);
  c->Q_mod_pi = (u64 *)calloc(Lm1 * K + 1, 8);
  for (uint32_t k = 0; k < K; ++k) {
    uint32_t cm = L + k;
    c->hat_pi_invmod_pi[k] = orc_invmod(product_of(c, L, K, (int)k, cm), c->P[cm]);            /* :132-133 */
    for (uint32_t l = 0; l < L; ++l) c->neg_hat_pi_mod_qj[l * K + k] = c->P[l] - product_of(c, L, K, (int)k, l); /* :135-136 */
  }
  for (uint32_t n = 2; n <= L; ++n) {
    uint32_t l = n - 2;
    for (uint32_t j = 0; j < n; ++j) c->hat_qj_invmod_qj[l * L + j] = orc_invmod(product_of(c, 0, n, (int)j, j), c->P[j]); /* :141-151 */
    for (uint32_t k = 0; k < K; ++k) {
      for (uint32_t j = 0; j < n; ++j) c->hat_qj_mod_pi[((size_t)l * K + k) * L + j] = product_of(c, 0, n, (int)j, L + k); /* :155-167 */
      c->Q_mod_pi[l * K + k] = product_of(c, 0, n, -1, L + k);                                /* :178-186 */
    }
  }
  for (uint32_t j = 0; j < L; ++j) {                                                          /* :170-174 */
    u64 Pm = product_of(c, L, K, -1, j);
    c->P_mod_qj[j] = Pm; c->P_invmod_qj[j] = orc_invmod(Pm, c->P[j]);
  }
  return c;
}

void orc_ctx_destroy(orc_ctx *c) {
  if (!c) return;
  for (uint32_t m = 0; m < c->L + c->K; ++m) { free(c->w[m]); free(c->wp[m]); free(c->wi[m]); free(c->wip[m]); }
  free(c->w); free(c->wp); free(c->wi); free(c->wip); free(c->ninv); free(c->ninvp); free(c->R64); free(c->R64p);
  free(c->P); free(c->psi);
  free(c->hat_pi_invmod_pi); free(c->neg_hat_pi_mod_qj); free(c->P_invmod_qj); free(c->P_mod_qj);
  free(c->hat_qj_invmod_qj); free(c->hat_qj_mod_pi); free(c->Q_mod_pi);
  free(c);
}
u64 orc_prime(const orc_ctx *c, uint32_t idx) { return c->P[idx]; }

const u64 *orc_bconv_table(const orc_ctx *c, int which, size_t *count) {
  size_t L = c->L, K = c->K, Lm1 = L > 1 ? L - 1 : 0;
  switch (which) {
    case 0: *count = K; return c->hat_pi_invmod_pi;
    case 1: *count = L * K; return c->neg_hat_pi_mod_qj;
    case 2: *count = L; return c->P_invmod_qj;
    case 3: *count = L; return c->P_mod_qj;
    case 4: *count = Lm1 * L; return c->hat_qj_invmod_qj;
    case 5: *count = Lm1 * K * L; return c->hat_qj_mod_pi;
    case 6: *count = Lm1 * K; return c->Q_mod_pi;
  }
  *count = 0; return NULL;
}

/* ------------------------------------------------------------------------------------------ */
/* a1: negacyclic NTT (yell::ntt<N>::forward/backward restated; Harvey lazy butterflies)        */
/* ------------------------------------------------------------------------------------------ */
void orc_ntt_forward(const orc_ctx *c, u64 *a, uint32_t idx) {
  const size_t N = c->N; const u64 p = c->P[idx], twop = 2 * p;
  const u64 *w = c->w[idx], *wp = c->wp[idx];
  for (size_t j = 0; j < N; ++j) a[j] %= p; /* inputs may be residues of another prime (evaluator.cpp:667-668) */
  size_t t = N;
  for (size_t m = 1; m < N; m <<= 1) {
    t >>= 1;
    for (size_t i = 0; i < m; ++i) {
      const u64 W = w[m + i], Wp = wp[m + i];
      u64 *x = a + 2 * i * t, *y = x + t;
      for (size_t j = 0; j < t; ++j) {
        u64 X = x[j]; X -= (X >= twop) ? twop : 0;
        u64 T = shoup_lazy(y[j], W, Wp, p);
        x[j] = X + T; y[j] = X - T + twop;
      }
    }
  }
  for (size_t j = 0; j < N; ++j) { u64 v = a[j]; v -= (v >= twop) ? twop : 0; v -= (v >= p) ? p : 0; a[j] = v; }
}

void orc_ntt_backward(const orc_ctx *c, u64 *a, uint32_t idx) {
  const size_t N = c->N; const u64 p = c->P[idx], twop = 2 * p;
  const u64 *w = c->wi[idx], *wp = c->wip[idx];
  for (size_t j = 0; j < N; ++j) a[j] %= p;
  size_t t = 1;
  for (size_t m = N; m > 1; m >>= 1) {
    size_t h = m >> 1, j1 = 0;
    for (size_t i = 0; i < h; ++i) {
      const u64 W = w[h + i], Wp = wp[h + i];
      for (size_t j = j1; j < j1 + t; ++j) {
        u64 U = a[j], V = a[j + t];
        u64 s = U + V; s -= (s >= twop) ? twop : 0;
        a[j] = s;
        a[j + t] = shoup_lazy(U - V + twop, W, Wp, p);
      }
      j1 += 2 * t;
    }
    t <<= 1;
  }
  for (size_t j = 0; j < N; ++j) a[j] = orc_mulmod_shoup(a[j], c->ninv[idx], c->ninvp[idx], p); /* N^-1 scaling */
}

void orc_ntt_forward_schoolbook(const orc_ctx *c, u64 *out, const u64 *in, uint32_t idx) {
  const size_t N = c->N; const u64 p = c->P[idx];
  for (size_t d = 0; d < N; ++d) { /* out[d] = sum_k in[k] * psi^((2*brv(d)+1)*k)  (src/yell.cpp:16-21) */
    u64 root = orc_powmod(c->psi[idx], 2 * brv(d, c->logN) + 1, p), f = 1, acc = 0;
    for (size_t k = 0; k < N; ++k) { acc = orc_addmod(acc, orc_mulmod(in[k] % p, f, p), p); f = orc_mulmod(f, root, p); }
    out[d] = acc;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* a5/a6: automorphisms                                                                         */
/* ------------------------------------------------------------------------------------------ */
void orc_apply_galois(const orc_ctx *c, u64 *poly, uint32_t nlimbs, u64 galois) { /* src/yell.cpp:4-25 */
  if (!poly || galois == 1) return;
  const size_t N = c->N; const uint32_t logn = c->logN;
  const u64 mask = ((u64)2 << logn) - 1;
  u64 *tmp = (u64 *)malloc(N * 8);
  for (uint32_t cm = 0; cm < nlimbs; ++cm) {
    u64 *src = poly + (size_t)cm * N;
    for (size_t d = 0; d < N; ++d) {
      u64 raw = (galois * (2 * brv(d, logn) + 1)) & mask;
      tmp[d] = src[brv((raw - 1) >> 1, logn)];
    }
    memcpy(src, tmp, N * 8);
  }
  free(tmp);
}

void orc_apply_galois_non_ntt(const orc_ctx *c, u64 *poly, uint32_t nlimbs, u64 galois) { /* src/yell.cpp:29-53 */
  if (!poly || galois == 1) return;
  const size_t N = c->N; const uint32_t logn = c->logN;
  const u64 mask = ((u64)1 << logn) - 1;
  u64 *tmp = (u64 *)malloc(N * 8);
  for (uint32_t cm = 0; cm < nlimbs; ++cm) {
    u64 *in = poly + (size_t)cm * N;
    u64 raw = 0;
    for (size_t d = 0; d < N; ++d, raw += galois) {
      u64 t = in[d];
      tmp[raw & mask] = ((raw >> logn) & 1) ? (t > 0 ? c->P[cm] - t : 0) : t; /* P[cm]: limb k uses modulus k (:47) */
    }
    memcpy(in, tmp, N * 8);
  }
  free(tmp);
}

u64 orc_rotation_galois(const orc_ctx *c, int offset, int dir_right) { /* rot_key.cpp:6-24 */
  const long nr_slots = (long)(c->N >> 1); const u64 m = (u64)c->N << 1;
  long off = dir_right ? offset : -(long)offset;
  int sign = off > 0;
  off = (off < 0 ? -off : off) % nr_slots;
  if (sign) off = nr_slots - off;
  u64 g = 1;
  while (off--) g = (g * 3) & (m - 1);
  return g;
}

/* ------------------------------------------------------------------------------------------ */
/* a8-a11: BaseConverter                                                                         */
/* ------------------------------------------------------------------------------------------ */
/* internal::do_approximated_basis_conversion, base_converter.cpp:313-349 (no overflow correction:
 * the P_mod_qj argument is unused there).  The 128-bit lazy sum is folded every 16 terms so the
 * result is the mathematically exact residue for any limb count. */
static void approx_conv(const orc_ctx *c, u64 *rop, const u64 *op, uint32_t n, const u64 *inv_tbl, const u64 *hat_tbl,
                        uint32_t src_first_idx, uint32_t target_idx) {
  const size_t N = c->N;
  u128 *y = (u128 *)calloc(N, sizeof(u128));
  for (uint32_t k = 0; k < n; ++k) {
    const uint32_t kcm = src_first_idx + k;
    const u64 pk = c->P[kcm], inv = inv_tbl[k], invs = orc_shoupify(inv, pk), hat = hat_tbl[k];
    const u64 *x = op + (size_t)k * N;
    for (size_t d = 0; d < N; ++d) y[d] += (u128)orc_mulmod_shoup(x[d], inv, invs, pk) * hat;
    if ((k & 15) == 15) for (size_t d = 0; d < N; ++d) y[d] = reduce128(c, y[d], target_idx);
  }
  for (size_t d = 0; d < N; ++d) rop[d] = reduce128(c, y[d], target_idx);
  free(y);
}

void orc_approx_convert_to_special_basis(const orc_ctx *c, u64 *rop, const u64 *op, uint32_t Li, uint32_t k) { /* :276-295 */
  const uint32_t lcm = Li - 2;
  approx_conv(c, rop, op, Li, c->hat_qj_invmod_qj + (size_t)lcm * c->L, c->hat_qj_mod_pi + ((size_t)lcm * c->K + k) * c->L, 0, c->L + k);
}

void orc_neg_approx_convert_to_normal_basis(const orc_ctx *c, u64 *rop, const u64 *sp, uint32_t j) { /* :297-311 */
  approx_conv(c, rop, sp, c->K, c->hat_pi_invmod_pi, c->neg_hat_pi_mod_qj + (size_t)j * c->K, c->L, j);
}

int orc_mod_up(const orc_ctx *c, u64 *rop, const u64 *op, uint32_t Li) { /* :189-203 */
  if (!rop || Li > c->L) return -1;
  if (Li <= 1) return -2; /* std::invalid_argument("nmoduli <= 1"), :284-285 */
  const size_t N = c->N;
  memcpy(rop, op, (size_t)Li * N * 8);
  for (uint32_t k = 0; k < c->K; ++k) orc_approx_convert_to_special_basis(c, rop + (size_t)(Li + k) * N, op, Li, k);
  return 0;
}

int orc_mod_down(const orc_ctx *c, u64 *rop, const u64 *op, uint32_t L0) { /* :205-242 */
  if (!rop) return -1;
  const size_t N = c->N;
  const u64 *sp = op + (size_t)L0 * N; /* special limbs, power basis */
  for (uint32_t j = 0; j < L0; ++j) {
    u64 *r = rop + (size_t)j * N;
    orc_neg_approx_convert_to_normal_basis(c, r, sp, j);
    orc_ntt_forward(c, r, j);
  }
  for (uint32_t j = 0; j < L0; ++j) {
    const u64 p = c->P[j], pinv = c->P_invmod_qj[j], ps = orc_shoupify(pinv, p);
    u64 *dst = rop + (size_t)j * N; const u64 *src = op + (size_t)j * N;
    for (size_t d = 0; d < N; ++d) dst[d] = orc_mulmod_shoup(src[d] + dst[d], pinv, ps, p);
  }
  return 0;
}

int orc_rescale(const orc_ctx *c, u64 *rop, const u64 *op, uint32_t n) { /* :244-274 */
  if (!rop || n < 2) return -1;
  const size_t N = c->N;
  const u64 last_prime = c->P[n - 1];
  u64 *tail = (u64 *)malloc(N * 8);
  for (uint32_t j = 0; j + 1 < n; ++j) {
    memcpy(tail, op + (size_t)(n - 1) * N, N * 8);
    orc_ntt_forward(c, tail, j);
    const u64 p = c->P[j];
    const u64 qinv = orc_invmod(last_prime % p, p), qs = orc_shoupify(qinv, p); /* recomputed per call, :261 */
    u64 *dst = rop + (size_t)j * N; const u64 *head = op + (size_t)j * N;
    for (size_t d = 0; d < N; ++d) dst[d] = orc_mulmod_shoup(head[d] + p - tail[d], qinv, qs, p); /* floor-style, no rounding term */
  }
  free(tail);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* a12-a16: Evaluator                                                                            */
/* ------------------------------------------------------------------------------------------ */
void orc_cross_mult(const orc_ctx *c, u64 *d0, u64 *d1, u64 *d2, const u64 *b0, const u64 *a0, const u64 *b1,
                    const u64 *a1, uint32_t Li) { /* evaluator.cpp:621-647 */
  const size_t N = c->N;
  for (uint32_t cm = 0; cm < Li; ++cm) {
    const u64 p = c->P[cm];
    const size_t o = (size_t)cm * N;
    for (size_t d = 0; d < N; ++d) {
      u64 vb0 = b0[o + d], va0 = a0[o + d], vb1 = b1[o + d], va1 = a1[o + d];
      u128 v = (u128)vb0 * va1 + (u128)vb1 * va0;
      d1[o + d] = reduce128(c, v, cm);
      d0[o + d] = orc_mulmod(vb0, vb1, p);
      d2[o + d] = orc_mulmod(va0, va1, p);
    }
  }
}

void orc_relinearize(const orc_ctx *c, u64 *d0, u64 *d1, const u64 *d2, const u64 *beta, const u64 *alpha,
                     uint32_t Li) { /* evaluator.cpp:649-693 */
  const size_t N = c->N; const size_t L = c->L;
  u64 *pw = (u64 *)malloc((size_t)Li * N * 8);
  memcpy(pw, d2, (size_t)Li * N * 8);
  for (uint32_t j = 0; j < Li; ++j) orc_ntt_backward(c, pw + (size_t)j * N, j); /* ct2_power.backward(), :659-660 */
  u64 *ct2 = (u64 *)malloc(N * 8);
  u128 *acc0 = (u128 *)malloc(N * sizeof(u128)), *acc1 = (u128 *)malloc(N * sizeof(u128));
  for (uint32_t i = 0; i < Li; ++i) {
    memset(acc0, 0, N * sizeof(u128)); memset(acc1, 0, N * sizeof(u128));
    for (uint32_t j = 0; j < Li; ++j) {
      memcpy(ct2, pw + (size_t)j * N, N * 8);
      orc_ntt_forward(c, ct2, i); /* NTT under q_i of residues mod q_j, also when j == i (:668) */
      const u64 *bj = beta + ((size_t)j * L + i) * N, *aj = alpha + ((size_t)j * L + i) * N;
      for (size_t d = 0; d < N; ++d) { acc0[d] += (u128)bj[d] * ct2[d]; acc1[d] += (u128)aj[d] * ct2[d]; } /* lazy_muladd */
      if ((j & 15) == 15) for (size_t d = 0; d < N; ++d) { acc0[d] = reduce128(c, acc0[d], i); acc1[d] = reduce128(c, acc1[d], i); }
    }
    u64 *r0 = d0 + (size_t)i * N, *r1 = d1 + (size_t)i * N;
    for (size_t d = 0; d < N; ++d) { r0[d] = reduce128(c, acc0[d] + r0[d], i); r1[d] = reduce128(c, acc1[d] + r1[d], i); } /* :682-691 */
  }
  free(pw); free(ct2); free(acc0); free(acc1);
}

void orc_rescale_ntt(const orc_ctx *c, u64 *r0, u64 *r1, u64 *ct0, u64 *ct1, uint32_t Li) { /* evaluator.cpp:695-710 */
  const size_t N = c->N;
  orc_ntt_backward(c, ct0 + (size_t)(Li - 1) * N, Li - 1);
  orc_rescale(c, r0, ct0, Li);
  orc_ntt_backward(c, ct1 + (size_t)(Li - 1) * N, Li - 1);
  orc_rescale(c, r1, ct1, Li);
}

int orc_mul_relin_rescale(const orc_ctx *c, u64 *out_b, u64 *out_a, const u64 *c0b, const u64 *c0a, const u64 *c1b,
                          const u64 *c1a, uint32_t Li, const u64 *beta, const u64 *alpha) { /* evaluator.cpp:119-126 */
  if (Li < 2 || Li > c->L) return -1;
  const size_t n = (size_t)Li * c->N;
  u64 *d0 = (u64 *)malloc(n * 8), *d1 = (u64 *)malloc(n * 8), *d2 = (u64 *)malloc(n * 8);
  orc_cross_mult(c, d0, d1, d2, c0b, c0a, c1b, c1a, Li);
  orc_relinearize(c, d0, d1, d2, beta, alpha, Li);
  orc_rescale_ntt(c, out_b, out_a, d0, d1, Li);
  free(d0); free(d1); free(d2);
  return 0;
}

void orc_add(const orc_ctx *c, u64 *rop, const u64 *op, uint32_t nlimbs) { /* poly operator+=, evaluator.cpp:284-285 */
  const size_t N = c->N;
  for (uint32_t cm = 0; cm < nlimbs; ++cm) {
    const u64 p = c->P[cm];
    for (size_t d = 0; d < N; ++d) { size_t o = (size_t)cm * N + d; rop[o] = orc_addmod(rop[o], op[o], p); }
  }
}

void orc_multiply_plain(const orc_ctx *c, u64 *bx, u64 *ax, const u64 *pt, uint32_t nlimbs) { /* evaluator.cpp:88-111 */
  const size_t N = c->N;
  for (uint32_t cm = 0; cm < nlimbs; ++cm) {
    const u64 p = c->P[cm];
    for (size_t d = 0; d < N; ++d) {
      size_t o = (size_t)cm * N + d;
      bx[o] = orc_mulmod(bx[o], pt[o], p); ax[o] = orc_mulmod(ax[o], pt[o], p);
    }
  }
}

int orc_rotate(const orc_ctx *c, u64 *bx, u64 *ax, uint32_t Li, u64 galois, const u64 *rk_beta, const u64 *rk_alpha) {
  /* Impl::rotate_slots_inplace evaluator.cpp:299-311 + apply_rotation_key :343-399 */
  if (!bx || !ax) return -1;
  if (galois == 0) return 0; /* :303-304 */
  if (Li > c->L) return -1;  /* "Too many moduli." :353-354 */
  if (Li <= 1) return -2;    /* mod_up throws, base_converter.cpp:284-285 */
  const size_t N = c->N; const uint32_t K = c->K, L = c->L;
  orc_apply_galois(c, bx, Li, galois);
  orc_apply_galois(c, ax, Li, galois);
  for (uint32_t i = 0; i < Li; ++i) orc_ntt_backward(c, ax + (size_t)i * N, i); /* d[1]->backward(), :360 */
  const size_t ext = (size_t)(Li + K) * N;
  u64 *e0 = (u64 *)malloc(ext * 8), *e1 = (u64 *)malloc(ext * 8);
  orc_mod_up(c, e0, ax, Li);
  for (uint32_t cm = 0; cm < Li + K; ++cm) orc_ntt_forward(c, e0 + (size_t)cm * N, cm < Li ? cm : L + (cm - Li)); /* :363-366 */
  memcpy(e1, e0, ext * 8);
  for (uint32_t cm = 0; cm < Li + K; ++cm) { /* :373-384 */
    const uint32_t idx = cm < Li ? cm : L + (cm - Li);
    const u64 p = c->P[idx];
    const u64 *b = rk_beta + (size_t)idx * N, *a = rk_alpha + (size_t)idx * N;
    u64 *o0 = e0 + (size_t)cm * N, *o1 = e1 + (size_t)cm * N;
    for (size_t d = 0; d < N; ++d) { o0[d] = orc_mulmod(o0[d], b[d], p); o1[d] = orc_mulmod(o1[d], a[d], p); }
  }
  for (uint32_t k = 0; k < K; ++k) { /* :386-392 */
    orc_ntt_backward(c, e0 + (size_t)(Li + k) * N, L + k);
    orc_ntt_backward(c, e1 + (size_t)(Li + k) * N, L + k);
  }
  u64 *cpy = (u64 *)malloc((size_t)Li * N * 8);
  memcpy(cpy, bx, (size_t)Li * N * 8); /* :394 */
  orc_mod_down(c, bx, e0, Li);
  orc_mod_down(c, ax, e1, Li);
  orc_add(c, bx, cpy, Li); /* :398 */
  free(e0); free(e1); free(cpy);
  return 0;
}

int orc_rotate_non_ntt(const orc_ctx *c, u64 *bx, u64 *ax, uint32_t Li, u64 galois, const u64 *rk_beta, const u64 *rk_alpha) {
  /* Impl::rotate_slots_inplace_non_ntt evaluator.cpp:327-339 + apply_rotation_key_non_ntt :520-574, literally: the
   * ciphertext is in the power basis, every limb of d_rotk goes NTT -> key product -> INTT, and approximated_mod_down
   * (base_converter.cpp:205-242) is then applied as written, i.e. its NTT-domain correction is added to power-basis limbs. */
  if (!bx || !ax) return -1;
  if (galois == 0) return 0; /* :331-332 */
  if (Li > c->L) return -1;  /* "Too many moduli." :530-531 */
  if (Li <= 1) return -2;    /* mod_up throws, base_converter.cpp:284-285 */
  const size_t N = c->N; const uint32_t K = c->K, L = c->L;
  orc_apply_galois_non_ntt(c, bx, Li, galois); /* :333-334 */
  orc_apply_galois_non_ntt(c, ax, Li, galois);
  const size_t ext = (size_t)(Li + K) * N;
  u64 *e0 = (u64 *)malloc(ext * 8), *e1 = (u64 *)malloc(ext * 8);
  orc_mod_up(c, e0, ax, Li); /* :537 */
  for (uint32_t cm = 0; cm < Li + K; ++cm) orc_ntt_forward(c, e0 + (size_t)cm * N, cm < Li ? cm : L + (cm - Li)); /* :539-542 */
  memcpy(e1, e0, ext * 8); /* :543 */
  for (uint32_t cm = 0; cm < Li + K; ++cm) { /* :549-560 */
    const uint32_t idx = cm < Li ? cm : L + (cm - Li);
    const u64 p = c->P[idx];
    const u64 *b = rk_beta + (size_t)idx * N, *a = rk_alpha + (size_t)idx * N;
    u64 *o0 = e0 + (size_t)cm * N, *o1 = e1 + (size_t)cm * N;
    for (size_t d = 0; d < N; ++d) { o0[d] = orc_mulmod(o0[d], b[d], p); o1[d] = orc_mulmod(o1[d], a[d], p); }
  }
  for (uint32_t cm = 0; cm < Li + K; ++cm) { /* :563-567 */
    const uint32_t idx = cm < Li ? cm : L + (cm - Li);
    orc_ntt_backward(c, e0 + (size_t)cm * N, idx);
    orc_ntt_backward(c, e1 + (size_t)cm * N, idx);
  }
  u64 *cpy = (u64 *)malloc((size_t)Li * N * 8);
  memcpy(cpy, bx, (size_t)Li * N * 8); /* :569 */
  orc_mod_down(c, bx, e0, Li); /* :571-572 */
  orc_mod_down(c, ax, e1, Li);
  orc_add(c, bx, cpy, Li); /* :573 */
  free(e0); free(e1); free(cpy);
  return 0;
}

int orc_rotate_mixed(const orc_ctx *c, u64 *bx, u64 *ax, uint32_t Li, u64 galois, const u64 *mk_beta, const u64 *mk_alpha) {
  /* rotate_slots_inplace(MixedRotKey) evaluator.cpp:313-325 + apply_rotation_key(MixedRotKey) :403-516 */
  if (!bx || !ax) return -1;
  if (galois == 0) return 0;
  if (Li < 1 || Li > c->L || c->K < 1) return -1;
  const size_t N = c->N; const uint32_t L = c->L, k = L; /* k = index_sp_prime(0) */
  const size_t kl = (size_t)(L + 1) * N;               /* one key polynomial */
  orc_apply_galois(c, bx, Li, galois);
  orc_apply_galois(c, ax, Li, galois);
  u128 *bacc = (u128 *)calloc((size_t)(Li + 1) * N, sizeof(u128)), *aacc = (u128 *)calloc((size_t)(Li + 1) * N, sizeof(u128));
  u64 *pw = (u64 *)malloc(N * 8), *v = (u64 *)malloc(N * 8);
  for (uint32_t i = 0; i < Li; ++i) {
    memcpy(pw, ax + (size_t)i * N, N * 8);
    orc_ntt_backward(c, pw, i);                                            /* :425-427 */
    for (uint32_t j = 0; j < Li + 1; ++j) {
      const uint32_t jidx = j < Li ? j : k;
      const u64 *src;
      if (j == i) src = ax + (size_t)j * N;                                /* :436-437 */
      else { memcpy(v, pw, N * 8); orc_ntt_forward(c, v, jidx); src = v; } /* :439-442 */
      const u64 *bk = mk_beta + (size_t)i * kl + (size_t)jidx * N, *ak = mk_alpha + (size_t)i * kl + (size_t)jidx * N;
      u128 *b_ = bacc + (size_t)j * N, *a_ = aacc + (size_t)j * N;
      for (size_t d = 0; d < N; ++d) { b_[d] += (u128)bk[d] * src[d]; a_[d] += (u128)ak[d] * src[d]; } /* :445-451 */
      if ((i & 15) == 15) for (size_t d = 0; d < N; ++d) { b_[d] = reduce128(c, b_[d], jidx); a_[d] = reduce128(c, a_[d], jidx); }
    }
  }
  u64 *ans_b = (u64 *)malloc((size_t)(Li + 1) * N * 8), *ans_a = (u64 *)malloc((size_t)(Li + 1) * N * 8);
  for (uint32_t j = 0; j < Li + 1; ++j) {                                  /* :456-473 */
    const uint32_t jidx = j < Li ? j : k;
    for (size_t d = 0; d < N; ++d) {
      ans_b[(size_t)j * N + d] = reduce128(c, bacc[(size_t)j * N + d], jidx);
      ans_a[(size_t)j * N + d] = reduce128(c, aacc[(size_t)j * N + d], jidx);
    }
  }
  orc_ntt_backward(c, ans_a + (size_t)Li * N, k);                          /* :479-480 */
  orc_ntt_backward(c, ans_b + (size_t)Li * N, k);
  const u64 qk = c->P[k];
  for (uint32_t j = 0; j < Li; ++j) {                                      /* :485-515 */
    const u64 qj = c->P[j], qinv = orc_invmod(qk % qj, qj), qs = orc_shoupify(qinv, qj);
    memcpy(v, ans_a + (size_t)Li * N, N * 8);
    orc_ntt_forward(c, v, j);
    for (size_t d = 0; d < N; ++d) ax[(size_t)j * N + d] = orc_mulmod_shoup(ans_a[(size_t)j * N + d] + qj - v[d], qinv, qs, qj);
    memcpy(v, ans_b + (size_t)Li * N, N * 8);
    orc_ntt_forward(c, v, j);
    for (size_t d = 0; d < N; ++d) {
      u64 t = orc_mulmod_shoup(ans_b[(size_t)j * N + d] + qj - v[d], qinv, qs, qj);
      bx[(size_t)j * N + d] = orc_addmod(bx[(size_t)j * N + d], t, qj);
    }
  }
  free(bacc); free(aacc); free(pw); free(v); free(ans_b); free(ans_a);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* plumbing: samplers and key generation (host only; streams are this repo's, not yell's)       */
/* ------------------------------------------------------------------------------------------ */
u64 orc_splitmix64(u64 *s) {
  u64 z = (*s += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

void orc_uniform_poly(const orc_ctx *c, u64 *poly, uint32_t first_idx, uint32_t nlimbs, u64 seed) {
  const size_t N = c->N;
  for (uint32_t k = 0; k < nlimbs; ++k) {
    u64 st = seed ^ (0xA5A5A5A5ull * (k + 1));
    const u64 p = c->P[first_idx + k];
    for (size_t d = 0; d < N; ++d) poly[(size_t)k * N + d] = orc_splitmix64(&st) % p;
  }
}

/* small signed coefficient v -> residue per limb; idx0 = global index of limb 0 */
static void set_small(const orc_ctx *c, u64 *poly, const int *v, uint32_t idx0, uint32_t nlimbs) {
  const size_t N = c->N;
  for (uint32_t k = 0; k < nlimbs; ++k) {
    const u64 p = c->P[idx0 + k];
    for (size_t d = 0; d < N; ++d) poly[(size_t)k * N + d] = v[d] >= 0 ? (u64)v[d] : p - (u64)(-v[d]);
  }
}
/* centred binomial, variance 10.5 ~ sigma 3.24 (context.hpp:11 sigma = 3.2); integer-only so it is reproducible */
static void sample_gauss(int *v, size_t N, u64 *st) {
  for (size_t d = 0; d < N; ++d) {
    u64 r = orc_splitmix64(st);
    v[d] = __builtin_popcountll(r & 0x1FFFFF) - __builtin_popcountll((r >> 21) & 0x1FFFFF);
  }
}
static void sample_zo(int *v, size_t N, u64 *st) { /* ZO_dist: 0 w.p. 1/2, +-1 w.p. 1/4 */
  for (size_t d = 0; d < N; ++d) { u64 r = orc_splitmix64(st) >> 62; v[d] = r == 0 ? 1 : (r == 1 ? -1 : 0); }
}
static void sample_hwt(int *v, size_t N, uint32_t h, u64 *st) {
  memset(v, 0, N * sizeof(int));
  for (uint32_t n = 0; n < h;) {
    u64 r = orc_splitmix64(st); size_t pos = (size_t)(r % N);
    if (v[pos]) continue;
    v[pos] = (r >> 63) ? 1 : -1; ++n;
  }
}

void orc_keygen_sk(const orc_ctx *c, u64 *sx, u64 *ext_sx, uint32_t hwt, u64 seed) { /* keys.hpp:9-24 */
  const size_t N = c->N; u64 st = seed;
  int *v = (int *)malloc(N * sizeof(int));
  sample_hwt(v, N, hwt, &st);
  set_small(c, sx, v, 0, c->L); set_small(c, ext_sx, v, c->L, c->K);
  for (uint32_t cm = 0; cm < c->L; ++cm) orc_ntt_forward(c, sx + (size_t)cm * N, cm);
  for (uint32_t k = 0; k < c->K; ++k) orc_ntt_forward(c, ext_sx + (size_t)k * N, c->L + k);
  free(v);
}

void orc_keygen_pk(const orc_ctx *c, u64 *pk_b, u64 *pk_a, const u64 *sx, u64 seed) { /* keys.hpp:39-47: b = -(e + a*s) */
  const size_t N = c->N; u64 st = seed;
  int *v = (int *)malloc(N * sizeof(int));
  sample_gauss(v, N, &st);
  set_small(c, pk_b, v, 0, c->L);
  orc_uniform_poly(c, pk_a, 0, c->L, seed ^ 0x1234567ull);
  for (uint32_t cm = 0; cm < c->L; ++cm) {
    const u64 p = c->P[cm]; u64 *b = pk_b + (size_t)cm * N; const u64 *a = pk_a + (size_t)cm * N, *s = sx + (size_t)cm * N;
    orc_ntt_forward(c, b, cm);
    for (size_t d = 0; d < N; ++d) { u64 t = orc_addmod(b[d], orc_mulmod(a[d], s[d], p), p); b[d] = t ? p - t : 0; }
  }
  free(v);
}

void orc_keygen_mult(const orc_ctx *c, u64 *beta, u64 *alpha, const u64 *sx, u64 seed) { /* mult_key.cpp:3-30 */
  const size_t N = c->N; const uint32_t L = c->L; u64 st = seed;
  int *v = (int *)malloc(N * sizeof(int));
  for (uint32_t j = 0; j < L; ++j) {
    u64 *bj = beta + (size_t)j * L * N, *aj = alpha + (size_t)j * L * N;
    orc_uniform_poly(c, aj, 0, L, seed ^ (0xABCDEFull * (j + 1)));
    sample_gauss(v, N, &st);
    set_small(c, bj, v, 0, L);
    for (uint32_t i = 0; i < L; ++i) {
      const u64 p = c->P[i]; u64 *b = bj + (size_t)i * N; const u64 *a = aj + (size_t)i * N, *s = sx + (size_t)i * N;
      orc_ntt_forward(c, b, i);
      for (size_t d = 0; d < N; ++d) {
        u64 t = orc_addmod(b[d], orc_mulmod(a[d], s[d], p), p);
        t = t ? p - t : 0;                                      /* -e - alpha*s */
        if (i == j) t = orc_addmod(t, orc_mulmod(s[d], s[d], p), p); /* + [s^2] on limb j only (:19-27) */
        b[d] = t;
      }
    }
  }
  free(v);
}

void orc_keygen_rot(const orc_ctx *c, u64 *beta, u64 *alpha, const u64 *sx, const u64 *ext_sx, u64 galois, u64 seed) {
  /* rot_key.cpp:23-79: beta = P*sigma_g(s) - e - alpha*s over L+K limbs */
  const size_t N = c->N; const uint32_t L = c->L, K = c->K; u64 st = seed;
  int *v = (int *)malloc(N * sizeof(int));
  u64 *rot = (u64 *)malloc((size_t)L * N * 8);
  memcpy(rot, sx, (size_t)L * N * 8);
  orc_apply_galois(c, rot, L, galois);
  orc_uniform_poly(c, alpha, 0, L + K, seed ^ 0x600Dull);
  sample_gauss(v, N, &st);
  set_small(c, beta, v, 0, L + K);
  for (uint32_t cm = 0; cm < L + K; ++cm) {
    const u64 p = c->P[cm]; u64 *b = beta + (size_t)cm * N; const u64 *a = alpha + (size_t)cm * N;
    const u64 *s = cm < L ? sx + (size_t)cm * N : ext_sx + (size_t)(cm - L) * N;
    orc_ntt_forward(c, b, cm);
    const u64 Pm = cm < L ? c->P_mod_qj[cm] : 0; /* lift_normal_moduli, :64-79 */
    for (size_t d = 0; d < N; ++d) {
      u64 t = orc_addmod(b[d], orc_mulmod(a[d], s[d], p), p);
      t = t ? p - t : 0;
      if (cm < L) t = orc_addmod(t, orc_mulmod(rot[(size_t)cm * N + d], Pm, p), p);
      b[d] = t;
    }
  }
  free(v); free(rot);
}

void orc_keygen_mixed_rot(const orc_ctx *c, u64 *beta, u64 *alpha, const u64 *sx, const u64 *ext_sx, u64 galois, u64 seed) {
  /* rot_key.cpp:81-120: beta_j = -(e + alpha_j * s) + [qK * sigma_g(s)] on limb j, over L normal limbs + P[L] */
  const size_t N = c->N; const uint32_t L = c->L; u64 st = seed;
  const size_t kl = (size_t)(L + 1) * N;
  int *v = (int *)malloc(N * sizeof(int));
  u64 *ext = (u64 *)malloc(kl * 8), *rot = (u64 *)malloc(kl * 8);
  memcpy(ext, sx, (size_t)L * N * 8); memcpy(ext + (size_t)L * N, ext_sx, N * 8);   /* :90-94 */
  memcpy(rot, ext, kl * 8);
  orc_apply_galois(c, rot, L + 1, galois);
  const u64 qK = c->P[L];
  for (uint32_t j = 0; j < L; ++j) {
    u64 *bj = beta + (size_t)j * kl, *aj = alpha + (size_t)j * kl;
    orc_uniform_poly(c, aj, 0, L + 1, seed ^ (0x51EDull * (j + 1)));
    sample_gauss(v, N, &st);
    set_small(c, bj, v, 0, L + 1);
    for (uint32_t cm = 0; cm < L + 1; ++cm) {
      const u64 p = c->P[cm]; u64 *b = bj + (size_t)cm * N; const u64 *a = aj + (size_t)cm * N, *s = ext + (size_t)cm * N;
      orc_ntt_forward(c, b, cm);
      for (size_t d = 0; d < N; ++d) {
        u64 t = orc_addmod(b[d], orc_mulmod(a[d], s[d], p), p);
        t = t ? p - t : 0;
        if (cm == j) t = orc_addmod(t, orc_mulmod(rot[(size_t)j * N + d], qK % p, p), p);   /* :111-116 */
        b[d] = t;
      }
    }
  }
  free(v); free(ext); free(rot);
}

void orc_encrypt(const orc_ctx *c, u64 *bx, u64 *ax, const u64 *msg, uint32_t Li, const u64 *pk_b, const u64 *pk_a, u64 seed) {
  /* encryptor.cpp:9-37 */
  const size_t N = c->N; u64 st = seed;
  int *v = (int *)calloc(N, sizeof(int));
  u64 *u = (u64 *)malloc((size_t)Li * N * 8);
  sample_zo(v, N, &st); set_small(c, u, v, 0, Li);
  sample_gauss(v, N, &st); set_small(c, bx, v, 0, Li);
  sample_gauss(v, N, &st); set_small(c, ax, v, 0, Li);
  for (uint32_t cm = 0; cm < Li; ++cm) {
    const u64 p = c->P[cm]; const size_t o = (size_t)cm * N;
    orc_ntt_forward(c, u + o, cm);
    for (size_t d = 0; d < N; ++d) bx[o + d] = orc_addmod(bx[o + d], msg[o + d], p); /* e0 + msg in the power basis */
    orc_ntt_forward(c, bx + o, cm); orc_ntt_forward(c, ax + o, cm);
    for (size_t d = 0; d < N; ++d) {
      bx[o + d] = orc_addmod(bx[o + d], orc_mulmod(u[o + d], pk_b[o + d], p), p);
      ax[o + d] = orc_addmod(ax[o + d], orc_mulmod(u[o + d], pk_a[o + d], p), p);
    }
  }
  free(v); free(u);
}

void orc_encrypt_core(const orc_ctx *c, u64 *bx, u64 *ax, const u64 *u, const u64 *e0m, const u64 *e1, uint32_t Li,
                      const u64 *pk_b, const u64 *pk_a) { /* encryptor.cpp:15-35 */
  const size_t N = c->N;
  u64 *uh = (u64 *)malloc(N * 8);
  for (uint32_t cm = 0; cm < Li; ++cm) {
    const u64 p = c->P[cm]; const size_t o = (size_t)cm * N;
    memcpy(uh, u + o, N * 8); memcpy(bx + o, e0m + o, N * 8); memcpy(ax + o, e1 + o, N * 8);
    orc_ntt_forward(c, uh, cm); orc_ntt_forward(c, bx + o, cm); orc_ntt_forward(c, ax + o, cm);
    for (size_t d = 0; d < N; ++d) {
      bx[o + d] = orc_addmod(bx[o + d], orc_mulmod(uh[d], pk_b[o + d], p), p);
      ax[o + d] = orc_addmod(ax[o + d], orc_mulmod(uh[d], pk_a[o + d], p), p);
    }
  }
  free(uh);
}

void orc_decrypt(const orc_ctx *c, u64 *rop, const u64 *bx, const u64 *ax, uint32_t Li, const u64 *sx) { /* decryptor.cpp:11-38 */
  const size_t N = c->N;
  for (uint32_t cm = 0; cm < Li; ++cm) {
    const size_t o = (size_t)cm * N;
    for (size_t d = 0; d < N; ++d) rop[o + d] = reduce128(c, (u128)bx[o + d] + (u128)sx[o + d] * ax[o + d], cm);
    orc_ntt_backward(c, rop + o, cm);
  }
}

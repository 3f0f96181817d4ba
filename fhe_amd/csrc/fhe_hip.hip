// fhe_hip.hip — C-ABI (include/fhe_hip.h) over the gfx950 kernels: context, tables, key store,
// and the evaluator pipelines (multiply+relinearize+rescale, rotation key switch).
// Host-side table construction follows BaseConverter::Impl::Impl() (reference src/base_converter.cpp:122-187).
#include "../../include/fhe_hip.h"
#include "ntt_kernels.h"
#include "pointwise_kernels.h"
#include "bconv_mfma_kernels.h"
#include "encoder_kernels.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <algorithm>
#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

typedef unsigned __int128 u128;

static thread_local std::string g_create_error;
// no exception crosses the C ABI (std::bad_alloc from the host-side containers is the only realistic one)
static int abi_exception() { try { g_create_error = "C++ exception inside the library (out of host memory?)"; } catch (...) {} return FHE_HIP_ERR_HIP; }

// Scratch of the entry points that allocate nothing (include/fhe_hip.h: fhe_hip_mul_relin_rescale_noalloc).  One copy per GPU
// the code object is loaded on, held until the process ends.  In host code this name is only a handle for
// hipGetSymbolAddress (fhe_hip_ctx_create, once per context with its device current); kernels receive the address it returns.
static __device__ __align__(256) unsigned char fhe_hip_scratch[FHE_HIP_NOALLOC_SCRATCH_BYTES];

namespace {

struct RotKeyDev {
  u64 *beta_m = nullptr, *alpha_m = nullptr;  // [(L+K)][N], Montgomery form
};

struct MixedRotKeyDev {
  u64 *beta_m = nullptr, *alpha_m = nullptr;  // [L][L+1][N], Montgomery form
};

struct ProfEntry {
  std::string name;
  uint64_t launches = 0;
  double total_ms = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

}  // namespace

struct fhe_hip_ctx {
  int device = 0;
  uint32_t logN = 0, L = 0, K = 0, M = 0, logR = 0;
  u64 N = 0;
  bool sp = false;  // every prime = 1 (mod 2^32): multiplication-free Montgomery quotient
  hipStream_t stream = nullptr;
  std::string err;
  std::vector<u64> primes;
  // device tables
  PrimeC *d_primes = nullptr;
  u64 *d_tw_fwd = nullptr, *d_tw_inv = nullptr;  // [M][N]
  u32 *d_iota = nullptr;                         // [M] identity modulus map
  u32 *d_rotmap = nullptr;                       // [L+1][M]: row Li: cm < Li ? cm : L + cm - Li
  u64 *d_hat_pi_inv_m = nullptr;                 // [K]          mod p_k
  u64 *d_neg_hat_pi_qj_m = nullptr;              // [L][K]       mod q_l, times 2^128 (bconv_fold)
  u64 *d_Pinv_qj_m = nullptr;                    // [L]          mod q_j
  u64 *d_modup_scale_m = nullptr;                // [L-1][M]   N^-1 * hat_qj_inv[Li-2][j] mod q_j (row Li-2), Montgomery form
  u64 *d_moddown_scale_m = nullptr;              // [M]        N^-1 * hat_pi_inv[k] mod p_k at index L+k, Montgomery form
  u64 *d_hat_qj_inv_m = nullptr;                 // [L-1][L]     mod q_j
  u64 *d_hat_qj_pi_m = nullptr;                  // [L-1][K][L]  mod p_k, times 2^128
  u64 *d_qinv_m = nullptr;                       // [M][M]: (P[a]^-1 mod P[j]) Montgomery, row a
  // base conversion on the matrix cores (bconv_mfma_kernels.h): per-prime bias constant, and the Toeplitz byte images of the
  // conversion tables, built on first use (key: 0 = special -> normal, Li = normal limbs 0..Li-1 -> special)
  BconvTgt *d_bconv_tgt = nullptr;               // [M]  per-prime constants of a conversion target (accumulator start values)
  std::map<u32, uint4 *> bconv_tables;
  std::vector<u64> h_neg_hat, h_hat_qj_pi;       // host copies of d_neg_hat_pi_qj_m / d_hat_qj_pi_m (the byte images are built from them)
  uint32_t bconv_mode = 0;                       // FHE_HIP_TUNE_BCONV: 0 = by shape and size, 1 = vector-ALU kernel only, 2 = MFMA kernel whenever the shape allows
  bool bconv_attr_set[4][2] = {{false, false}, {false, false}, {false, false}, {false, false}};
  // encoder (fhe_hip_encode / _decode), built on first use: psi^brv(i) as complex doubles, the generator-3 slot index map, and the
  // Garner constants of the CRT lift over the normal primes
  cplx *d_enc_roots = nullptr; u32 *d_enc_map = nullptr;
  u64 *d_enc_prod_m = nullptr, *d_enc_ginv_m = nullptr; double *d_enc_qd = nullptr;
  bool dec_attr_set = false;
  std::map<u32, u64 *> enc_half;   // limb count n -> mixed-radix digits of floor(q_0 .. q_{n-1} / 2)
  // keys
  u64 *d_mk_beta = nullptr, *d_mk_alpha = nullptr;  // [L][L][N] Montgomery form
  std::map<u64, RotKeyDev> rot_keys;
  std::map<u64, MixedRotKeyDev> mixed_keys;
  std::map<u64, u32 *> galois_perm;
  // workspaces (grow-only)
  void *ws[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t ws_bytes[6] = {0, 0, 0, 0, 0, 0};
  // entry points that allocate nothing carve their workspaces out of this device's copy of the static scratch array
  // instead (fixed_ws set for the duration of such a call; fixed_used = bytes carved so far)
  // fhe_hip_batch_run: copy streams, hand-over events and three device staging slots {in0, in1, out / in-place}
  hipStream_t br_up[2] = {nullptr, nullptr}, br_down[2] = {nullptr, nullptr};   // two copy lanes per direction
  hipEvent_t br_ev_up[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}}, br_ev_done[3] = {nullptr, nullptr, nullptr};
  void *br_buf[3][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}};
  size_t br_bytes[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  unsigned char *scratch = nullptr;
  bool fixed_ws = false; size_t fixed_used = 0;
  uint32_t relin_group = 0;  // output limbs per relinearisation pass (0 = all Li at once): fhe_hip_set_tuning
  uint32_t relin4 = 1;       // digit key switch with unreduced key-MAC accumulators (relin_rows4_mac_kernel); 0: the Montgomery-MAC kernel
  bool mk_128 = false;       // the MultKey / MixedRotKey copies are stored times 2^128 (relin4's form) instead of times 2^64
  std::map<u64, bool> mixed_128;
  int num_cus = 256;         // persistent kernels size their grids by it
  uint32_t onepass = 1;      // N = 2^12 .. 2^14: plain NTTs as one kernel with the limb in LDS (FHE_HIP_TUNE_ONEPASS)
  uint32_t ntt_chunk = 0;    // limbs per pass pair of fhe_hip_ntt_forward / _backward (0 = as many as one grid takes)
  // fhe_hip_ntt_forward / _backward at N >= 2^15 (two passes over memory): chunks of the call are issued alternately on two
  // library-owned streams, forked from and joined to the caller's stream by events, so that one chunk's column pass runs under
  // another chunk's row pass and a chunk's intermediate is re-read from the Infinity Cache (FHE_HIP_TUNE_NTT_STREAMS)
  uint32_t ntt_streams = 2;
  hipStream_t ntt_aux[2] = {nullptr, nullptr};
  hipEvent_t ntt_fork = nullptr, ntt_join[2] = {nullptr, nullptr};
  void *stage[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t stage_bytes[4] = {0, 0, 0, 0};
  // per-launch timing: consecutive launches share an event (end of one = start of the next) while nothing else was enqueued
  hipEvent_t chain_ev = nullptr; bool chain_valid = false; std::vector<hipEvent_t> prof_events;
  // moduli_idx of the last fhe_hip_ntt_* call, kept on both sides: a repeated call with the same indices uploads nothing
  u32 *d_idx = nullptr; size_t d_idx_cap = 0; std::vector<u32> h_idx;
  // profiling
  bool prof = false;
  std::vector<ProfEntry> prof_entries;
};

namespace {

#define HIPCHK(c, call)                                                                        \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      (c)->err = std::string(#call) + ": " + hipGetErrorString(e_);                            \
      return FHE_HIP_ERR_HIP;                                                                  \
    }                                                                                          \
  } while (0)

#define FAIL(c, code, msg) do { (c)->err = (msg); return (code); } while (0)
#define TRY(x) do { int rc_ = (x); if (rc_ != FHE_HIP_OK) return rc_; } while (0)

// ---- host modular helpers (table construction only) -------------------------------------------
u64 h_mulmod(u64 a, u64 b, u64 p) { return (u64)(((u128)a * b) % p); }
u64 h_powmod(u64 a, u64 e, u64 p) {
  u64 r = 1; a %= p;
  while (e) { if (e & 1) r = h_mulmod(r, a, p); a = h_mulmod(a, a, p); e >>= 1; }
  return r;
}
u64 h_invmod(u64 a, u64 p) { return h_powmod(a % p, p - 2, p); }
u64 h_to_mont(u64 x, u64 p) { return (u64)((((u128)(x % p)) << 64) % p); }
size_t h_brv(size_t x, uint32_t bits) {
  size_t r = 0;
  for (uint32_t i = 0; i < bits; ++i) { r = (r << 1) | (x & 1); x >>= 1; }
  return r;
}
// fHE::product_of (include/fHE/yell.hpp:9-21) over P[first..first+n) skipping `skip`, mod P[m]
u64 h_product_of(const std::vector<u64> &P, uint32_t first, uint32_t n, int skip, uint32_t m) {
  u64 prod = 1;
  for (uint32_t i = 0; i < n; ++i) {
    if ((int)i == skip) continue;
    prod = h_mulmod(prod, P[first + i] % P[m], P[m]);
  }
  return prod;
}

template <typename T>
int upload(fhe_hip_ctx *c, T **dst, const std::vector<T> &v) {
  size_t bytes = (v.size() ? v.size() : 1) * sizeof(T);
  HIPCHK(c, hipMalloc((void **)dst, bytes));
  if (!v.empty()) HIPCHK(c, hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return FHE_HIP_OK;
}

int ensure_ws(fhe_hip_ctx *c, int slot, size_t bytes) {
  if (c->ws_bytes[slot] >= bytes) return FHE_HIP_OK;
  if (c->fixed_ws) {  // no allocation: carve from the static scratch array (slots are fresh for every such call)
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (c->ws[slot] || need > (size_t)FHE_HIP_NOALLOC_SCRATCH_BYTES - c->fixed_used)
      FAIL(c, FHE_HIP_ERR_RANGE, "operands too large for the fixed scratch buffer of the entry points that allocate nothing (FHE_HIP_NOALLOC_SCRATCH_BYTES)");
    c->ws[slot] = c->scratch + c->fixed_used;
    c->ws_bytes[slot] = bytes;
    c->fixed_used += need;
    return FHE_HIP_OK;
  }
  if (c->ws[slot]) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(c->ws[slot])); c->ws[slot] = nullptr; c->ws_bytes[slot] = 0; }
  HIPCHK(c, hipMalloc(&c->ws[slot], bytes));
  c->ws_bytes[slot] = bytes;
  return FHE_HIP_OK;
}
int ensure_stage(fhe_hip_ctx *c, int slot, size_t bytes) {
  if (c->stage_bytes[slot] >= bytes) return FHE_HIP_OK;
  if (c->stage[slot]) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(c->stage[slot])); c->stage[slot] = nullptr; c->stage_bytes[slot] = 0; }
  HIPCHK(c, hipMalloc(&c->stage[slot], bytes));
  c->stage_bytes[slot] = bytes;
  return FHE_HIP_OK;
}

// ---- launch helper with optional per-kernel event timing ---------------------------------------
ProfEntry *prof_entry(fhe_hip_ctx *c, const char *name) {
  for (auto &e : c->prof_entries) if (e.name == name) return &e;
  c->prof_entries.push_back(ProfEntry());
  c->prof_entries.back().name = name;
  return &c->prof_entries.back();
}
void prof_drain(fhe_hip_ctx *c) {
  for (auto &e : c->prof_entries) {
    for (auto &pr : e.pending) {
      hipEventSynchronize(pr.second);
      float ms = 0;
      hipEventElapsedTime(&ms, pr.first, pr.second);
      e.total_ms += ms; e.launches += 1;
    }
    e.pending.clear();
  }
  for (hipEvent_t ev : c->prof_events) hipEventDestroy(ev);
  c->prof_events.clear();
  c->chain_valid = false;
}
// anything the library enqueues besides a timed launch ends the chain of shared events (so does every entry: check_ctx)
inline hipError_t prof_chain_break(fhe_hip_ctx *c) { c->chain_valid = false; return hipSuccess; }
inline hipEvent_t prof_event(fhe_hip_ctx *c) {
  hipEvent_t ev = nullptr;
  hipEventCreate(&ev);
  hipEventRecord(ev, c->stream);
  c->prof_events.push_back(ev);
  return ev;
}

#define LAUNCH_SMEM(c, name, kern, grid, block, smem, ...)                                      \
  do {                                                                                           \
    hipEvent_t ev0_ = nullptr;                                                                   \
    if ((c)->prof && !(c)->fixed_ws) ev0_ = (c)->chain_valid ? (c)->chain_ev : prof_event(c);                      \
    hipLaunchKernelGGL(kern, grid, block, smem, (c)->stream, __VA_ARGS__);                       \
    if ((c)->prof && !(c)->fixed_ws) {                                                           \
      (c)->chain_ev = prof_event(c); (c)->chain_valid = true;                                    \
      prof_entry((c), name)->pending.push_back({ev0_, (c)->chain_ev});                           \
    }                                                                                            \
    HIPCHK(c, hipGetLastError());                                                                \
  } while (0)
#define LAUNCH(c, name, kern, grid, block, ...) LAUNCH_SMEM(c, name, kern, grid, block, 0, __VA_ARGS__)

#define LAUNCH_SP(c, name, kern, grid, block, ...)                                               \
  do {                                                                                           \
    if ((c)->sp) LAUNCH(c, name, (kern<true>), grid, block, __VA_ARGS__);                        \
    else LAUNCH(c, name, (kern<false>), grid, block, __VA_ARGS__);                               \
  } while (0)

// ---- NTT launchers ------------------------------------------------------------------------------
struct NttPlan {
  const u64 *src; u64 *dst;
  const u32 *midx; u32 mdiv, mper, smod;
  u32 nl;        // limbs per batch entry
  u32 batch;
  u64 src_bstride, dst_bstride;
  u32 skip_diag = 0;
  u32 l0 = 0;              // first limb id of this launch (limb id = grid y + l0; dst is addressed by the limb id)
  bool cols_only = false;  // run the column pass only (the row pass is fused into another kernel)
  // forward only: epilogue fused into the row pass (NttArgs::epi); dst is then only the column-pass scratch
  u32 epi = 0;
  const u64 *epi_src = nullptr, *epi_c = nullptr, *epi_add = nullptr;
  u64 *epi_out = nullptr;
  u64 epi_src_bstride = 0, epi_add_bstride = 0, epi_out_bstride = 0;
  const u64 *inv_scale = nullptr;     // inverse only: per-modulus factor replacing N^-1 in the column pass (NttArgs::inv_scale)
  const u32 *perm = nullptr;          // inverse only: gather the source through a galois index table
  const u32 *epi_add_perm = nullptr;  // gather the epilogue addend through a galois index table
  const u64 *pre_a = nullptr, *pre_k = nullptr;  // inverse, 8-element row pass only: element = src + pre_k * pre_a (NttArgs::pre_a)
  const u64 *epi_key0 = nullptr, *epi_key1 = nullptr; u64 epi_half = 0;  // epi == 4 (NttArgs::epi_key0)
};

template <bool SP, bool INV>
int launch_cols(fhe_hip_ctx *c, const NttArgs &a, u32 nl, u32 batch) {
  const char *name = INV ? "ntt_inv_cols" : "ntt_fwd_cols";
#define COLS_CASE(LR)                                                                            \
  case LR: {                                                                                     \
    dim3 grid(256 / ColCfg<LR>::CT, nl, batch), block(ColCfg<LR>::THREADS);                      \
    LAUNCH(c, name, (ntt_cols_kernel<SP, INV, LR>), grid, block, a);                             \
    break;                                                                                       \
  }
  switch (c->logR) {
    COLS_CASE(2) COLS_CASE(3) COLS_CASE(4) COLS_CASE(5) COLS_CASE(6) COLS_CASE(7) COLS_CASE(8)
    default: FAIL(c, FHE_HIP_ERR_RANGE, "unsupported logN");
  }
#undef COLS_CASE
  return FHE_HIP_OK;
}

// g^-1 mod 2N for odd g (Newton iteration mod 2^64, then masked)
static u64 galois_inverse(u64 g, u64 N) {
  u64 inv = g;
  for (int i = 0; i < 6; ++i) inv *= 2 - g * inv;
  return inv & (2 * N - 1);
}

// row pass: 8 rows of a limb per workgroup (all R rows when R < 8), 32 threads per row
template <bool SP, bool INV>
int launch_rows(fhe_hip_ctx *c, const NttArgs &a, u32 nl, u32 batch) {
  const u32 R = 1u << c->logR, rows = R < 8 ? R : 8;
  const dim3 grid(R / rows, nl, batch), block(32 * rows);
  if constexpr (!INV) {
    if (a.epi == 1) { LAUNCH(c, "ntt_fwd_rows_rescale", (ntt_rows8_fwd_kernel<SP, 1>), grid, block, a); return FHE_HIP_OK; }
    if (a.epi == 2) { LAUNCH(c, "ntt_fwd_rows_moddown", (ntt_rows8_fwd_kernel<SP, 2>), grid, block, a); return FHE_HIP_OK; }
    if (a.epi == 3) { LAUNCH(c, "ntt_fwd_rows_encrypt", (ntt_rows8_fwd_kernel<SP, 3>), grid, block, a); return FHE_HIP_OK; }
    if (a.epi == 4) { LAUNCH(c, "ntt_fwd_rows_rotdown", (ntt_rows8_fwd_kernel<SP, 4>), dim3(R / rows, batch, nl), block, a); return FHE_HIP_OK; }  // limb slowest (kernel comment)
    LAUNCH(c, "ntt_fwd_rows", (ntt_rows8_fwd_kernel<SP, 0>), grid, block, a);
  } else {
    if (a.pre_a && a.perm) FAIL(c, FHE_HIP_ERR_ARG, "internal: inverse row pass with both a gather and a pre-multiply");
    if (a.pre_a) LAUNCH(c, "ntt_inv_rows_decrypt", (ntt_rows8_inv_kernel<SP, true, false>), grid, block, a);
    else if (a.perm) LAUNCH(c, "ntt_inv_rows", (ntt_rows8_inv_kernel<SP, false, true>), grid, block, a);
    else LAUNCH(c, "ntt_inv_rows", (ntt_rows8_inv_kernel<SP, false, false>), grid, block, a);
  }
  return FHE_HIP_OK;
}

// one-pass NTT (a whole limb in one workgroup's LDS): N = 2^12 .. 2^14, plain transforms only
template <bool SP, bool INV>
int launch_onepass(fhe_hip_ctx *c, const NttArgs &a, u32 nl, u32 batch) {
  const char *name = INV ? "ntt_inv_onepass" : "ntt_fwd_onepass";
#define ONE_CASE(LR)                                                                             \
  case LR: {                                                                                     \
    dim3 block(OneCfg<LR>::T);                                                                   \
    const u32 total = nl * batch, slots = (u32)c->num_cus * (64 >> LR);  /* workgroups the LDS lets a CU hold: 4, 2, 1 */ \
    dim3 pgrid(total < slots ? total : slots);                                                   \
    if constexpr (INV) LAUNCH(c, name, (ntt_onepass_inv_kernel<SP, LR>), pgrid, block, a, nl, total); \
    else LAUNCH(c, name, (ntt_onepass_fwd_kernel<SP, LR>), pgrid, block, a, nl, total);          \
    break;                                                                                       \
  }
  switch (c->logR) {
    ONE_CASE(4) ONE_CASE(5) ONE_CASE(6)
    default: FAIL(c, FHE_HIP_ERR_RANGE, "internal: one-pass NTT outside N = 2^12 .. 2^14");
  }
#undef ONE_CASE
  return FHE_HIP_OK;
}

template <bool SP>
int ntt_run_sp(fhe_hip_ctx *c, const NttPlan &p, bool inverse) {
  if (p.nl == 0 || p.batch == 0) return FHE_HIP_OK;
  NttArgs a;
  a.tw = inverse ? c->d_tw_inv : c->d_tw_fwd;
  a.primes = c->d_primes; a.midx = p.midx; a.mdiv = p.mdiv; a.mper = p.mper;
  auto inv32 = [](u32 d) { return (u32)(((1ull << 32) + d - 1) / (d ? d : 1)); };  // ceil(2^32 / d); unused for d == 1
  a.mdiv_inv = inv32(p.mdiv); a.mper_inv = inv32(p.mper);
  a.logN = c->logN; a.logR = c->logR; a.l0 = p.l0; a.skip_diag = p.skip_diag;
  a.dst = p.dst; a.dst_bstride = p.dst_bstride;
  a.epi = inverse ? 0 : p.epi; a.epi_src = p.epi_src; a.epi_c = p.epi_c; a.epi_add = p.epi_add; a.epi_out = p.epi_out;
  a.epi_src_bstride = p.epi_src_bstride; a.epi_add_bstride = p.epi_add_bstride; a.epi_out_bstride = p.epi_out_bstride;
  a.epi_key0 = p.epi_key0; a.epi_key1 = p.epi_key1; a.epi_half = p.epi_half;
  a.perm = nullptr; a.pre_a = nullptr; a.pre_k = nullptr; a.epi_add_perm = p.epi_add_perm; a.inv_scale = inverse ? p.inv_scale : nullptr;
  // first pass reads the (possibly broadcast) source, second pass runs in place on dst
  NttArgs first = a, second = a;
  first.src = p.src; first.src_bstride = p.src_bstride; first.smod = p.smod; first.smod_inv = inv32(p.smod);
  if (inverse) { first.perm = p.perm; first.pre_a = p.pre_a; first.pre_k = p.pre_k; }
  second.src = p.dst; second.src_bstride = p.dst_bstride; second.smod = p.nl; second.smod_inv = inv32(p.nl);
  if (c->onepass && c->logR >= 4 && c->logR <= 6 && !p.cols_only && !p.skip_diag && (inverse ? (!p.perm && !p.pre_a) : p.epi == 0)) {
    if (inverse) TRY((launch_onepass<SP, true>(c, first, p.nl, p.batch)));
    else TRY((launch_onepass<SP, false>(c, first, p.nl, p.batch)));
    return FHE_HIP_OK;
  }
  if (!inverse) {
    TRY((launch_cols<SP, false>(c, first, p.nl, p.batch)));
    if (!p.cols_only) TRY((launch_rows<SP, false>(c, second, p.nl, p.batch)));
  } else {
    if (!p.cols_only) TRY((launch_rows<SP, true>(c, first, p.nl, p.batch)));  // cols_only: the row pass was fused elsewhere
    TRY((launch_cols<SP, true>(c, second, p.nl, p.batch)));
  }
  return FHE_HIP_OK;
}

int ntt_run(fhe_hip_ctx *c, const NttPlan &p, bool inverse) {
  // grid.y / grid.z limits: split large limb counts over several launches
  if (p.skip_diag && p.skip_diag != p.mdiv) FAIL(c, FHE_HIP_ERR_ARG, "internal: skip_diag must equal mdiv");
  // a limb offset is only meaningful for a forward column-pass-only plan: the in-place second pass addresses its source by
  // (limb id) mod nl, and an inverse plan's first pass would do the same with the row pass
  if (p.l0 != 0 && (!p.cols_only || inverse)) FAIL(c, FHE_HIP_ERR_ARG, "internal: NttPlan::l0 needs a forward cols_only plan");
  if (p.nl + p.l0 > 65535 || p.batch > 65535 || p.mdiv > 65535 || p.mper > 65535 || p.smod > 65535)
    FAIL(c, FHE_HIP_ERR_RANGE, "too many limbs in one NTT launch");  // also the range limb_div is exact on
  return c->sp ? ntt_run_sp<true>(c, p, inverse) : ntt_run_sp<false>(c, p, inverse);
}

dim3 pw_grid(const fhe_hip_ctx *c, u32 y, u32 z) { return dim3((unsigned)(c->N / (2 * PW_THREADS)), y, z); }

// ---- staging of host buffers ---------------------------------------------------------------------
struct Staged {
  fhe_hip_ctx *c; int kind;
  struct Out { void *host; void *dev; size_t bytes; };
  std::vector<Out> outs;
  int next = 0;
  Staged(fhe_hip_ctx *ctx, int mem_kind) : c(ctx), kind(mem_kind) {}
  // returns device pointer for an input
  int in(const void *p, size_t bytes, const u64 **dev) {
    if (kind == FHE_HIP_MEM_DEVICE) { *dev = (const u64 *)p; return FHE_HIP_OK; }
    int slot = next++;
    TRY(ensure_stage(c, slot, bytes));
    HIPCHK(c, (prof_chain_break(c), hipMemcpyAsync(c->stage[slot], p, bytes, hipMemcpyHostToDevice, c->stream)));
    *dev = (const u64 *)c->stage[slot];
    return FHE_HIP_OK;
  }
  int out(void *p, size_t bytes, u64 **dev, bool also_in) {
    if (kind == FHE_HIP_MEM_DEVICE) { *dev = (u64 *)p; return FHE_HIP_OK; }
    int slot = next++;
    TRY(ensure_stage(c, slot, bytes));
    if (also_in) HIPCHK(c, (prof_chain_break(c), hipMemcpyAsync(c->stage[slot], p, bytes, hipMemcpyHostToDevice, c->stream)));
    *dev = (u64 *)c->stage[slot];
    outs.push_back({p, c->stage[slot], bytes});
    return FHE_HIP_OK;
  }
  int finish() {
    if (kind == FHE_HIP_MEM_DEVICE) return FHE_HIP_OK;
    for (auto &o : outs) HIPCHK(c, (prof_chain_break(c), hipMemcpyAsync(o.host, o.dev, o.bytes, hipMemcpyDeviceToHost, c->stream)));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return FHE_HIP_OK;
  }
};

int check_ctx(fhe_hip_ctx *c) {
  if (!c) return FHE_HIP_ERR_ARG;
  c->chain_valid = false;  // the caller may have enqueued work of its own on the stream since the last call
  hipError_t e = hipSetDevice(c->device);
  if (e != hipSuccess) { c->err = std::string("hipSetDevice: ") + hipGetErrorString(e); return FHE_HIP_ERR_HIP; }
  return FHE_HIP_OK;
}

// ---- device pipelines (all pointers device) ------------------------------------------------------

// Base conversion with ready-made digits on the matrix cores (bconv_mfma_kernels.h) when the shape allows and the launch is
// large enough to fill persistent workgroups; returns false when the vector-ALU kernel has to run instead.
template <int NS, bool PRE>
int launch_bconv_mfma(fhe_hip_ctx *c, const char *name, const BconvMfmaArgs &m, unsigned grid, size_t smem) {
  const void *fn = c->sp ? (const void *)bconv_mfma_kernel<true, NS, PRE> : (const void *)bconv_mfma_kernel<false, NS, PRE>;
  bool &attr = c->bconv_attr_set[NS - 1][PRE ? 1 : 0];
  if (!attr) {   // the table takes more LDS than the default dynamic limit
    HIPCHK(c, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr = true;
  }
  if (c->sp) LAUNCH_SMEM(c, name, (bconv_mfma_kernel<true, NS, PRE>), dim3(grid), dim3(BCONV_MFMA_THREADS), smem, m);
  else LAUNCH_SMEM(c, name, (bconv_mfma_kernel<false, NS, PRE>), dim3(grid), dim3(BCONV_MFMA_THREADS), smem, m);
  return FHE_HIP_OK;
}
int try_bconv_mfma(fhe_hip_ctx *c, const char *name, const BconvArgs &a, const u64 *hat_h, u32 table_key, u32 table_ntgt, u32 batch, bool pre, bool *done) {
  *done = false;
  if (c->bconv_mode == 1 || c->fixed_ws || a.nsrc > 32 || a.ntgt == 0 || c->N < BCONV_MFMA_BLOCK) return FHE_HIP_OK;
  const u32 NS = (a.nsrc + 7) / 8, ngroups = (a.ntgt + 3) / 4;
  const size_t smem = (size_t)ngroups * 2 * NS * 64 * 16 + (size_t)a.ntgt * sizeof(BconvTgt);   // byte image + the targets' constants
  if (smem > 158 * 1024) return FHE_HIP_OK;
  const u32 bpe = (u32)(c->N / BCONV_MFMA_BLOCK), nblocks = bpe * batch;
  if (c->bconv_mode == 0 && nblocks < 2 * (u32)c->num_cus) return FHE_HIP_OK;   // too small to pay for the per-workgroup table load
  uint4 *table = nullptr;
  auto it = c->bconv_tables.find(table_key);
  if (it != c->bconv_tables.end()) table = it->second;
  else {
    // all table_ntgt targets of the conversion (a launch for fewer targets reads a prefix: the image is grouped by target).
    // Fragment ((grp NS + st) 2 + tile), lane (row m, k-group kg), byte b: source limb 8 st + 2 kg + (b >> 3), digit byte i = b & 7,
    // target 4 grp + (m >> 2), constant byte j = 4 tile + (m & 3) of c = hat * 256^i * 2^64 mod p in balanced bytes.
    const u32 tgroups = (table_ntgt + 3) / 4, nfrag = tgroups * 2 * NS * 64;
    std::vector<unsigned char> img((size_t)nfrag * 16, 0);
    for (u32 t = 0; t < table_ntgt; ++t) {
      const u64 p = c->primes[a.tgt_first + t];
      const u64 rinv = h_invmod(h_to_mont(1, p), p);   // 2^-64 mod p
      for (u32 k = 0; k < a.nsrc; ++k) {
        u64 cki = h_mulmod(hat_h[(size_t)t * a.hat_stride + k], rinv, p);   // hat * 2^64 mod p (the stored form carries 2^128)
        for (u32 i = 0; i < 8; ++i) {
          const u64 bal = (cki + 0x8080808080808080ull) ^ 0x8080808080808080ull;
          const u32 st = k / 8, kg = (k % 8) / 2, b = (k % 2) * 8 + i;
          for (u32 j = 0; j < 8; ++j) {
            const u32 m = (t % 4) * 4 + (j % 4), tile = j / 4, lane = kg * 16 + m;
            img[((((size_t)(t / 4) * NS + st) * 2 + tile) * 64 + lane) * 16 + b] = (unsigned char)(bal >> (8 * j));
          }
          cki = h_mulmod(cki, 256, p);
        }
      }
    }
    HIPCHK(c, hipMalloc((void **)&table, img.size()));
    c->bconv_tables[table_key] = table;
    HIPCHK(c, prof_chain_break(c));
    HIPCHK(c, hipMemcpyAsync(table, img.data(), img.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));   // (img goes out of scope)
  }
  BconvMfmaArgs m;
  m.out = a.out; m.src = a.src; m.table = table; m.tgt = c->d_bconv_tgt;
  m.primes = a.primes; m.inv_m = a.inv_m; m.src_first = a.src_first;
  m.nsrc = a.nsrc; m.src_limbs = a.src_limbs; m.src_off = a.src_off;
  m.tgt_first = a.tgt_first; m.out_limbs = a.out_limbs; m.out_off = a.out_off; m.ntgt = a.ntgt; m.ngroups = ngroups;
  m.nblocks = nblocks; m.blocks_per_entry = bpe; m.N = c->N;
  const unsigned grid = nblocks < (u32)c->num_cus ? nblocks : (unsigned)c->num_cus;
  switch (NS * 2 + (pre ? 1 : 0)) {
    case 2: TRY((launch_bconv_mfma<1, false>(c, name, m, grid, smem))); break;
    case 3: TRY((launch_bconv_mfma<1, true>(c, name, m, grid, smem))); break;
    case 4: TRY((launch_bconv_mfma<2, false>(c, name, m, grid, smem))); break;
    case 5: TRY((launch_bconv_mfma<2, true>(c, name, m, grid, smem))); break;
    case 6: TRY((launch_bconv_mfma<3, false>(c, name, m, grid, smem))); break;
    case 7: TRY((launch_bconv_mfma<3, true>(c, name, m, grid, smem))); break;
    case 8: TRY((launch_bconv_mfma<4, false>(c, name, m, grid, smem))); break;
    default: TRY((launch_bconv_mfma<4, true>(c, name, m, grid, smem))); break;
  }
  *done = true;
  return FHE_HIP_OK;
}

// BaseConverter::approximated_mod_up body on device: special limbs of rop from power-basis op
int dev_conv_to_special(fhe_hip_ctx *c, u64 *rop, u32 rop_limbs, u32 rop_off, const u64 *op, u32 op_limbs, u32 Li, u32 batch, bool pre_scaled = false) {
  if (c->K == 0 || batch == 0) return FHE_HIP_OK;  // nothing to convert to
  BconvArgs a;
  a.out = rop; a.src = op;
  a.inv_m = c->d_hat_qj_inv_m + (size_t)(Li - 2) * c->L;
  a.hat_m = c->d_hat_qj_pi_m + (size_t)(Li - 2) * c->K * c->L;
  a.primes = c->d_primes;
  a.nsrc = Li; a.src_first = 0; a.src_limbs = op_limbs; a.src_off = 0;
  a.tgt_first = c->L; a.out_limbs = rop_limbs; a.out_off = rop_off; a.hat_stride = c->L; a.ntgt = c->K;
  a.N = c->N;
  const dim3 grid((unsigned)(c->N / PW_THREADS), (c->K + BCONV_TC - 1) / BCONV_TC, batch);
  // pre_scaled: op already holds the digits [x_k * inv_k]_{q_k} (the producing inverse NTT was scaled by N^-1 * inv_k)
  { bool done; TRY(try_bconv_mfma(c, "bconv_modup", a, c->h_hat_qj_pi.data() + (size_t)(Li - 2) * c->K * c->L, Li, c->K, batch, pre_scaled, &done)); if (done) return FHE_HIP_OK; }
  if (pre_scaled) { if (c->sp) LAUNCH(c, "bconv_modup", (bconv_kernel<true, true>), grid, dim3(PW_THREADS), a); else LAUNCH(c, "bconv_modup", (bconv_kernel<false, true>), grid, dim3(PW_THREADS), a); }
  else { if (c->sp) LAUNCH(c, "bconv_modup", (bconv_kernel<true, false>), grid, dim3(PW_THREADS), a); else LAUNCH(c, "bconv_modup", (bconv_kernel<false, false>), grid, dim3(PW_THREADS), a); }
  return FHE_HIP_OK;
}
// approx_convert_to_special_basis (src/base_converter.cpp:276-295), ONE special target k: rop = [b][N]
int dev_conv_to_special_one(fhe_hip_ctx *c, u64 *rop, const u64 *op, u32 Li, u32 k, u32 batch) {
  BconvArgs a;
  a.out = rop; a.src = op;
  a.inv_m = c->d_hat_qj_inv_m + (size_t)(Li - 2) * c->L;
  a.hat_m = c->d_hat_qj_pi_m + ((size_t)(Li - 2) * c->K + k) * c->L;   // row k of the Li-2 table
  a.primes = c->d_primes;
  a.nsrc = Li; a.src_first = 0; a.src_limbs = Li; a.src_off = 0;
  a.tgt_first = c->L + k; a.out_limbs = 1; a.out_off = 0; a.hat_stride = c->L; a.ntgt = 1;
  a.N = c->N;
  const dim3 grid((unsigned)(c->N / PW_THREADS), 1, batch);
  if (c->sp) LAUNCH(c, "bconv_modup", (bconv_kernel<true, false>), grid, dim3(PW_THREADS), a);
  else LAUNCH(c, "bconv_modup", (bconv_kernel<false, false>), grid, dim3(PW_THREADS), a);
  return FHE_HIP_OK;
}
// neg_approx_convert_to_normal_basis for j < L0: special limbs (power basis) at limb offset sp_off of op
int dev_conv_to_normal(fhe_hip_ctx *c, u64 *rop, u32 L0, const u64 *op, u32 op_limbs, u32 sp_off, u32 batch, bool pre_scaled = false) {
  BconvArgs a;
  a.out = rop; a.src = op;
  a.inv_m = c->d_hat_pi_inv_m;
  a.hat_m = c->d_neg_hat_pi_qj_m;
  a.primes = c->d_primes;
  a.nsrc = c->K; a.src_first = c->L; a.src_limbs = op_limbs; a.src_off = sp_off;
  a.tgt_first = 0; a.out_limbs = L0; a.out_off = 0; a.hat_stride = c->K; a.ntgt = L0;
  a.N = c->N;
  const dim3 grid((unsigned)(c->N / PW_THREADS), (L0 + BCONV_TC - 1) / BCONV_TC, batch);
  // (the table of the special -> normal conversion does not depend on L0: a launch for fewer targets uses its first groups)
  { bool done; TRY(try_bconv_mfma(c, "bconv_moddown", a, c->h_neg_hat.data(), 0, c->L, batch, pre_scaled, &done)); if (done) return FHE_HIP_OK; }
  if (pre_scaled) { if (c->sp) LAUNCH(c, "bconv_moddown", (bconv_kernel<true, true>), grid, dim3(PW_THREADS), a); else LAUNCH(c, "bconv_moddown", (bconv_kernel<false, true>), grid, dim3(PW_THREADS), a); }
  else { if (c->sp) LAUNCH(c, "bconv_moddown", (bconv_kernel<true, false>), grid, dim3(PW_THREADS), a); else LAUNCH(c, "bconv_moddown", (bconv_kernel<false, false>), grid, dim3(PW_THREADS), a); }
  return FHE_HIP_OK;
}

// approximated_mod_down: op [e][L0+K][N] -> rop [e][L0][N]; optional addend [e][L0][N]
int dev_mod_down(fhe_hip_ctx *c, u64 *rop, const u64 *op, u32 L0, u32 batch, const u64 *addend, int ws_slot) {
  const size_t bytes = (size_t)batch * L0 * c->N * 8;
  TRY(ensure_ws(c, ws_slot, bytes));
  u64 *conv = (u64 *)c->ws[ws_slot];
  TRY(dev_conv_to_normal(c, conv, L0, op, L0 + c->K, L0, batch));
  // forward NTT of the conversion with the tail (op_j + conv_j) * P^-1 [+ addend] fused into its row pass (:227-240)
  NttPlan p{conv, conv, c->d_iota, 1, L0, L0, L0, batch, (u64)L0 * c->N, (u64)L0 * c->N};
  p.epi = 2; p.epi_src = op; p.epi_src_bstride = (u64)(L0 + c->K) * c->N; p.epi_c = c->d_Pinv_qj_m;
  p.epi_add = addend; p.epi_add_bstride = (u64)L0 * c->N; p.epi_out = rop; p.epi_out_bstride = (u64)L0 * c->N;
  TRY(ntt_run(c, p, false));
  return FHE_HIP_OK;
}

// BaseConverter::rescale: op [e][n][N] (last limb power basis) -> rop [e][n-1][N]
int dev_rescale(fhe_hip_ctx *c, u64 *rop, const u64 *op, u32 n, u32 entries, int ws_slot) {
  const u64 N = c->N;
  TRY(ensure_ws(c, ws_slot, (size_t)entries * (n - 1) * N * 8));
  u64 *tail = (u64 *)c->ws[ws_slot];
  // tail[e][j] = NTT_j(op[e][n-1]) for j < n-1 (src limb broadcast: smod = 1 from the last limb)
  // the row pass ends with rop_j = (op_j + q_j - tail_j) * q_last^-1 (base_converter.cpp:264-271): tail never leaves the chip
  NttPlan p{op + (size_t)(n - 1) * N, tail, c->d_iota, 1, n - 1, 1, n - 1, entries, (u64)n * N, (u64)(n - 1) * N};
  p.epi = 1; p.epi_src = op; p.epi_src_bstride = (u64)n * N; p.epi_c = c->d_qinv_m + (size_t)(n - 1) * c->M;
  p.epi_out = rop; p.epi_out_bstride = (u64)(n - 1) * N;
  TRY(ntt_run(c, p, false));
  return FHE_HIP_OK;
}

// Digit key switch shared by Mul::reLinearize (src/evaluator.cpp:663-691) and apply_rotation_key(MixedRotKey) (:424-473):
// column pass of the nout*nd digit NTTs (diagonal skipped), then the fused row pass + key MAC.  The column-pass output t1 is
// produced and consumed in groups of `group` output limbs ([b][group][nd][N] at a time), which bounds its size.
// ra carries everything but t1 / i0 / gi / nb / rb_count; pw = [b][nd][N] power-basis digits.
int run_digit_keyswitch(fhe_hip_ctx *c, RelinArgs ra, const u64 *pw, u64 *t1, const u32 *omap, u32 nout, u32 nd, u32 nb, u32 group) {
  const u64 N = c->N;
  const u32 R = 1u << c->logR, rows = R < 8 ? R : 8;
  ra.nb = nb; ra.rb_count = R / rows; ra.omap = omap; ra.nd = nd; ra.nout = nout; ra.logN = c->logN; ra.logR = c->logR;
  ra.tw = c->d_tw_fwd; ra.primes = c->d_primes; ra.t1 = t1;
  for (u32 i0 = 0; i0 < nout; i0 += group) {
    const u32 gi = nout - i0 < group ? nout - i0 : group;
    // limb id = i*nd + j uses modulus omap[i] and source digit j; ids i0*nd .. (i0+gi)*nd - 1, written group-relative
    NttPlan pf{pw, t1 - (size_t)i0 * nd * N, omap, nd, nout, nd, gi * nd, nb, (u64)nd * N, (u64)gi * nd * N};
    pf.l0 = i0 * nd; pf.skip_diag = nd; pf.cols_only = true;
    TRY(ntt_run(c, pf, false));
    ra.i0 = i0; ra.gi = gi;
    if (c->relin4) {
      ra.rb_count = R / 4;   // four rows per workgroup, one wave each
      LAUNCH_SP(c, "relin_rows_mac", relin_rows4_mac_kernel, dim3(nb * ra.rb_count * gi), dim3(256), ra);
    } else {
      LAUNCH_SP(c, "relin_rows_mac", relin_rows8p_mac_kernel, dim3(nb * ra.rb_count * gi), dim3(32 * rows), ra);
    }
  }
  return FHE_HIP_OK;
}

// Mul::reLinearize: ct3 [b][3][Li][N] -> out [b][2][Li][N]
int dev_relin(fhe_hip_ctx *c, u64 *out, const u64 *ct3, u32 Li, u32 batch) {
  const u64 N = c->N;
  if (!c->d_mk_beta) FAIL(c, FHE_HIP_ERR_NOKEY, "mult key not loaded");
  if (c->mk_128 != (c->relin4 != 0)) FAIL(c, FHE_HIP_ERR_NOKEY, "mult key was loaded for the other relinearisation kernel (FHE_HIP_TUNE_RELIN_KERNEL): load it again");
  u32 group = c->relin_group ? (c->relin_group < Li ? c->relin_group : Li) : Li;
  u32 chunk = batch;
  if (c->fixed_ws) {
    // everything has to fit what is left of the fixed scratch buffer: whole batch at once, the largest group that fits
    // (limbs are multiples of 256 bytes for every supported N, so ensure_ws rounds nothing up: the same count as the entry point's
    // precheck batch * (9 Li - 2) limbs, which therefore guarantees g >= 1 here)
    const size_t left = (size_t)FHE_HIP_NOALLOC_SCRATCH_BYTES - c->fixed_used, limb = (size_t)N * 8;
    const size_t pw_limbs = (size_t)batch * Li;
    if (left / limb <= pw_limbs) FAIL(c, FHE_HIP_ERR_RANGE, "operands too large for the fixed scratch buffer (FHE_HIP_NOALLOC_SCRATCH_BYTES)");
    const size_t g = (left / limb - pw_limbs) / ((size_t)batch * Li);
    if (g < 1) FAIL(c, FHE_HIP_ERR_RANGE, "operands too large for the fixed scratch buffer (FHE_HIP_NOALLOC_SCRATCH_BYTES)");
    if (g < group) group = (u32)g;
  } else {
    // bound the intermediate: process the batch in chunks of at most 8 GiB
    const size_t per_ct = (size_t)group * Li * N * 8;
    chunk = (u32)((8ull << 30) / per_ct);
    if (chunk < 1) chunk = 1;
    if (chunk > batch) chunk = batch;
  }
  TRY(ensure_ws(c, 0, (size_t)chunk * Li * N * 8));
  TRY(ensure_ws(c, 1, (size_t)chunk * group * Li * N * 8));
  u64 *pw = (u64 *)c->ws[0], *t = (u64 *)c->ws[1];
  for (u32 b0 = 0; b0 < batch; b0 += chunk) {
    const u32 nb = (batch - b0 < chunk) ? batch - b0 : chunk;
    const u64 *in = ct3 + (size_t)b0 * 3 * Li * N;
    // ct2_power = INTT(ct[2])   (evaluator.cpp:659-660)
    NttPlan pi{in + (size_t)2 * Li * N, pw, c->d_iota, 1, Li, Li, Li, nb, (u64)3 * Li * N, (u64)Li * N};
    TRY(ntt_run(c, pi, true));
    // t[i][j] = NTT_i(ct2_power[j]) (evaluator.cpp:666-668), j == i skipped; the row pass, the key MAC (:672-679) and the
    // final "+ ct[l]" (:682-691) are fused in relin_rows8p_mac_kernel
    RelinArgs ra;
    ra.out = out + (size_t)b0 * 2 * Li * N;
    ra.diag = in + (size_t)2 * Li * N; ra.diag_bstride = (u64)3 * Li * N;            // ct[2] limb i (j == i)
    ra.add0 = in; ra.add1 = in + (size_t)Li * N; ra.add_bstride = (u64)3 * Li * N;   // + ct[0], + ct[1]
    ra.beta_m = c->d_mk_beta; ra.alpha_m = c->d_mk_alpha; ra.klimbs = c->L;
    TRY(run_digit_keyswitch(c, ra, pw, t, c->d_iota, Li, Li, nb, group));
  }
  return FHE_HIP_OK;
}

// Mul::rescale_ntt: ct [b][2][Li][N] (modified: last limbs go to the power basis) -> out [b][2][Li-1][N]
int dev_rescale_ntt(fhe_hip_ctx *c, u64 *out, u64 *ct, u32 Li, u32 batch) {
  const u64 N = c->N;
  // INTT of the last limb of each half (evaluator.cpp:707): entries = 2*batch polys of Li limbs
  NttPlan pi{ct + (size_t)(Li - 1) * N, ct + (size_t)(Li - 1) * N, c->d_iota + (Li - 1), 1, 1, 1, 1, 2 * batch, (u64)Li * N, (u64)Li * N};
  TRY(ntt_run(c, pi, true));
  return dev_rescale(c, out, ct, Li, 2 * batch, 2);
}

int dev_mul_relin_rescale(fhe_hip_ctx *c, u64 *out, const u64 *c0, const u64 *c1, u32 Li, u32 batch) {
  const u64 N = c->N;
  TRY(ensure_ws(c, 3, (size_t)batch * 3 * Li * N * 8));
  TRY(ensure_ws(c, 4, (size_t)batch * 2 * Li * N * 8));
  TRY(ensure_ws(c, 2, (size_t)batch * 2 * (Li - 1) * N * 8));  // rescale tails (dev_rescale), reserved before dev_relin sizes its own
  u64 *ct3 = (u64 *)c->ws[3], *rl = (u64 *)c->ws[4];
  LAUNCH_SP(c, "cross_mult", cross_mult_kernel, pw_grid(c, Li, batch), dim3(PW_THREADS), ct3, c0, c1, (const PrimeC *)c->d_primes, Li, N);
  TRY(dev_relin(c, rl, ct3, Li, batch));
  return dev_rescale_ntt(c, out, rl, Li, batch);
}

int get_galois_perm(fhe_hip_ctx *c, u64 galois, u32 **perm) {
  auto it = c->galois_perm.find(galois);
  if (it != c->galois_perm.end()) { *perm = it->second; return FHE_HIP_OK; }
  u32 *d = nullptr;
  HIPCHK(c, hipMalloc((void **)&d, c->N * sizeof(u32)));
  LAUNCH(c, "galois_perm", galois_perm_kernel, dim3((unsigned)((c->N + 255) / 256)), dim3(256), d, c->logN, galois);
  c->galois_perm[galois] = d;
  *perm = d;
  return FHE_HIP_OK;
}

// rotate_slots_inplace + apply_rotation_key (evaluator.cpp:299-311, 343-399), out of place: in, out = [b][2][Li][N], out != in.
// Nine launches.  galois() of both halves is applied inside the loads of its readers (inverse row pass of ax, the mod-down
// tail), so no permuted copy is ever written; the key product of the normal limbs is formed in that tail as well; the two
// halves of the key-switched polynomial travel as 2*batch entries, one launch per step.  E = [b][2][K][N] holds the special
// limbs only (the normal limbs of the extended polynomial are galois(ax) itself: NTT(INTT(x)) = x).
int dev_rotate_to(fhe_hip_ctx *c, u64 galois, u64 *out, const u64 *in, u32 Li, u32 batch) {
  const u64 N = c->N; const u32 K = c->K;
  auto it = c->rot_keys.find(galois);
  if (it == c->rot_keys.end()) FAIL(c, FHE_HIP_ERR_NOKEY, "rotation key not loaded for this galois element");
  u32 *perm = nullptr;
  if (galois != 1) TRY(get_galois_perm(c, galois, &perm));
  TRY(ensure_ws(c, 0, (size_t)batch * Li * N * 8));        // base-conversion digits of galois(ax), power basis
  TRY(ensure_ws(c, 4, (size_t)batch * 2 * K * N * 8));     // E
  TRY(ensure_ws(c, 2, (size_t)batch * 2 * Li * N * 8));    // conversions back to the normal basis, both halves
  u64 *pw = (u64 *)c->ws[0], *E = (u64 *)c->ws[4], *conv = (u64 *)c->ws[2];
  // power basis of galois(ax) (d[1]->backward(), :360), scaled so that it comes out as the mod-up digits [ax * hat_inv]_q
  NttPlan pi{in + (size_t)Li * N, pw, c->d_iota, 1, Li, Li, Li, batch, (u64)2 * Li * N, (u64)Li * N};
  pi.perm = perm;
  pi.inv_scale = c->d_modup_scale_m + (size_t)(Li - 2) * c->M;
  TRY(ntt_run(c, pi, true));
  TRY(dev_conv_to_special(c, E, 2 * K, 0, pw, Li, Li, batch, true));  // special limbs -> E[b][0]
  // forward NTT of the K special limbs (:363-366): column pass; the row pass, both key products (:373-384) and the row pass of
  // the two inverse NTTs (:386-392) run fused in rot_special_rows8_kernel, which leaves half 1 in E[b][1]
  NttPlan pf{E, E, c->d_iota + c->L, 1, K, K, K, batch, (u64)2 * K * N, (u64)2 * K * N};
  pf.cols_only = true;
  TRY(ntt_run(c, pf, false));
  RotSpArgs ra;
  ra.e0 = E; ra.e1 = E + (size_t)K * N; ra.estride = (u64)2 * K * N; ra.beta_m = it->second.beta_m; ra.alpha_m = it->second.alpha_m;
  ra.tw_fwd = c->d_tw_fwd; ra.tw_inv = c->d_tw_inv; ra.primes = c->d_primes;
  ra.first_mod = c->L; ra.mmap = nullptr; ra.first_limb = 0; ra.logN = c->logN; ra.logR = c->logR;
  const u32 R = 1u << c->logR, rows = R < 8 ? R : 8;
  LAUNCH_SP(c, "rot_special_rows", rot_special_rows8_kernel, dim3(R / rows, batch, K), dim3(32 * rows), ra);
  // from here on the two halves are 2*batch entries of K special limbs: column pass of their inverse NTTs (output = mod-down digits)
  NttPlan pc{E, E, c->d_iota + c->L, 1, K, K, K, 2 * batch, (u64)K * N, (u64)K * N};
  pc.cols_only = true; pc.inv_scale = c->d_moddown_scale_m;
  TRY(ntt_run(c, pc, true));
  // approximated_mod_down (base_converter.cpp:205-242): conversion to the normal basis, forward NTT, and the tail fused with
  // the normal-limb key product and "+ galois(bx)" in the row pass (NttArgs::epi == 4), written straight into out
  TRY(dev_conv_to_normal(c, conv, Li, E, K, 0, 2 * batch, true));
  NttPlan p{conv, conv, c->d_iota, 1, Li, Li, Li, 2 * batch, (u64)Li * N, (u64)Li * N};
  p.epi = 4; p.epi_src = in; p.epi_src_bstride = (u64)2 * Li * N; p.epi_half = (u64)Li * N; p.epi_c = c->d_Pinv_qj_m;
  p.epi_key0 = it->second.beta_m; p.epi_key1 = it->second.alpha_m; p.epi_add_perm = perm;
  p.epi_out = out; p.epi_out_bstride = (u64)Li * N;
  TRY(ntt_run(c, p, false));
  return FHE_HIP_OK;
}

// Evaluator::Impl::rotate_slots_inplace_non_ntt + apply_rotation_key_non_ntt (src/evaluator.cpp:327-339, 520-574): ct in the power
// basis, in place; reproduced literally, including approximated_mod_down adding its NTT-domain correction to power-basis limbs.
int dev_rotate_non_ntt(fhe_hip_ctx *c, u64 galois, u64 *ct, u32 Li, u32 batch) {
  const u64 N = c->N; const u32 K = c->K, nl = Li + K;
  auto it = c->rot_keys.find(galois);
  if (it == c->rot_keys.end()) FAIL(c, FHE_HIP_ERR_NOKEY, "rotation key not loaded for this galois element");
  const size_t ctb = (size_t)batch * 2 * Li * N * 8;
  TRY(ensure_ws(c, 3, ctb));                                   // g = galois(ct): [b][2][Li][N]
  TRY(ensure_ws(c, 4, (size_t)batch * 2 * nl * N * 8));        // E[b][2][Li+K][N]: the extended polynomial, then both key products
  TRY(ensure_ws(c, 2, (size_t)batch * 2 * Li * N * 8));        // conversions back to the normal basis, both halves
  u64 *g = (u64 *)c->ws[3], *E = (u64 *)c->ws[4], *conv = (u64 *)c->ws[2];
  if (galois == 1) {
    HIPCHK(c, (prof_chain_break(c), hipMemcpyAsync(g, ct, ctb, hipMemcpyDeviceToDevice, c->stream)));
  } else {
    LAUNCH(c, "galois_coeff", galois_coeff_kernel, dim3((unsigned)(N / PW_THREADS), 2 * Li, batch), dim3(PW_THREADS), g, (const u64 *)ct,
           (const PrimeC *)c->d_primes, c->logN, galois_inverse(galois, c->N), Li);
  }
  // ax_g is already in the power basis (:537): mod up into the special limbs of E[b][0], then the forward NTT of all Li+K limbs
  // (:539-542): the column pass reads the normal limbs from g where they lie (no copy into E) and the special limbs in place
  TRY(dev_conv_to_special(c, E, 2 * nl, Li, g + (size_t)Li * N, 2 * Li, Li, batch));
  const u32 *rotmap = c->d_rotmap + (size_t)Li * c->M;
  NttPlan pn{g + (size_t)Li * N, E, rotmap, 1, nl, Li, Li, batch, (u64)2 * Li * N, (u64)2 * nl * N};
  pn.cols_only = true;
  TRY(ntt_run(c, pn, false));
  if (K) {
    NttPlan ps{E, E, rotmap, 1, nl, nl, K, batch, (u64)2 * nl * N, (u64)2 * nl * N};
    ps.cols_only = true; ps.l0 = Li;
    TRY(ntt_run(c, ps, false));
  }
  // forward row pass, key product over all Li+K limbs (:544-556) and the row pass of both inverse NTTs (:557-567) on the same rows
  RotSpArgs ra;
  ra.e0 = E; ra.e1 = E + (size_t)nl * N; ra.estride = (u64)2 * nl * N; ra.beta_m = it->second.beta_m; ra.alpha_m = it->second.alpha_m;
  ra.tw_fwd = c->d_tw_fwd; ra.tw_inv = c->d_tw_inv; ra.primes = c->d_primes;
  ra.first_mod = 0; ra.mmap = rotmap; ra.first_limb = 0; ra.logN = c->logN; ra.logR = c->logR;
  const u32 R = 1u << c->logR, rows = R < 8 ? R : 8;
  LAUNCH_SP(c, "rot_special_rows", rot_special_rows8_kernel, dim3(R / rows, batch, nl), dim3(32 * rows), ra);
  // the two halves are now 2*batch entries of Li+K limbs: column pass of their inverse NTTs
  NttPlan pc{E, E, rotmap, 1, nl, nl, nl, 2 * batch, (u64)nl * N, (u64)nl * N};
  pc.cols_only = true;
  TRY(ntt_run(c, pc, true));
  // mod down both halves (one conversion launch) straight into ct (strided [b][2][Li][N]); rop[0] += galois(bx) read from g (:568-572)
  if (K) TRY(dev_conv_to_normal(c, conv, Li, E, nl, Li, 2 * batch, false));
  else HIPCHK(c, (prof_chain_break(c), hipMemsetAsync(conv, 0, (size_t)batch * 2 * Li * N * 8, c->stream)));
  for (int h = 0; h < 2; ++h) {
    u64 *cv = conv + (size_t)h * Li * N;
    NttPlan p{cv, cv, c->d_iota, 1, Li, Li, Li, batch, (u64)2 * Li * N, (u64)2 * Li * N};
    p.epi = 2; p.epi_src = E + (size_t)h * nl * N; p.epi_src_bstride = (u64)2 * nl * N; p.epi_c = c->d_Pinv_qj_m;
    p.epi_add = h == 0 ? g : nullptr; p.epi_add_bstride = (u64)2 * Li * N;
    p.epi_out = ct + (size_t)h * Li * N; p.epi_out_bstride = (u64)2 * Li * N;
    TRY(ntt_run(c, p, false));
  }
  return FHE_HIP_OK;
}

// rotate_slots_inplace(MixedRotKey) + apply_rotation_key(MixedRotKey) (evaluator.cpp:313-325, 403-516); ct in place
int dev_rotate_mixed(fhe_hip_ctx *c, u64 galois, u64 *ct, u32 Li, u32 batch) {
  const u64 N = c->N; const u32 nout = Li + 1;
  auto it = c->mixed_keys.find(galois);
  if (it == c->mixed_keys.end()) FAIL(c, FHE_HIP_ERR_NOKEY, "mixed rotation key not loaded for this galois element");
  if (c->mixed_128[galois] != (c->relin4 != 0)) FAIL(c, FHE_HIP_ERR_NOKEY, "mixed rotation key was loaded for the other key-switch kernel (FHE_HIP_TUNE_RELIN_KERNEL): load it again");
  const size_t ctb = (size_t)batch * 2 * Li * N * 8;
  TRY(ensure_ws(c, 3, ctb));                                       // g = galois(ct)
  TRY(ensure_ws(c, 0, (size_t)batch * Li * N * 8));                // power basis of the digits
  TRY(ensure_ws(c, 1, (size_t)batch * nout * Li * N * 8));         // t1
  TRY(ensure_ws(c, 4, (size_t)batch * 2 * nout * N * 8));          // ans_bx / ans_ax over Li+1 limbs
  u64 *g = (u64 *)c->ws[3], *pw = (u64 *)c->ws[0], *t = (u64 *)c->ws[1], *ans = (u64 *)c->ws[4];
  if (galois == 1) {
    HIPCHK(c, (prof_chain_break(c), hipMemcpyAsync(g, ct, ctb, hipMemcpyDeviceToDevice, c->stream)));
  } else {
    u32 *perm = nullptr;
    TRY(get_galois_perm(c, galois, &perm));
    LAUNCH(c, "galois_gather", galois_gather_kernel, dim3((unsigned)(N / PW_THREADS), 2 * Li, batch), dim3(PW_THREADS), g, (const u64 *)ct,
           (const u32 *)perm, N, (u64)2 * Li * N, (u64)2 * Li * N);
  }
  // [ct_1]_qi -> power basis (:425-427)
  NttPlan pi{g + (size_t)Li * N, pw, c->d_iota, 1, Li, Li, Li, batch, (u64)2 * Li * N, (u64)Li * N};
  TRY(ntt_run(c, pi, true));
  // NTT_j(digit i) for the Li+1 targets (limb id = j*Li + i), diagonal skipped (:436-437), fused with the key MAC
  const u32 *omap = c->d_rotmap + (size_t)Li * c->M;   // j < Li ? j : L
  RelinArgs ra;
  ra.out = ans;
  ra.diag = g + (size_t)Li * N; ra.diag_bstride = (u64)2 * Li * N;
  ra.add0 = nullptr; ra.add1 = nullptr; ra.add_bstride = 0;
  ra.beta_m = it->second.beta_m; ra.alpha_m = it->second.alpha_m; ra.klimbs = c->L + 1;
  TRY(run_digit_keyswitch(c, ra, pw, t, omap, nout, Li, batch, nout));
  // rescale by the special prime (:475-515): entries = 2*batch polys of Li+1 limbs, last limb lives under P[L]
  NttPlan pl{ans + (size_t)Li * N, ans + (size_t)Li * N, c->d_iota + c->L, 1, 1, 1, 1, 2 * batch, (u64)nout * N, (u64)nout * N};
  TRY(ntt_run(c, pl, true));
  // per half: tail_j = NTT_j(last limb) for j < Li with the rescale arithmetic (:485-515) as the epilogue of the row pass,
  // rop_j = (ans_j - tail_j) * P[L]^-1; the bx half also takes "+ galois(ct[0])" (:503-514) there.  Nothing is re-read.
  TRY(ensure_ws(c, 2, (size_t)batch * Li * N * 8));
  u64 *tail = (u64 *)c->ws[2];
  for (u32 h = 0; h < 2; ++h) {
    const u64 *half = ans + (size_t)h * nout * N;
    NttPlan ptl{half + (size_t)Li * N, tail, c->d_iota, 1, Li, 1, Li, batch, (u64)2 * nout * N, (u64)Li * N};
    ptl.epi = 1; ptl.epi_src = half; ptl.epi_src_bstride = (u64)2 * nout * N; ptl.epi_c = c->d_qinv_m + (size_t)c->L * c->M;
    ptl.epi_out = ct + (size_t)h * Li * N; ptl.epi_out_bstride = (u64)2 * Li * N;
    if (h == 0) { ptl.epi_add = g; ptl.epi_add_bstride = (u64)2 * Li * N; }
    TRY(ntt_run(c, ptl, false));
  }
  return FHE_HIP_OK;
}

}  // namespace

// =====================================================================================================
// C-ABI
// =====================================================================================================
extern "C" {

const char *fhe_hip_last_error(const fhe_hip_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int fhe_hip_ctx_create(fhe_hip_ctx **out, uint32_t logN, uint32_t L, uint32_t K, const uint64_t *primes, const uint64_t *psi, int device) {
  try {
  if (!out || !primes || !psi) { g_create_error = "null argument"; return FHE_HIP_ERR_ARG; }
  *out = nullptr;
  if (logN < 10 || logN > 16) { g_create_error = "logN must be in [10, 16]"; return FHE_HIP_ERR_RANGE; }
  if (L < 1 || L + K > 255) { g_create_error = "need 1 <= L and L + K <= 255"; return FHE_HIP_ERR_RANGE; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    g_create_error = "no usable HIP device (this library has no CPU fallback)";
    return FHE_HIP_ERR_NODEVICE;
  }
  fhe_hip_ctx *c = new fhe_hip_ctx();
  auto fail = [&](int rc) { g_create_error = c->err; fhe_hip_ctx_destroy(c); return rc; };
  c->device = device; c->logN = logN; c->L = L; c->K = K; c->M = L + K; c->N = 1ull << logN; c->logR = logN - 8;
  const uint32_t M = c->M; const u64 N = c->N;
  if (hipSetDevice(device) != hipSuccess) { c->err = "hipSetDevice failed"; return fail(FHE_HIP_ERR_HIP); }
  { int cus = 0; if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) != hipSuccess || cus <= 0) cus = 256; c->num_cus = cus; }
  c->primes.assign(primes, primes + M);
  c->sp = true;
  for (uint32_t m = 0; m < M; ++m) {
    const u64 p = primes[m];
    if (p >= (1ull << 62) || p < (1ull << 20) || (p - 1) % (2 * N) != 0) { c->err = "prime out of contract (p < 2^62, 2N | p-1)"; return fail(FHE_HIP_ERR_RANGE); }
    if (h_powmod(psi[m], N, p) != p - 1) { c->err = "psi is not a primitive 2N-th root of unity"; return fail(FHE_HIP_ERR_RANGE); }
    if ((p & 0xFFFFFFFFull) != 1) c->sp = false;
  }
  // measured (profiles/r03/ntt_streams.txt): the two-stream form gains 4 % on the SP kernels, which sit nearer the memory
  // system's limit, and nothing on the generic-prime kernels (30 % more instructions per butterfly for the same bytes)
  c->ntt_streams = c->sp ? 2 : 1;
  {
    // the lazy butterflies take residues of one prime of the table as inputs under another (relinearisation digits, rescale
    // tails, mixed rotation) and need them below 4p: primes of one table must lie within a factor 4 of each other; equal
    // primes have no inverse modulo each other
    u64 pmin = primes[0], pmax = primes[0];
    for (uint32_t m = 0; m < M; ++m) {
      if (primes[m] < pmin) pmin = primes[m];
      if (primes[m] > pmax) pmax = primes[m];
      for (uint32_t k = 0; k < m; ++k)
        if (primes[k] == primes[m]) { c->err = "duplicate prime in the table"; return fail(FHE_HIP_ERR_RANGE); }
    }
    if (pmax / 4 >= pmin) { c->err = "primes of one table must satisfy max(p) < 4 * min(p) (lazy reduction across primes)"; return fail(FHE_HIP_ERR_RANGE); }
  }
  {
    // this device's copy of the static scratch array (entry points that allocate nothing); the array's name is only a handle here
    void *sp = nullptr;
    const hipError_t e = hipGetSymbolAddress(&sp, HIP_SYMBOL(fhe_hip_scratch));
    if (e != hipSuccess || !sp) { c->err = std::string("hipGetSymbolAddress(fhe_hip_scratch): ") + hipGetErrorString(e); return fail(FHE_HIP_ERR_HIP); }
    c->scratch = (unsigned char *)sp;
  }
  std::vector<PrimeC> pc(M);
  std::vector<u64> twf((size_t)M * N), twi((size_t)M * N);
  for (uint32_t m = 0; m < M; ++m) {
    const u64 p = primes[m];
    PrimeC &P = pc[m];
    P.p = p; P.twop = 2 * p; P.p1 = (u32)(p >> 32);
    u64 inv = 1;  // Newton iteration for p^-1 mod 2^64
    for (int i = 0; i < 6; ++i) inv *= 2 - p * inv;
    P.pinv = inv; P.n32 = 0u - (u32)inv;
    P.r2 = (u64)((((u128)h_to_mont(1, p)) << 64) % p);
    P.ninv_m = h_to_mont(h_invmod(N % p, p), p);
    const u64 ps = psi[m] % p, psinv = h_invmod(ps, p);
    u64 f = h_to_mont(1, p), g = f;  // running powers kept in Montgomery form
    for (size_t k = 0; k < N; ++k) {
      const size_t r = h_brv(k, logN);
      twf[(size_t)m * N + r] = f; twi[(size_t)m * N + r] = g;
      f = h_mulmod(f, ps, p); g = h_mulmod(g, psinv, p);
    }
  }
  int rc;
  if ((rc = upload(c, &c->d_primes, pc)) || (rc = upload(c, &c->d_tw_fwd, twf)) || (rc = upload(c, &c->d_tw_inv, twi))) return fail(rc);
  std::vector<u32> iota(M), rotmap((size_t)(L + 1) * M, 0);
  for (uint32_t m = 0; m < M; ++m) iota[m] = m;
  for (uint32_t Li = 0; Li <= L; ++Li)
    for (uint32_t cm = 0; cm < Li + K && cm < M; ++cm) rotmap[(size_t)Li * M + cm] = cm < Li ? cm : L + (cm - Li);  // evaluator.cpp:374
  if ((rc = upload(c, &c->d_iota, iota)) || (rc = upload(c, &c->d_rotmap, rotmap))) return fail(rc);
  // BaseConverter tables (base_converter.cpp:122-187), stored in Montgomery form of the prime they multiply under
  const std::vector<u64> &P = c->primes;
  std::vector<u64> hat_pi_inv(K), neg_hat((size_t)L * K), Pinv(L);
  const size_t Lm1 = L > 1 ? L - 1 : 0;
  std::vector<u64> hat_qj_inv(Lm1 * L, 0), hat_qj_pi(Lm1 * K * L, 0), qinv((size_t)M * M, 0);
  for (uint32_t k = 0; k < K; ++k) {
    const uint32_t cm = L + k;
    hat_pi_inv[k] = h_to_mont(h_invmod(h_product_of(P, L, K, (int)k, cm), P[cm]), P[cm]);
    for (uint32_t l = 0; l < L; ++l) neg_hat[(size_t)l * K + k] = h_to_mont(h_to_mont(P[l] - h_product_of(P, L, K, (int)k, l), P[l]), P[l]);  // x 2^128: bconv_fold
  }
  for (uint32_t n = 2; n <= L; ++n) {
    const uint32_t l = n - 2;
    for (uint32_t j = 0; j < n; ++j) hat_qj_inv[(size_t)l * L + j] = h_to_mont(h_invmod(h_product_of(P, 0, n, (int)j, j), P[j]), P[j]);
    for (uint32_t k = 0; k < K; ++k)
      for (uint32_t j = 0; j < n; ++j) hat_qj_pi[((size_t)l * K + k) * L + j] = h_to_mont(h_to_mont(h_product_of(P, 0, n, (int)j, L + k), P[L + k]), P[L + k]);
  }
  for (uint32_t j = 0; j < L; ++j) Pinv[j] = h_to_mont(h_invmod(h_product_of(P, L, K, -1, j), P[j]), P[j]);
  for (uint32_t a = 0; a < M; ++a)
    for (uint32_t j = 0; j < M; ++j)
      if (a != j) qinv[(size_t)a * M + j] = h_to_mont(h_invmod(P[a] % P[j], P[j]), P[j]);  // math::inv_mod_prime(P[a], j), base_converter.cpp:261
  // inverse-NTT scale factors with the base-conversion digit constants folded in (the digits then cost nothing extra)
  std::vector<u64> modup_scale(Lm1 * M, 0), moddown_scale(M, 0);
  for (uint32_t n = 2; n <= L; ++n)
    for (uint32_t j = 0; j < n; ++j)
      modup_scale[(size_t)(n - 2) * M + j] = h_to_mont(h_mulmod(h_invmod(N % P[j], P[j]), h_invmod(h_product_of(P, 0, n, (int)j, j), P[j]), P[j]), P[j]);
  for (uint32_t k = 0; k < K; ++k)
    moddown_scale[L + k] = h_to_mont(h_mulmod(h_invmod(N % P[L + k], P[L + k]), h_invmod(h_product_of(P, L, K, (int)k, L + k), P[L + k]), P[L + k]), P[L + k]);
  if ((rc = upload(c, &c->d_modup_scale_m, modup_scale)) || (rc = upload(c, &c->d_moddown_scale_m, moddown_scale))) return fail(rc);
  if ((rc = upload(c, &c->d_hat_pi_inv_m, hat_pi_inv)) || (rc = upload(c, &c->d_neg_hat_pi_qj_m, neg_hat)) || (rc = upload(c, &c->d_Pinv_qj_m, Pinv)) ||
      (rc = upload(c, &c->d_hat_qj_inv_m, hat_qj_inv)) || (rc = upload(c, &c->d_hat_qj_pi_m, hat_qj_pi)) || (rc = upload(c, &c->d_qinv_m, qinv)))
    return fail(rc);
  c->h_neg_hat = neg_hat; c->h_hat_qj_pi = hat_qj_pi;
  {
    // bconv_mfma_kernel's accumulators start at 2^22 + a byte of D, D = -2^22 (2^64 - 1)/255 mod p: the eight byte weights
    // then carry a multiple of p into the sum
    std::vector<BconvTgt> tg(M);
    for (uint32_t m = 0; m < M; ++m) {
      const u64 p = primes[m];
      u64 b = 0;
      for (int s = 0; s < 8; ++s) b = (b + h_powmod(2, 22 + 8 * s, p)) % p;
      const u64 D = (p - b) % p;
      tg[m].p = p; tg[m].pinv = pc[m].pinv;
      for (int r = 0; r < 8; ++r) tg[m].init[r] = (1u << 22) + (u32)((D >> (8 * r)) & 0xFF);
    }
    if ((rc = upload(c, &c->d_bconv_tgt, tg))) return fail(rc);
  }
  *out = c;
  return FHE_HIP_OK;
  } catch (...) { return abi_exception(); }
}

void fhe_hip_ctx_destroy(fhe_hip_ctx *c) {
  if (!c) return;
  hipSetDevice(c->device);
  hipDeviceSynchronize();
  prof_drain(c);
  void *ptrs[] = {c->d_primes, c->d_tw_fwd, c->d_tw_inv, c->d_iota, c->d_rotmap, c->d_hat_pi_inv_m, c->d_neg_hat_pi_qj_m, c->d_Pinv_qj_m,
                  c->d_hat_qj_inv_m, c->d_hat_qj_pi_m, c->d_qinv_m, c->d_mk_beta, c->d_mk_alpha, c->d_modup_scale_m, c->d_moddown_scale_m, c->d_bconv_tgt};
  for (void *p : ptrs) if (p) hipFree(p);
  for (auto &kv : c->bconv_tables) hipFree(kv.second);
  for (auto &kv : c->rot_keys) { hipFree(kv.second.beta_m); hipFree(kv.second.alpha_m); }
  for (auto &kv : c->mixed_keys) { hipFree(kv.second.beta_m); hipFree(kv.second.alpha_m); }
  for (auto &kv : c->galois_perm) hipFree(kv.second);
  for (auto &kv : c->enc_half) hipFree(kv.second);
  { void *enc[] = {c->d_enc_roots, c->d_enc_map, c->d_enc_prod_m, c->d_enc_ginv_m, c->d_enc_qd}; for (void *p : enc) if (p) hipFree(p); }
  for (void *p : c->ws) if (p) hipFree(p);
  for (void *p : c->stage) if (p) hipFree(p);
  if (c->d_idx) hipFree(c->d_idx);
  for (int k = 0; k < 2; ++k) { if (c->ntt_aux[k]) hipStreamDestroy(c->ntt_aux[k]); if (c->ntt_join[k]) hipEventDestroy(c->ntt_join[k]); }
  if (c->ntt_fork) hipEventDestroy(c->ntt_fork);
  for (int k = 0; k < 3; ++k) {
    for (int j = 0; j < 3; ++j) if (c->br_buf[k][j]) hipFree(c->br_buf[k][j]);
    for (int l = 0; l < 2; ++l) if (c->br_ev_up[l][k]) hipEventDestroy(c->br_ev_up[l][k]);
    if (c->br_ev_done[k]) hipEventDestroy(c->br_ev_done[k]);
  }
  for (int l = 0; l < 2; ++l) {
    if (c->br_up[l]) hipStreamDestroy(c->br_up[l]);
    if (c->br_down[l]) hipStreamDestroy(c->br_down[l]);
  }
  delete c;
}

int fhe_hip_set_stream(fhe_hip_ctx *c, void *s) { TRY(check_ctx(c)); c->stream = (hipStream_t)s; return FHE_HIP_OK; }
int fhe_hip_sync(fhe_hip_ctx *c) { TRY(check_ctx(c)); HIPCHK(c, hipStreamSynchronize(c->stream)); return FHE_HIP_OK; }

constexpr size_t NTT_OVERLAP_CHUNK = 128;  // limbs per chunk of the two-stream form: a 64 MB intermediate at N = 2^16 (profiles/r02/ntt_streams.txt)
static int ntt_api(fhe_hip_ctx *c, uint64_t *limbs, const uint32_t *moduli_idx, size_t n_limbs, int mem_kind, bool inverse) {
  try {
  TRY(check_ctx(c));
  if (n_limbs == 0) return FHE_HIP_OK;  // an empty batch is a no-op, whatever the pointers
  if (!limbs || !moduli_idx) FAIL(c, FHE_HIP_ERR_ARG, "null argument");
  for (size_t i = 0; i < n_limbs; ++i) if (moduli_idx[i] >= c->M) FAIL(c, FHE_HIP_ERR_RANGE, "moduli_idx out of range");
  Staged st(c, mem_kind);
  u64 *d = nullptr;
  TRY(st.out(limbs, n_limbs * c->N * 8, &d, true));
  if (c->h_idx.size() != n_limbs || std::memcmp(c->h_idx.data(), moduli_idx, n_limbs * sizeof(u32)) != 0) {
    HIPCHK(c, hipStreamSynchronize(c->stream));  // a previous upload or a kernel may still read the old copies
    if (c->d_idx_cap < n_limbs) {
      if (c->d_idx) { HIPCHK(c, hipFree(c->d_idx)); c->d_idx = nullptr; c->d_idx_cap = 0; }
      HIPCHK(c, hipMalloc((void **)&c->d_idx, n_limbs * sizeof(u32)));
      c->d_idx_cap = n_limbs;
    }
    c->h_idx.assign(moduli_idx, moduli_idx + n_limbs);
    HIPCHK(c, (prof_chain_break(c), hipMemcpyAsync(c->d_idx, c->h_idx.data(), n_limbs * sizeof(u32), hipMemcpyHostToDevice, c->stream)));
  }
  u32 *dm = c->d_idx;
  // Two passes over memory (N >= 2^15) and enough limbs: overlapped chunks on the two library-owned streams.  Kernel boundaries
  // order each chunk's passes (same stream) and make its intermediate visible; chunks touch disjoint limbs.
  const size_t ochunk = c->ntt_chunk ? c->ntt_chunk : NTT_OVERLAP_CHUNK;
  if (c->ntt_streams >= 2 && c->logR >= 7 && n_limbs >= 2 * ochunk) {
    if (!c->ntt_aux[0]) {
      for (int k = 0; k < 2; ++k) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->ntt_aux[k], hipStreamNonBlocking));
        HIPCHK(c, hipEventCreateWithFlags(&c->ntt_join[k], hipEventDisableTiming));
      }
      HIPCHK(c, hipEventCreateWithFlags(&c->ntt_fork, hipEventDisableTiming));
    }
    // per-launch timing would time kernels that overlap each other: the whole call is one profile entry instead
    // (first launch -> last completion, measured on the caller's stream around the fork and the join)
    const bool prof = c->prof;
    hipEvent_t ev0 = nullptr;
    if (prof) { ev0 = prof_event(c); c->prof = false; }
    hipStream_t user = c->stream;
    int rc = FHE_HIP_OK;
    hipError_t e = hipEventRecord(c->ntt_fork, user);
    for (int k = 0; k < 2 && e == hipSuccess; ++k) e = hipStreamWaitEvent(c->ntt_aux[k], c->ntt_fork, 0);
    size_t i = 0;
    for (size_t off = 0; off < n_limbs && e == hipSuccess && rc == FHE_HIP_OK; off += ochunk, ++i) {
      const u32 n = (u32)((n_limbs - off < ochunk) ? n_limbs - off : ochunk);
      NttPlan p{d + off * c->N, d + off * c->N, dm + off, 1, n, n, n, 1, 0, 0};
      c->stream = c->ntt_aux[i & 1];
      rc = ntt_run(c, p, inverse);
      c->stream = user;
    }
    // join on every path, so that the caller's stream never runs ahead of work already enqueued on the side streams
    for (int k = 0; k < 2; ++k) {
      hipError_t e2 = hipEventRecord(c->ntt_join[k], c->ntt_aux[k]);
      if (e2 == hipSuccess) e2 = hipStreamWaitEvent(user, c->ntt_join[k], 0);
      if (e == hipSuccess) e = e2;
    }
    c->prof = prof;
    if (e != hipSuccess) { c->err = std::string("two-stream NTT: ") + hipGetErrorString(e); return FHE_HIP_ERR_HIP; }
    TRY(rc);
    if (prof) {
      hipEvent_t ev1 = prof_event(c);
      prof_entry(c, inverse ? "ntt_inv_overlapped" : "ntt_fwd_overlapped")->pending.push_back({ev0, ev1});
      c->chain_valid = false;
    }
    return st.finish();
  }
  const size_t step = c->ntt_chunk ? c->ntt_chunk : 32768;
  for (size_t off = 0; off < n_limbs; off += step) {
    const u32 n = (u32)((n_limbs - off < step) ? n_limbs - off : step);
    NttPlan p{d + off * c->N, d + off * c->N, dm + off, 1, n, n, n, 1, 0, 0};
    TRY(ntt_run(c, p, inverse));
  }
  return st.finish();
  } catch (...) { return abi_exception(); }
}
int fhe_hip_ntt_forward(fhe_hip_ctx *c, uint64_t *limbs, const uint32_t *idx, size_t n, int mk) { return ntt_api(c, limbs, idx, n, mk, false); }
int fhe_hip_ntt_backward(fhe_hip_ctx *c, uint64_t *limbs, const uint32_t *idx, size_t n, int mk) { return ntt_api(c, limbs, idx, n, mk, true); }

static int galois_api(fhe_hip_ctx *c, uint64_t *polys, uint32_t nlimbs, uint64_t galois, size_t batch, int mem_kind, bool ntt_domain) {
  try {
  TRY(check_ctx(c));
  if (!polys) FAIL(c, FHE_HIP_ERR_ARG, "null argument");
  if (!(galois & 1) || galois >= 2 * c->N) FAIL(c, FHE_HIP_ERR_ARG, "galois must be odd and < 2N");
  if (nlimbs == 0 || nlimbs > c->M) FAIL(c, FHE_HIP_ERR_RANGE, "bad limb count");
  if (galois == 1 || batch == 0) return FHE_HIP_OK;  // src/yell.cpp:8-9
  const size_t bytes = batch * nlimbs * c->N * 8;
  Staged st(c, mem_kind);
  u64 *d = nullptr;
  TRY(st.out(polys, bytes, &d, true));
  TRY(ensure_ws(c, 0, bytes));
  u64 *tmp = (u64 *)c->ws[0];
  HIPCHK(c, (prof_chain_break(c), hipMemcpyAsync(tmp, d, bytes, hipMemcpyDeviceToDevice, c->stream)));
  dim3 grid((unsigned)(c->N / PW_THREADS), nlimbs, (unsigned)batch);
  if (ntt_domain) {
    u32 *perm = nullptr;
    TRY(get_galois_perm(c, galois, &perm));
    LAUNCH(c, "galois_gather", galois_gather_kernel, grid, dim3(PW_THREADS), d, (const u64 *)tmp, (const u32 *)perm, c->N, (u64)nlimbs * c->N,
           (u64)nlimbs * c->N);
  } else {
    LAUNCH(c, "galois_coeff", galois_coeff_kernel, grid, dim3(PW_THREADS), d, (const u64 *)tmp, (const PrimeC *)c->d_primes, c->logN, galois_inverse(galois, c->N), nlimbs);
  }
  return st.finish();
  } catch (...) { return abi_exception(); }
}
int fhe_hip_apply_galois(fhe_hip_ctx *c, uint64_t *p, uint32_t nl, uint64_t g, size_t b, int mk) { return galois_api(c, p, nl, g, b, mk, true); }
int fhe_hip_apply_galois_non_ntt(fhe_hip_ctx *c, uint64_t *p, uint32_t nl, uint64_t g, size_t b, int mk) { return galois_api(c, p, nl, g, b, mk, false); }

int fhe_hip_mod_up(fhe_hip_ctx *c, uint64_t *rop, const uint64_t *op, uint32_t Li, size_t batch, int mem_kind) {
  try {
  TRY(check_ctx(c));
  if (!rop || !op) FAIL(c, FHE_HIP_ERR_ARG, "null argument");
  if (Li > c->L) FAIL(c, FHE_HIP_ERR_RANGE, "Only support mod up from normal primes");
  if (Li <= 1) FAIL(c, FHE_HIP_ERR_RANGE, "nmoduli <= 1");  // base_converter.cpp:284-285
  if (batch == 0) return FHE_HIP_OK;
  const u64 N = c->N; const u32 nl = Li + c->K;
  if (c->K == 0) {  // no special primes: the result is the copy of the normal limbs (:196-197), no conversion to launch
    if (mem_kind == FHE_HIP_MEM_DEVICE) HIPCHK(c, (prof_chain_break(c), hipMemcpyAsync(rop, op, batch * Li * N * 8, hipMemcpyDeviceToDevice, c->stream)));
    else std::memmove(rop, op, batch * Li * N * 8);
    return FHE_HIP_OK;
  }
  Staged st(c, mem_kind);
  const u64 *dop = nullptr; u64 *drop = nullptr;
  TRY(st.in(op, batch * Li * N * 8, &dop));
  TRY(st.out(rop, batch * nl * N * 8, &drop, false));
  for (size_t b = 0; b < batch; ++b)
    HIPCHK(c, (prof_chain_break(c), hipMemcpyAsync(drop + b * nl * N, dop + b * Li * N, (size_t)Li * N * 8, hipMemcpyDeviceToDevice, c->stream)));
  TRY(dev_conv_to_special(c, drop, nl, Li, dop, Li, Li, (u32)batch));
  return st.finish();
  } catch (...) { return abi_exception(); }
}

int fhe_hip_mod_down(fhe_hip_ctx *c, uint64_t *rop, const uint64_t *op, uint32_t L0, size_t batch, int mem_kind) {
  try {
  TRY(check_ctx(c));
  if (!rop || !op) FAIL(c, FHE_HIP_ERR_ARG, "null argument");
  if (L0 < 1 || L0 > c->L || c->K < 1) FAIL(c, FHE_HIP_ERR_RANGE, "bad limb count");
  if (batch == 0) return FHE_HIP_OK;
  const u64 N = c->N;
  Staged st(c, mem_kind);
  const u64 *dop = nullptr; u64 *drop = nullptr;
  TRY(st.in(op, batch * (L0 + c->K) * N * 8, &dop));
  TRY(st.out(rop, batch * L0 * N * 8, &drop, false));
  TRY(dev_mod_down(c, drop, dop, L0, (u32)batch, nullptr, 2));
  return st.finish();
  } catch (...) { return abi_exception(); }
}

int fhe_hip_conv_to_normal(fhe_hip_ctx *c, uint64_t *rop, const uint64_t *sp, uint32_t L0, size_t batch, int mem_kind) {
  try {
  TRY(check_ctx(c));
  if (!rop || !sp) FAIL(c, FHE_HIP_ERR_ARG, "null argument");
  if (L0 < 1 || L0 > c->L) FAIL(c, FHE_HIP_ERR_RANGE, "target limb count out of range");
  if (c->K < 1) FAIL(c, FHE_HIP_ERR_RANGE, "conversion from the special basis needs special primes");
  if (batch == 0) return FHE_HIP_OK;
  Staged st(c, mem_kind);
  const u64 *dsp = nullptr; u64 *drop = nullptr;
  TRY(st.in(sp, batch * c->K * c->N * 8, &dsp));
  TRY(st.out(rop, batch * L0 * c->N * 8, &drop, false));
  TRY(dev_conv_to_normal(c, drop, L0, dsp, c->K, 0, (u32)batch));
  return st.finish();
  } catch (...) { return abi_exception(); }
}

int fhe_hip_conv_to_special(fhe_hip_ctx *c, uint64_t *rop, const uint64_t *op, uint32_t Li, uint32_t sp_index, size_t batch, int mem_kind) {
  try {
  TRY(check_ctx(c));
  if (!rop || !op) FAIL(c, FHE_HIP_ERR_ARG, "null argument");
  if (Li > c->L) FAIL(c, FHE_HIP_ERR_RANGE, "Only support mod up from normal primes");
  if (Li <= 1) FAIL(c, FHE_HIP_ERR_RANGE, "nmoduli <= 1");  // base_converter.cpp:284-285
  if (sp_index >= c->K) FAIL(c, FHE_HIP_ERR_RANGE, "special prime index out of range");
  if (batch == 0) return FHE_HIP_OK;
  Staged st(c, mem_kind);
  const u64 *dop = nullptr; u64 *drop = nullptr;
  TRY(st.in(op, batch * Li * c->N * 8, &dop));
  TRY(st.out(rop, batch * c->N * 8, &drop, false));
  TRY(dev_conv_to_special_one(c, drop, dop, Li, sp_index, (u32)batch));
  return st.finish();
  } catch (...) { return abi_exception(); }
}

int fhe_hip_mem_alloc(fhe_hip_ctx *c, uint64_t **dptr, size_t words) {
  try {
  TRY(check_ctx(c));
  if (!dptr) FAIL(c, FHE_HIP_ERR_ARG, "null argument");
  *dptr = nullptr;
  HIPCHK(c, hipMalloc((void **)dptr, (words ? words : 1) * sizeof(uint64_t)));
  return FHE_HIP_OK;
  } catch (...) { return abi_exception(); }
}
int fhe_hip_mem_free(fhe_hip_ctx *c, uint64_t *dptr) {
  try {
  TRY(check_ctx(c));
  if (!dptr) return FHE_HIP_OK;
  HIPCHK(c, hipStreamSynchronize(c->stream));  // work enqueued on the context's stream may still use the block
  HIPCHK(c, hipFree(dptr));
  return FHE_HIP_OK;
  } catch (...) { return abi_exception(); }
}
int fhe_hip_mem_copy(fhe_hip_ctx *c, void *dst, const void *src, size_t bytes, int direction) {
  try {
  TRY(check_ctx(c));
  if (bytes == 0) return FHE_HIP_OK;
  if (!dst || !src) FAIL(c, FHE_HIP_ERR_ARG, "null argument");
  hipMemcpyKind kind;
  if (direction == FHE_HIP_COPY_H2D) kind = hipMemcpyHostToDevice;
  else if (direction == FHE_HIP_COPY_D2H) kind = hipMemcpyDeviceToHost;
  else if (direction == FHE_HIP_COPY_D2D) kind = hipMemcpyDeviceToDevice;
  else FAIL(c, FHE_HIP_ERR_ARG, "unknown copy direction");
  HIPCHK(c, (prof_chain_break(c), hipMemcpyAsync(dst, src, bytes, kind, c->stream)));
  if (direction != FHE_HIP_COPY_D2D) HIPCHK(c, hipStreamSynchronize(c->stream));  // the host buffer is the caller's again on return
  return FHE_HIP_OK;
  } catch (...) { return abi_exception(); }
}

int fhe_hip_rescale(fhe_hip_ctx *c, uint64_t *rop, const uint64_t *op, uint32_t n, size_t batch, int mem_kind) {
  try {
  TRY(check_ctx(c));
  if (!rop || !op) FAIL(c, FHE_HIP_ERR_ARG, "null argument");
  if (n < 2 || n > c->M) FAIL(c, FHE_HIP_ERR_RANGE, "Already in the last moduli.");
  if (batch == 0) return FHE_HIP_OK;
  const u64 N = c->N;
  Staged st(c, mem_kind);
  const u64 *dop = nullptr; u64 *drop = nullptr;
  TRY(st.in(op, batch * n * N * 8, &dop));
  TRY(st.out(rop, batch * (n - 1) * N * 8, &drop, false));
  TRY(dev_rescale(c, drop, dop, n, (u32)batch, 2));
  return st.finish();
  } catch (...) { return abi_exception(); }
}

// key copy in Montgomery form (times 2^64), or times 2^128 for the kernel whose fold divides by 2^128 (relin_rows4_mac_kernel)
static int load_key(fhe_hip_ctx *c, u64 **dst, const uint64_t *src, size_t limbs, const u32 *midx, u32 mper, int mem_kind, bool twice = false) {
  const size_t bytes = limbs * c->N * 8;
  if (!*dst) HIPCHK(c, hipMalloc((void **)dst, bytes));
  HIPCHK(c, (prof_chain_break(c), hipMemcpyAsync(*dst, src, bytes, mem_kind == FHE_HIP_MEM_DEVICE ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->stream)));
  for (int k = 0; k < (twice ? 2 : 1); ++k)
    LAUNCH_SP(c, "to_mont", to_mont_kernel, pw_grid(c, (u32)limbs, 1), dim3(PW_THREADS), *dst, (const PrimeC *)c->d_primes, midx, mper, c->N);
  return FHE_HIP_OK;
}

int fhe_hip_load_mult_key(fhe_hip_ctx *c, const uint64_t *beta, const uint64_t *alpha, int mem_kind) {
  try {
  TRY(check_ctx(c));
  if (!beta || !alpha) FAIL(c, FHE_HIP_ERR_ARG, "null argument");
  // [L][L][N]: limb id = j*L + i uses modulus i = id % L
  TRY(load_key(c, &c->d_mk_beta, beta, (size_t)c->L * c->L, c->d_iota, c->L, mem_kind, c->relin4 != 0));
  TRY(load_key(c, &c->d_mk_alpha, alpha, (size_t)c->L * c->L, c->d_iota, c->L, mem_kind, c->relin4 != 0));
  c->mk_128 = c->relin4 != 0;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return FHE_HIP_OK;
  } catch (...) { return abi_exception(); }
}

int fhe_hip_load_rot_key(fhe_hip_ctx *c, uint64_t galois, const uint64_t *beta, const uint64_t *alpha, int mem_kind) {
  try {
  TRY(check_ctx(c));
  if (!beta || !alpha) FAIL(c, FHE_HIP_ERR_ARG, "null argument");
  if (c->K < 1) FAIL(c, FHE_HIP_ERR_RANGE, "rotation keys need special primes");
  if (galois != 0 && (!(galois & 1) || galois >= 2 * c->N)) FAIL(c, FHE_HIP_ERR_ARG, "galois must be odd and < 2N");
  RotKeyDev &rk = c->rot_keys[galois];
  TRY(load_key(c, &rk.beta_m, beta, c->M, c->d_iota, c->M, mem_kind));
  TRY(load_key(c, &rk.alpha_m, alpha, c->M, c->d_iota, c->M, mem_kind));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return FHE_HIP_OK;
  } catch (...) { return abi_exception(); }
}

int fhe_hip_add(fhe_hip_ctx *c, uint64_t *rop, const uint64_t *op, uint32_t nlimbs, size_t batch, int mem_kind) {
  try {
  TRY(check_ctx(c));
  if (!rop || !op) FAIL(c, FHE_HIP_ERR_ARG, "null argument");
  if (nlimbs < 1 || nlimbs > c->M) FAIL(c, FHE_HIP_ERR_RANGE, "bad limb count");
  if (batch == 0) return FHE_HIP_OK;
  const size_t bytes = batch * nlimbs * c->N * 8;
  Staged st(c, mem_kind);
  const u64 *dop = nullptr; u64 *drop = nullptr;
  TRY(st.in(op, bytes, &dop));
  TRY(st.out(rop, bytes, &drop, true));
  LAUNCH(c, "add", add_kernel, pw_grid(c, nlimbs, (u32)batch), dim3(PW_THREADS), drop, dop, (const PrimeC *)c->d_primes, c->N);
  return st.finish();
  } catch (...) { return abi_exception(); }
}

int fhe_hip_multiply_plain(fhe_hip_ctx *c, uint64_t *ct, const uint64_t *pt, uint32_t nlimbs, size_t batch, int mem_kind) {
  try {
  TRY(check_ctx(c));
  if (!ct || !pt) FAIL(c, FHE_HIP_ERR_ARG, "null argument");
  if (nlimbs < 1 || nlimbs > c->M) FAIL(c, FHE_HIP_ERR_RANGE, "bad limb count");
  if (batch == 0) return FHE_HIP_OK;
  Staged st(c, mem_kind);
  const u64 *dpt = nullptr; u64 *dct = nullptr;
  TRY(st.in(pt, (size_t)nlimbs * c->N * 8, &dpt));
  TRY(st.out(ct, batch * 2 * nlimbs * c->N * 8, &dct, true));
  LAUNCH_SP(c, "mul_plain", mul_plain_kernel, pw_grid(c, 2 * nlimbs, (u32)batch), dim3(PW_THREADS), dct, dpt, (const PrimeC *)c->d_primes, nlimbs, c->N);
  return st.finish();
  } catch (...) { return abi_exception(); }
}

static int check_li(fhe_hip_ctx *c, uint32_t Li, uint32_t min) {
  if (Li > c->L) FAIL(c, FHE_HIP_ERR_RANGE, "Too many moduli.");
  if (Li < min) FAIL(c, FHE_HIP_ERR_RANGE, "too few moduli");
  return FHE_HIP_OK;
}

int fhe_hip_cross_mult(fhe_hip_ctx *c, uint64_t *out3, const uint64_t *c0, const uint64_t *c1, uint32_t Li, size_t batch, int mem_kind) {
  try {
  TRY(check_ctx(c));
  if (!out3 || !c0 || !c1) FAIL(c, FHE_HIP_ERR_ARG, "null argument");
  TRY(check_li(c, Li, 1));
  if (batch == 0) return FHE_HIP_OK;
  const size_t ctb = batch * 2 * Li * c->N * 8;
  Staged st(c, mem_kind);
  const u64 *d0 = nullptr, *d1 = nullptr; u64 *dout = nullptr;
  TRY(st.in(c0, ctb, &d0)); TRY(st.in(c1, ctb, &d1));
  TRY(st.out(out3, ctb / 2 * 3, &dout, false));
  LAUNCH_SP(c, "cross_mult", cross_mult_kernel, pw_grid(c, Li, (u32)batch), dim3(PW_THREADS), dout, d0, d1, (const PrimeC *)c->d_primes, Li, c->N);
  return st.finish();
  } catch (...) { return abi_exception(); }
}

int fhe_hip_relin_only(fhe_hip_ctx *c, uint64_t *out, const uint64_t *ct3, uint32_t Li, size_t batch, int mem_kind) {
  try {
  TRY(check_ctx(c));
  if (!out || !ct3) FAIL(c, FHE_HIP_ERR_ARG, "null argument");
  TRY(check_li(c, Li, 1));
  if (batch == 0) return FHE_HIP_OK;
  const size_t limb_b = (size_t)Li * c->N * 8;
  Staged st(c, mem_kind);
  const u64 *din = nullptr; u64 *dout = nullptr;
  TRY(st.in(ct3, batch * 3 * limb_b, &din));
  TRY(st.out(out, batch * 2 * limb_b, &dout, false));
  TRY(dev_relin(c, dout, din, Li, (u32)batch));
  return st.finish();
  } catch (...) { return abi_exception(); }
}

int fhe_hip_relinearize(fhe_hip_ctx *c, uint64_t *out, const uint64_t *ct3, uint32_t Li, size_t batch, int mem_kind) {
  try {
  TRY(check_ctx(c));
  if (!out || !ct3) FAIL(c, FHE_HIP_ERR_ARG, "null argument");
  TRY(check_li(c, Li, 2));
  if (batch == 0) return FHE_HIP_OK;
  const u64 N = c->N;
  Staged st(c, mem_kind);
  const u64 *din = nullptr; u64 *dout = nullptr;
  TRY(st.in(ct3, batch * 3 * Li * N * 8, &din));
  TRY(st.out(out, batch * 2 * (Li - 1) * N * 8, &dout, false));
  TRY(ensure_ws(c, 4, batch * 2 * Li * N * 8));
  u64 *rl = (u64 *)c->ws[4];
  TRY(dev_relin(c, rl, din, Li, (u32)batch));
  TRY(dev_rescale_ntt(c, dout, rl, Li, (u32)batch));
  return st.finish();
  } catch (...) { return abi_exception(); }
}

int fhe_hip_mul_relin_rescale(fhe_hip_ctx *c, uint64_t *out, const uint64_t *c0, const uint64_t *c1, uint32_t Li, size_t batch, int mem_kind) {
  try {
  TRY(check_ctx(c));
  if (!out || !c0 || !c1) FAIL(c, FHE_HIP_ERR_ARG, "null argument");
  TRY(check_li(c, Li, 2));
  if (batch == 0) return FHE_HIP_OK;
  const u64 N = c->N;
  const size_t ctb = batch * 2 * Li * N * 8;
  Staged st(c, mem_kind);
  const u64 *d0 = nullptr, *d1 = nullptr; u64 *dout = nullptr;
  TRY(st.in(c0, ctb, &d0)); TRY(st.in(c1, ctb, &d1));
  TRY(st.out(out, batch * 2 * (Li - 1) * N * 8, &dout, false));
  TRY(dev_mul_relin_rescale(c, dout, d0, d1, Li, (u32)batch));
  return st.finish();
  } catch (...) { return abi_exception(); }
}

// fhe_hip_mul_relin_rescale for callers that cannot allocate: same kernels, every workspace carved out of this device's copy
// of fhe_hip_scratch (address taken in fhe_hip_ctx_create).  Calls no hipMalloc / hipFree / event creation; device pointers only.
namespace {
// sets the context's grow-only workspaces aside and routes ensure_ws to the static scratch array; puts everything back on
// every exit path (return, error, exception)
struct FixedWsScope {
  fhe_hip_ctx *c; void *saved_ws[6]; size_t saved_bytes[6];
  explicit FixedWsScope(fhe_hip_ctx *ctx) : c(ctx) {
    for (int k = 0; k < 6; ++k) { saved_ws[k] = c->ws[k]; saved_bytes[k] = c->ws_bytes[k]; c->ws[k] = nullptr; c->ws_bytes[k] = 0; }
    c->fixed_ws = true; c->fixed_used = 0;
  }
  ~FixedWsScope() {
    c->fixed_ws = false; c->fixed_used = 0;
    for (int k = 0; k < 6; ++k) { c->ws[k] = saved_ws[k]; c->ws_bytes[k] = saved_bytes[k]; }
  }
};
}  // namespace

int fhe_hip_mul_relin_rescale_noalloc(fhe_hip_ctx *c, uint64_t *out, const uint64_t *c0, const uint64_t *c1, uint32_t Li, size_t batch) {
  try {
  if (!c) return FHE_HIP_ERR_ARG;
  TRY(check_ctx(c));
  if (!out || !c0 || !c1) FAIL(c, FHE_HIP_ERR_ARG, "null argument");
  TRY(check_li(c, Li, 2));
  if (batch == 0) return FHE_HIP_OK;
  if (!c->d_mk_beta) FAIL(c, FHE_HIP_ERR_NOKEY, "mult key not loaded");
  // workspaces besides the relinearisation's own: ct3 (3 Li limbs) + relinearised ct (2 Li) + rescale tails (2 (Li-1)) per op;
  // dev_relin adds the power-basis digits (Li) and at least one group of column-pass output (Li)
  const size_t limb = (size_t)c->N * 8;
  if (batch > (size_t)FHE_HIP_NOALLOC_SCRATCH_BYTES / ((size_t)(9 * Li - 2) * limb))
    FAIL(c, FHE_HIP_ERR_RANGE, "batch * (9 Li - 2) * N * 8 bytes exceeds FHE_HIP_NOALLOC_SCRATCH_BYTES: split the batch");
  // the grow-only workspaces of the allocating entry points are set aside for the duration of this call
  FixedWsScope scope(c);
  return dev_mul_relin_rescale(c, (u64 *)out, (const u64 *)c0, (const u64 *)c1, Li, (u32)batch);
  } catch (...) { return abi_exception(); }
}

int fhe_hip_set_tuning(fhe_hip_ctx *c, int knob, uint64_t value) {
  TRY(check_ctx(c));
  if (knob == FHE_HIP_TUNE_RELIN_GROUP) { c->relin_group = value > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)value; return FHE_HIP_OK; }
  if (knob == FHE_HIP_TUNE_ONEPASS) { c->onepass = value ? 1u : 0u; return FHE_HIP_OK; }
  if (knob == FHE_HIP_TUNE_NTT_CHUNK) { c->ntt_chunk = value > 32768 ? 32768u : (uint32_t)value; return FHE_HIP_OK; }
  if (knob == FHE_HIP_TUNE_RELIN_KERNEL) { c->relin4 = value ? 1u : 0u; return FHE_HIP_OK; }
  if (knob == FHE_HIP_TUNE_BCONV) { if (value > 2) FAIL(c, FHE_HIP_ERR_RANGE, "FHE_HIP_TUNE_BCONV takes 0, 1 or 2"); c->bconv_mode = value; return FHE_HIP_OK; }
  if (knob == FHE_HIP_TUNE_NTT_STREAMS) { c->ntt_streams = value >= 2 ? 2u : 1u; return FHE_HIP_OK; }
  FAIL(c, FHE_HIP_ERR_ARG, "unknown tuning knob");
}

static int rotate_check(fhe_hip_ctx *c, uint64_t galois, uint32_t Li) {
  if (!(galois & 1) || galois >= 2 * c->N) FAIL(c, FHE_HIP_ERR_ARG, "galois must be odd and < 2N");
  TRY(check_li(c, Li, 2));  // Li <= 1: mod_up throws (base_converter.cpp:284-285)
  if (c->K < 1) FAIL(c, FHE_HIP_ERR_RANGE, "rotation needs special primes");
  return FHE_HIP_OK;
}

int fhe_hip_rotate_to(fhe_hip_ctx *c, uint64_t galois, uint64_t *out, const uint64_t *in, uint32_t Li, size_t batch, int mem_kind) {
  try {
  TRY(check_ctx(c));
  if (!out || !in) FAIL(c, FHE_HIP_ERR_ARG, "null argument");
  TRY(check_li(c, Li, 1));
  const size_t bytes = batch * 2 * Li * c->N * 8;
  {
    // out of place means disjoint: the last kernel still gathers `in` through the galois table while it writes `out`
    const uintptr_t a = (uintptr_t)in, b = (uintptr_t)out;
    if (a == b || (bytes && a < b + bytes && b < a + bytes))
      FAIL(c, FHE_HIP_ERR_ARG, "fhe_hip_rotate_to is out of place: out must not overlap in (fhe_hip_rotate works in place)");
  }
  if (galois == 0) {  // no rotation (evaluator.cpp:303-304): the result is the input
    if (batch == 0) return FHE_HIP_OK;
    if (mem_kind == FHE_HIP_MEM_DEVICE) HIPCHK(c, (prof_chain_break(c), hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, c->stream)));
    else std::memmove(out, in, bytes);
    return FHE_HIP_OK;
  }
  TRY(rotate_check(c, galois, Li));
  if (batch == 0) return FHE_HIP_OK;
  Staged st(c, mem_kind);
  const u64 *din = nullptr; u64 *dout = nullptr;
  TRY(st.in(in, bytes, &din));
  TRY(st.out(out, bytes, &dout, false));
  TRY(dev_rotate_to(c, galois, dout, din, Li, (u32)batch));
  return st.finish();
  } catch (...) { return abi_exception(); }
}

int fhe_hip_rotate(fhe_hip_ctx *c, uint64_t galois, uint64_t *ct, uint32_t Li, size_t batch, int mem_kind) {
  try {
  TRY(check_ctx(c));
  if (!ct) FAIL(c, FHE_HIP_ERR_ARG, "null argument");
  if (galois == 0) return FHE_HIP_OK;  // evaluator.cpp:303-304
  TRY(rotate_check(c, galois, Li));
  if (batch == 0) return FHE_HIP_OK;
  const size_t bytes = batch * 2 * Li * c->N * 8;
  if (mem_kind == FHE_HIP_MEM_DEVICE) {
    // the pipeline reads its input through the galois table until its last kernel: the result is formed beside the
    // ciphertexts and copied over them (callers that can give a separate output use fhe_hip_rotate_to and save the copy)
    TRY(ensure_ws(c, 3, bytes));
    TRY(dev_rotate_to(c, galois, (u64 *)c->ws[3], (const u64 *)ct, Li, (u32)batch));
    HIPCHK(c, (prof_chain_break(c), hipMemcpyAsync(ct, c->ws[3], bytes, hipMemcpyDeviceToDevice, c->stream)));
    return FHE_HIP_OK;
  }
  Staged st(c, mem_kind);  // host memory: staged in and out through two different device buffers anyway
  const u64 *din = nullptr; u64 *dout = nullptr;
  TRY(st.in(ct, bytes, &din));
  TRY(st.out(ct, bytes, &dout, false));
  TRY(dev_rotate_to(c, galois, dout, din, Li, (u32)batch));
  return st.finish();
  } catch (...) { return abi_exception(); }
}

int fhe_hip_rotate_non_ntt(fhe_hip_ctx *c, uint64_t galois, uint64_t *ct, uint32_t Li, size_t batch, int mem_kind) {
  try {
  TRY(check_ctx(c));
  if (!ct) FAIL(c, FHE_HIP_ERR_ARG, "null argument");
  if (galois == 0) return FHE_HIP_OK;  // evaluator.cpp:331-332
  TRY(rotate_check(c, galois, Li));
  if (batch == 0) return FHE_HIP_OK;
  Staged st(c, mem_kind);
  u64 *d = nullptr;
  TRY(st.out(ct, batch * 2 * Li * c->N * 8, &d, true));
  TRY(dev_rotate_non_ntt(c, galois, d, Li, (u32)batch));
  return st.finish();
  } catch (...) { return abi_exception(); }
}

// Single-process multi-GPU mode: contiguous split of the batch over the contexts (the rule of fhe_amd/batching.py:shard_bounds:
// the first batch % n_ctx contexts take one operation more).  Every context runs its slice as a three-stage pipeline over
// chunks of at most 32 operations and three device staging slots: an uploader thread (host -> device on a stream of its own),
// the slice's own thread (the operation, enqueued on the context's stream on device pointers) and a downloader thread (device
// -> host on a third stream).  The stages hand chunks over through HIP events (device side) and a progress counter (host
// side), so the upload of chunk k+1 and the download of chunk k-1 run under the computation of chunk k, and, the host link
// being full duplex, under each other: with pageable caller memory a copy blocks only the thread that issued it.
namespace {

constexpr size_t BR_CHUNK = 32;
constexpr int BR_SLOTS = 3;
constexpr int BR_LANES = 2;   // copy threads (and streams) per direction: one thread's pageable copy does not fill the host link

struct BatchPipe {
  std::mutex mu; std::condition_variable cv;
  size_t up_done[BR_LANES] = {0, 0}, comp_done = 0, down_done[BR_LANES] = {0, 0};   // chunks that have left each stage / lane
  bool failed = false;
  int rc = FHE_HIP_OK; std::string err;
  void fail(int code, const std::string &msg) {
    std::lock_guard<std::mutex> l(mu);
    if (!failed) { failed = true; rc = code; err = msg; }
    cv.notify_all();
  }
  // wait until every given counter exceeds k (or the slice has failed): false = stop
  bool wait_for(const size_t *counters, int n, size_t k) {
    std::unique_lock<std::mutex> l(mu);
    cv.wait(l, [&] {
      if (failed) return true;
      for (int i = 0; i < n; ++i) if (counters[i] <= k) return false;
      return true;
    });
    return !failed;
  }
  void advance(size_t &counter) {
    { std::lock_guard<std::mutex> l(mu); ++counter; }
    cv.notify_all();
  }
};

int batch_ensure(fhe_hip_ctx *c, size_t in_bytes, size_t out_bytes, bool two_inputs, bool one_input) {
  if (!c->br_up[0]) {
    for (int l = 0; l < BR_LANES; ++l) {
      HIPCHK(c, hipStreamCreateWithFlags(&c->br_up[l], hipStreamNonBlocking));
      HIPCHK(c, hipStreamCreateWithFlags(&c->br_down[l], hipStreamNonBlocking));
      for (int k = 0; k < BR_SLOTS; ++k) HIPCHK(c, hipEventCreateWithFlags(&c->br_ev_up[l][k], hipEventDisableTiming));
    }
    for (int k = 0; k < BR_SLOTS; ++k) HIPCHK(c, hipEventCreateWithFlags(&c->br_ev_done[k], hipEventDisableTiming));
  }
  for (int k = 0; k < BR_SLOTS; ++k) {
    const size_t want[3] = {(two_inputs || one_input) ? in_bytes : 0, two_inputs ? in_bytes : 0, out_bytes};
    for (int j = 0; j < 3; ++j) {
      if (c->br_bytes[k][j] >= want[j]) continue;
      if (c->br_buf[k][j]) { HIPCHK(c, hipStreamSynchronize(c->stream)); HIPCHK(c, hipFree(c->br_buf[k][j])); c->br_buf[k][j] = nullptr; c->br_bytes[k][j] = 0; }
      HIPCHK(c, hipMalloc(&c->br_buf[k][j], want[j]));
      c->br_bytes[k][j] = want[j];
    }
  }
  return FHE_HIP_OK;
}

}  // namespace

int fhe_hip_batch_run(fhe_hip_ctx *const *ctxs, uint32_t n_ctx, int op, uint64_t *out, const uint64_t *in0, const uint64_t *in1,
                      uint64_t galois, uint32_t Li, size_t batch, int mem_kind) {
  try {
  if (!ctxs || n_ctx == 0) return FHE_HIP_ERR_ARG;
  for (u32 g = 0; g < n_ctx; ++g) { if (!ctxs[g]) return FHE_HIP_ERR_ARG; ctxs[g]->err.clear(); }
  fhe_hip_ctx *c0 = ctxs[0];
  if (mem_kind != FHE_HIP_MEM_HOST) FAIL(c0, FHE_HIP_ERR_ARG, "fhe_hip_batch_run takes host pointers only");
  if (op != FHE_HIP_OP_MUL_RELIN_RESCALE && op != FHE_HIP_OP_ROTATE && op != FHE_HIP_OP_ROTATE_MIXED && op != FHE_HIP_OP_ROTATE_NON_NTT) FAIL(c0, FHE_HIP_ERR_ARG, "unknown op");
  const bool mul = op == FHE_HIP_OP_MUL_RELIN_RESCALE;
  if (!out || (mul && (!in0 || !in1))) FAIL(c0, FHE_HIP_ERR_ARG, "null argument");
  for (u32 g = 1; g < n_ctx; ++g)
    if (ctxs[g]->logN != c0->logN || ctxs[g]->L != c0->L || ctxs[g]->K != c0->K) FAIL(c0, FHE_HIP_ERR_ARG, "contexts of one set must share (logN, L, K)");
  for (u32 g = 0; g < n_ctx; ++g)
    for (u32 h = g + 1; h < n_ctx; ++h)
      if (ctxs[g] == ctxs[h]) FAIL(c0, FHE_HIP_ERR_ARG, "a context may appear once in a set (a context is not thread-safe)");
  TRY(check_li(c0, Li, op == FHE_HIP_OP_ROTATE_MIXED ? 1 : 2));  // the per-op minimum of the single-context entry points
  if (batch == 0) return FHE_HIP_OK;
  const u64 N = c0->N;
  const size_t in_ct = (size_t)2 * Li * N, out_ct = (size_t)2 * (mul ? Li - 1 : Li) * N;
  std::vector<int> rc(n_ctx, FHE_HIP_OK);

  auto slice = [&](u32 g) {
    fhe_hip_ctx *c = ctxs[g];
    const size_t base = batch / n_ctx, rem = batch % n_ctx;
    const size_t lo = g * base + std::min<size_t>(g, rem), hi = lo + base + (g < rem ? 1 : 0);
    if (hi == lo) return;
    if ((rc[g] = check_ctx(c)) != FHE_HIP_OK) return;
    const size_t per_chunk = std::min(BR_CHUNK, hi - lo), nchunks = (hi - lo + BR_CHUNK - 1) / BR_CHUNK;
    const bool oop = mul || op == FHE_HIP_OP_ROTATE;  // separate input slot(s): the rotation runs out of place on the device
    if ((rc[g] = batch_ensure(c, per_chunk * in_ct * 8, per_chunk * out_ct * 8, mul, op == FHE_HIP_OP_ROTATE)) != FHE_HIP_OK) return;
    BatchPipe pipe;
    auto chunk_n = [&](size_t k) { return std::min(BR_CHUNK, hi - lo - k * BR_CHUNK); };
    auto hip_fail = [&](const char *what, hipError_t e) { pipe.fail(FHE_HIP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); };
    // lane 0 / 1 of a direction move the first / second half of a chunk's words (two inputs: in0 / in1)
    auto uploader = [&](int lane) {
      hipError_t e = hipSetDevice(c->device);
      if (e != hipSuccess) return hip_fail("hipSetDevice", e);
      for (size_t k = 0; k < nchunks; ++k) {
        if (k >= BR_SLOTS && !pipe.wait_for(pipe.down_done, BR_LANES, k - BR_SLOTS)) return;  // the slot's previous chunk has been downloaded
        const int s = (int)(k % BR_SLOTS);
        const size_t at = lo + k * BR_CHUNK, n = chunk_n(k);
        if (mul) {
          const uint64_t *src = (lane == 0 ? in0 : in1) + at * in_ct;
          if ((e = hipMemcpyAsync(c->br_buf[s][lane], src, n * in_ct * 8, hipMemcpyHostToDevice, c->br_up[lane])) != hipSuccess) return hip_fail("hipMemcpyAsync (upload)", e);
        } else {
          const size_t words = n * out_ct, half = words / 2, w0 = lane == 0 ? 0 : half, wn = lane == 0 ? half : words - half;
          uint64_t *dst = (uint64_t *)c->br_buf[s][oop ? 0 : 2];  // in-place ops are uploaded into the slot they are computed and downloaded from
          if ((e = hipMemcpyAsync(dst + w0, out + at * out_ct + w0, wn * 8, hipMemcpyHostToDevice, c->br_up[lane])) != hipSuccess) return hip_fail("hipMemcpyAsync (upload)", e);
        }
        if ((e = hipEventRecord(c->br_ev_up[lane][s], c->br_up[lane])) != hipSuccess) return hip_fail("hipEventRecord", e);
        pipe.advance(pipe.up_done[lane]);
      }
    };
    auto downloader = [&](int lane) {
      hipError_t e = hipSetDevice(c->device);
      if (e != hipSuccess) return hip_fail("hipSetDevice", e);
      for (size_t k = 0; k < nchunks; ++k) {
        if (!pipe.wait_for(&pipe.comp_done, 1, k)) return;
        const int s = (int)(k % BR_SLOTS);
        const size_t at = lo + k * BR_CHUNK, n = chunk_n(k);
        const size_t words = n * out_ct, half = words / 2, w0 = lane == 0 ? 0 : half, wn = lane == 0 ? half : words - half;
        if ((e = hipStreamWaitEvent(c->br_down[lane], c->br_ev_done[s], 0)) != hipSuccess) return hip_fail("hipStreamWaitEvent", e);
        if ((e = hipMemcpyAsync(out + at * out_ct + w0, (const uint64_t *)c->br_buf[s][2] + w0, wn * 8, hipMemcpyDeviceToHost, c->br_down[lane])) != hipSuccess) return hip_fail("hipMemcpyAsync (download)", e);
        if ((e = hipStreamSynchronize(c->br_down[lane])) != hipSuccess) return hip_fail("hipStreamSynchronize", e);
        pipe.advance(pipe.down_done[lane]);
      }
    };
    std::thread copy_t[2 * BR_LANES];
    bool all_started = true;
    try { for (int l = 0; l < BR_LANES; ++l) { copy_t[l] = std::thread(uploader, l); copy_t[BR_LANES + l] = std::thread(downloader, l); } }
    catch (...) { all_started = false; pipe.fail(FHE_HIP_ERR_HIP, "cannot start the copy threads of fhe_hip_batch_run"); }
    for (size_t k = 0; k < nchunks && all_started; ++k) {
      if (!pipe.wait_for(pipe.up_done, BR_LANES, k)) break;
      const int s = (int)(k % BR_SLOTS);
      const size_t n = chunk_n(k);
      hipError_t e = hipSuccess;
      for (int l = 0; l < BR_LANES && e == hipSuccess; ++l) e = hipStreamWaitEvent(c->stream, c->br_ev_up[l][s], 0);
      if (e != hipSuccess) { hip_fail("hipStreamWaitEvent", e); break; }
      uint64_t *d = (uint64_t *)c->br_buf[s][2];
      int r;
      if (mul) r = fhe_hip_mul_relin_rescale(c, d, (const uint64_t *)c->br_buf[s][0], (const uint64_t *)c->br_buf[s][1], Li, n, FHE_HIP_MEM_DEVICE);
      else if (op == FHE_HIP_OP_ROTATE) r = fhe_hip_rotate_to(c, galois, d, (const uint64_t *)c->br_buf[s][0], Li, n, FHE_HIP_MEM_DEVICE);
      else if (op == FHE_HIP_OP_ROTATE_MIXED) r = fhe_hip_rotate_mixed(c, galois, d, Li, n, FHE_HIP_MEM_DEVICE);
      else r = fhe_hip_rotate_non_ntt(c, galois, d, Li, n, FHE_HIP_MEM_DEVICE);
      if (r != FHE_HIP_OK) { pipe.fail(r, c->err); break; }
      if ((e = hipEventRecord(c->br_ev_done[s], c->stream)) != hipSuccess) { hip_fail("hipEventRecord", e); break; }
      pipe.advance(pipe.comp_done);
    }
    for (auto &t : copy_t) if (t.joinable()) t.join();
    for (int l = 0; l < BR_LANES; ++l) { hipStreamSynchronize(c->br_up[l]); hipStreamSynchronize(c->br_down[l]); }
    hipStreamSynchronize(c->stream);  // nothing of this call outlives it
    if (pipe.failed) { rc[g] = pipe.rc; c->err = pipe.err; }
  };
  std::vector<std::thread> workers;
  u32 started = 1;
  try {  // no exception may cross the C ABI: a slice whose thread cannot be started runs on the calling thread
    workers.reserve(n_ctx);
    for (; started < n_ctx; ++started) workers.emplace_back(slice, started);
  } catch (...) {
  }
  slice(0);
  for (u32 g = started; g < n_ctx; ++g) slice(g);
  for (auto &w : workers) w.join();
  for (u32 g = 0; g < n_ctx; ++g) if (rc[g] != FHE_HIP_OK) return rc[g];
  return FHE_HIP_OK;
  } catch (...) { return abi_exception(); }
}

int fhe_hip_load_mixed_rot_key(fhe_hip_ctx *c, uint64_t galois, const uint64_t *beta, const uint64_t *alpha, int mem_kind) {
  try {
  TRY(check_ctx(c));
  if (!beta || !alpha) FAIL(c, FHE_HIP_ERR_ARG, "null argument");
  if (c->K < 1) FAIL(c, FHE_HIP_ERR_RANGE, "mixed rotation keys need one special prime");
  if (galois != 0 && (!(galois & 1) || galois >= 2 * c->N)) FAIL(c, FHE_HIP_ERR_ARG, "galois must be odd and < 2N");
  MixedRotKeyDev &rk = c->mixed_keys[galois];
  // [L][L+1][N]: limb id = i*(L+1) + j uses modulus j = id % (L+1) (limb L is the special prime P[L])
  TRY(load_key(c, &rk.beta_m, beta, (size_t)c->L * (c->L + 1), c->d_iota, c->L + 1, mem_kind, c->relin4 != 0));
  TRY(load_key(c, &rk.alpha_m, alpha, (size_t)c->L * (c->L + 1), c->d_iota, c->L + 1, mem_kind, c->relin4 != 0));
  c->mixed_128[galois] = c->relin4 != 0;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  return FHE_HIP_OK;
  } catch (...) { return abi_exception(); }
}

int fhe_hip_rotate_mixed(fhe_hip_ctx *c, uint64_t galois, uint64_t *ct, uint32_t Li, size_t batch, int mem_kind) {
  try {
  TRY(check_ctx(c));
  if (!ct) FAIL(c, FHE_HIP_ERR_ARG, "null argument");
  if (galois == 0) return FHE_HIP_OK;  // evaluator.cpp:317-318
  if (!(galois & 1) || galois >= 2 * c->N) FAIL(c, FHE_HIP_ERR_ARG, "galois must be odd and < 2N");
  TRY(check_li(c, Li, 1));
  if (c->K < 1) FAIL(c, FHE_HIP_ERR_RANGE, "mixed rotation needs one special prime");
  if (batch == 0) return FHE_HIP_OK;
  Staged st(c, mem_kind);
  u64 *d = nullptr;
  TRY(st.out(ct, batch * 2 * Li * c->N * 8, &d, true));
  TRY(dev_rotate_mixed(c, galois, d, Li, (u32)batch));
  return st.finish();
  } catch (...) { return abi_exception(); }
}

int fhe_hip_decrypt(fhe_hip_ctx *c, uint64_t *out, const uint64_t *ct, const uint64_t *sx, uint32_t Li, size_t batch, int mem_kind) {
  try {
  TRY(check_ctx(c));
  if (!out || !ct || !sx) FAIL(c, FHE_HIP_ERR_ARG, "null argument");
  if (Li < 1 || Li > c->L) FAIL(c, FHE_HIP_ERR_RANGE, "Impossible.");  // decryptor.cpp:17-18
  if (batch == 0) return FHE_HIP_OK;
  const u64 N = c->N;
  Staged st(c, mem_kind);
  const u64 *dct = nullptr, *dsx = nullptr; u64 *dout = nullptr;
  TRY(st.in(ct, batch * 2 * Li * N * 8, &dct));
  TRY(st.in(sx, (size_t)Li * N * 8, &dsx));
  TRY(st.out(out, batch * Li * N * 8, &dout, false));
  // bx + sx * ax (decryptor.cpp:24-35) is formed while the inverse row pass loads its rows; rop->backward() (:36) follows
  NttPlan p{dct, dout, c->d_iota, 1, Li, Li, Li, (u32)batch, (u64)2 * Li * N, (u64)Li * N};
  p.pre_a = dct + (size_t)Li * N; p.pre_k = dsx;
  TRY(ntt_run(c, p, true));
  return st.finish();
  } catch (...) { return abi_exception(); }
}

int fhe_hip_encrypt(fhe_hip_ctx *c, uint64_t *ct, const uint64_t *u, const uint64_t *e0m, const uint64_t *e1, const uint64_t *pk_b,
                    const uint64_t *pk_a, uint32_t Li, size_t batch, int mem_kind) {
  try {
  TRY(check_ctx(c));
  if (!ct || !u || !e0m || !e1 || !pk_b || !pk_a) FAIL(c, FHE_HIP_ERR_ARG, "null argument");
  if (Li < 1 || Li > c->L) FAIL(c, FHE_HIP_ERR_RANGE, "bad limb count");
  if (batch == 0) return FHE_HIP_OK;
  const u64 N = c->N;
  const size_t pb = (size_t)Li * N * 8;
  // The forward NTTs read the sampled polynomials where they lie and write straight into their place ([e][half][Li][N] for
  // e0+m and e1, a workspace for u): no per-entry copies.  Host buffers (6 inputs + 1 output exceed the staging slots) are
  // first brought over as they are.
  TRY(ensure_ws(c, 0, batch * pb));
  TRY(ensure_ws(c, 4, 2 * pb));
  u64 *uh = (u64 *)c->ws[0], *dpk = (u64 *)c->ws[4];
  u64 *dct = (u64 *)ct;
  const u64 *du = (const u64 *)u, *de0 = (const u64 *)e0m, *de1 = (const u64 *)e1, *dpkb = (const u64 *)pk_b, *dpka = (const u64 *)pk_a;
  if (mem_kind != FHE_HIP_MEM_DEVICE) {
    TRY(ensure_ws(c, 3, batch * 2 * pb));
    TRY(ensure_ws(c, 5, batch * 3 * pb));
    dct = (u64 *)c->ws[3];
    u64 *in3 = (u64 *)c->ws[5];
    HIPCHK(c, (prof_chain_break(c), hipMemcpyAsync(in3, u, batch * pb, hipMemcpyHostToDevice, c->stream)));
    HIPCHK(c, (prof_chain_break(c), hipMemcpyAsync(in3 + batch * Li * N, e0m, batch * pb, hipMemcpyHostToDevice, c->stream)));
    HIPCHK(c, (prof_chain_break(c), hipMemcpyAsync(in3 + 2 * batch * Li * N, e1, batch * pb, hipMemcpyHostToDevice, c->stream)));
    HIPCHK(c, (prof_chain_break(c), hipMemcpyAsync(dpk, pk_b, pb, hipMemcpyHostToDevice, c->stream)));
    HIPCHK(c, (prof_chain_break(c), hipMemcpyAsync(dpk + (size_t)Li * N, pk_a, pb, hipMemcpyHostToDevice, c->stream)));
    du = in3; de0 = in3 + batch * Li * N; de1 = in3 + 2 * batch * Li * N; dpkb = dpk; dpka = dpk + (size_t)Li * N;
  }
  {
    NttPlan pu{du, uh, c->d_iota, 1, Li, Li, Li, (u32)batch, (u64)Li * N, (u64)Li * N};                    // u.forward(), encryptor.cpp:16
    TRY(ntt_run(c, pu, false));
    // bx/ax.forward_lazy() (:30,34) with the add_product_of steps (:31,35) as the epilogue of the row pass: the column
    // pass writes into the half's own place, the row pass reads it there and overwrites it with NTT(e) + u^ * pk
    NttPlan p0{de0, dct, c->d_iota, 1, Li, Li, Li, (u32)batch, (u64)Li * N, (u64)2 * Li * N};
    p0.epi = 3; p0.epi_src = uh; p0.epi_src_bstride = (u64)Li * N; p0.epi_add = dpkb; p0.epi_add_bstride = 0;
    p0.epi_out = dct; p0.epi_out_bstride = (u64)2 * Li * N;
    TRY(ntt_run(c, p0, false));
    NttPlan p1{de1, dct + (size_t)Li * N, c->d_iota, 1, Li, Li, Li, (u32)batch, (u64)Li * N, (u64)2 * Li * N};
    p1.epi = 3; p1.epi_src = uh; p1.epi_src_bstride = (u64)Li * N; p1.epi_add = dpka; p1.epi_add_bstride = 0;
    p1.epi_out = dct + (size_t)Li * N; p1.epi_out_bstride = (u64)2 * Li * N;
    TRY(ntt_run(c, p1, false));
  }
  if (mem_kind != FHE_HIP_MEM_DEVICE) {
    HIPCHK(c, (prof_chain_break(c), hipMemcpyAsync(ct, dct, batch * 2 * pb, hipMemcpyDeviceToHost, c->stream)));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return FHE_HIP_OK;
  } catch (...) { return abi_exception(); }
}

}  // extern "C"

// ---- Encoder: reference include/fHE/encoder.hpp:31-45, src/encoder.cpp ---------------------------------
namespace {
int ensure_encoder_tables(fhe_hip_ctx *c) {
  if (c->d_enc_roots) return FHE_HIP_OK;
  const u64 N = c->N; const u32 logN = c->logN, L = c->L;
  std::vector<cplx> roots(N);
  std::vector<u32> map(N);
  const long double two_pi = 6.283185307179586476925286766559005768L;
  for (u64 i = 0; i < N; ++i) {  // roots_[i] = psi^brv(i), psi = exp(2 pi i / 2N)  (encoder.cpp:27-33)
    const long double ang = two_pi * (long double)h_brv(i, logN) / (long double)(2 * N);
    roots[i] = make_double2((double)cosl(ang), (double)sinl(ang));
  }
  u64 pos = 1;
  for (u64 i = 0; i < N / 2; ++i) {  // encoder.cpp:9-25
    map[i] = (u32)h_brv((pos - 1) >> 1, logN);
    map[N / 2 + i] = (u32)h_brv((2 * N - pos - 1) >> 1, logN);
    pos = (pos * 3) & (2 * N - 1);
  }
  std::vector<u64> prodm((size_t)L * L, 0), ginv(L, 0);   // Garner constants: (q_0 .. q_{j-1} mod q_k) * 2^128 at [k][j], (q_0 .. q_{k-1})^-1 mod q_k
  std::vector<double> qd(L);
  const std::vector<u64> &P = c->primes;
  for (u32 k = 0; k < L; ++k) {
    qd[k] = (double)P[k];
    u64 prod = 1;
    for (u32 j = 0; j < k; ++j) { prodm[(size_t)k * L + j] = h_to_mont(h_to_mont(prod, P[k]), P[k]); prod = h_mulmod(prod, P[j] % P[k], P[k]); }
    if (k) ginv[k] = h_to_mont(h_invmod(prod, P[k]), P[k]);
  }
  int rc;
  if ((rc = upload(c, &c->d_enc_map, map)) || (rc = upload(c, &c->d_enc_prod_m, prodm)) || (rc = upload(c, &c->d_enc_ginv_m, ginv)) ||
      (rc = upload(c, &c->d_enc_qd, qd)) || (rc = upload(c, &c->d_enc_roots, roots))) return rc;
  return FHE_HIP_OK;
}
// mixed-radix digits of floor(Q_n / 2), Q_n = q_0 .. q_{n-1}: exact multi-word arithmetic on the host, once per limb count
int ensure_half_digits(fhe_hip_ctx *c, u32 n, const u64 **out) {
  auto it = c->enc_half.find(n);
  if (it != c->enc_half.end()) { *out = it->second; return FHE_HIP_OK; }
  std::vector<u64> big{1};
  for (u32 k = 0; k < n; ++k) {  // big *= q_k
    u128 carry = 0;
    for (auto &w : big) { carry += (u128)w * c->primes[k]; w = (u64)carry; carry >>= 64; }
    if (carry) big.push_back((u64)carry);
  }
  { u64 carry = 0; for (size_t i = big.size(); i-- > 0;) { const u64 w = big[i]; big[i] = (w >> 1) | (carry << 63); carry = w & 1; } }
  std::vector<u64> digits(n);
  for (u32 k = 0; k < n; ++k) {  // digit = big mod q_k, big /= q_k
    u128 rem = 0;
    for (size_t i = big.size(); i-- > 0;) { const u128 cur = (rem << 64) | big[i]; big[i] = (u64)(cur / c->primes[k]); rem = cur % c->primes[k]; }
    digits[k] = (u64)rem;
  }
  u64 *d = nullptr;
  TRY(upload(c, &d, digits));
  c->enc_half[n] = d;
  *out = d;
  return FHE_HIP_OK;
}
template <bool INV> int enc_fft(fhe_hip_ctx *c, cplx *w, u32 batch) {
  const u32 R = 1u << c->logR;
  const size_t smem = (size_t)R * ENC_CT * sizeof(cplx);
  if constexpr (INV) {
    LAUNCH(c, "enc_ifft_rows", (enc_fft_rows_kernel<true>), dim3(R / 2, batch), dim3(256), w, (const cplx *)c->d_enc_roots, c->logN);
    LAUNCH_SMEM(c, "enc_ifft_cols", (enc_fft_cols_kernel<true>), dim3(256 / ENC_CT, batch), dim3(256), smem, w, (const cplx *)c->d_enc_roots, c->logN);
  } else {
    LAUNCH_SMEM(c, "enc_fft_cols", (enc_fft_cols_kernel<false>), dim3(256 / ENC_CT, batch), dim3(256), smem, w, (const cplx *)c->d_enc_roots, c->logN);
    LAUNCH(c, "enc_fft_rows", (enc_fft_rows_kernel<false>), dim3(R / 2, batch), dim3(256), w, (const cplx *)c->d_enc_roots, c->logN);
  }
  return FHE_HIP_OK;
}
}  // namespace

extern "C" {

int fhe_hip_encode(fhe_hip_ctx *c, uint64_t *msg, const double *values, size_t nvals, uint32_t nlimbs, double scale, size_t batch, int mem_kind) {
  try {
  TRY(check_ctx(c));
  if (!msg || (!values && nvals)) FAIL(c, FHE_HIP_ERR_ARG, "null argument");
  if (nlimbs < 1 || nlimbs > c->L) FAIL(c, FHE_HIP_ERR_RANGE, "bad limb count");
  if (!(scale > 0.0) || !std::isfinite(scale)) FAIL(c, FHE_HIP_ERR_ARG, "scale must be positive and finite");
  if (batch == 0) return FHE_HIP_OK;
  if (batch > 65535) FAIL(c, FHE_HIP_ERR_RANGE, "batch too large for one call");
  const u64 N = c->N;
  const size_t used = nvals < N / 2 ? nvals : N / 2;  // "Just handle the first nr_slots values" (encoder.cpp:87-91)
  TRY(ensure_encoder_tables(c));
  TRY(ensure_ws(c, 5, batch * N * sizeof(cplx)));
  cplx *w = (cplx *)c->ws[5];
  Staged st(c, mem_kind);
  const u64 *dvals = nullptr; u64 *dmsg = nullptr;
  if (nvals) TRY(st.in(values, batch * nvals * sizeof(cplx), &dvals));
  TRY(st.out(msg, batch * nlimbs * N * 8, &dmsg, false));
  HIPCHK(c, (prof_chain_break(c), hipMemsetAsync(w, 0, batch * N * sizeof(cplx), c->stream)));
  if (used) LAUNCH(c, "enc_scatter", enc_scatter_kernel, dim3((unsigned)((used + 255) / 256), (unsigned)batch), dim3(256), w, (const cplx *)dvals,
                   (const u32 *)c->d_enc_map, N, (u64)nvals);
  TRY((enc_fft<true>(c, w, (u32)batch)));
  LAUNCH(c, "enc_round", enc_round_kernel, dim3((unsigned)(N / 256), (unsigned)batch), dim3(256), dmsg, (const cplx *)w, (const PrimeC *)c->d_primes, nlimbs, N,
         scale / (double)N);
  return st.finish();
  } catch (...) { return abi_exception(); }
}

int fhe_hip_decode(fhe_hip_ctx *c, double *slots, const uint64_t *msg, uint32_t nlimbs, double scale, size_t batch, int mem_kind) {
  try {
  TRY(check_ctx(c));
  if (!slots || !msg) FAIL(c, FHE_HIP_ERR_ARG, "null argument");
  if (nlimbs < 1 || nlimbs > c->L || nlimbs > DEC_MAX_LIMBS) FAIL(c, FHE_HIP_ERR_RANGE, "bad limb count");
  if (!(scale > 0.0) || !std::isfinite(scale)) FAIL(c, FHE_HIP_ERR_ARG, "scale must be positive and finite");
  if (batch == 0) return FHE_HIP_OK;
  if (batch > 65535) FAIL(c, FHE_HIP_ERR_RANGE, "batch too large for one call");
  const u64 N = c->N;
  TRY(ensure_encoder_tables(c));
  const u64 *half = nullptr;
  TRY(ensure_half_digits(c, nlimbs, &half));
  TRY(ensure_ws(c, 5, batch * N * sizeof(cplx)));
  cplx *w = (cplx *)c->ws[5];
  Staged st(c, mem_kind);
  const u64 *dmsg = nullptr; u64 *dslots = nullptr;
  TRY(st.in(msg, batch * nlimbs * N * 8, &dmsg));
  TRY(st.out(slots, batch * (N / 2) * sizeof(cplx), &dslots, false));
  DecLiftArgs a;
  a.w = w; a.msg = dmsg; a.primes = c->d_primes; a.prod_m = c->d_enc_prod_m; a.ginv_m = c->d_enc_ginv_m; a.half = half; a.qd = c->d_enc_qd;
  a.n = nlimbs; a.L = c->L; a.N = N; a.inv_scale = 1.0 / scale;
  if (nlimbs <= DEC_REG_LIMBS) {
    LAUNCH_SP(c, "dec_lift", dec_lift_reg_kernel, dim3((unsigned)(N / 256), (unsigned)batch), dim3(256), a);
  } else {
    const size_t smem = (size_t)nlimbs * nlimbs * 8 + (size_t)3 * nlimbs * DEC_THREADS * sizeof(u32);   // Garner constants + the digits' 21-bit limbs (128 KB at 64 limbs)
    if (!c->dec_attr_set) {
      HIPCHK(c, hipFuncSetAttribute((const void *)dec_lift_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      HIPCHK(c, hipFuncSetAttribute((const void *)dec_lift_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      c->dec_attr_set = true;
    }
    if (c->sp) LAUNCH_SMEM(c, "dec_lift", (dec_lift_kernel<true>), dim3((unsigned)(N / DEC_THREADS), (unsigned)batch), dim3(DEC_THREADS), smem, a);
    else LAUNCH_SMEM(c, "dec_lift", (dec_lift_kernel<false>), dim3((unsigned)(N / DEC_THREADS), (unsigned)batch), dim3(DEC_THREADS), smem, a);
  }
  TRY((enc_fft<false>(c, w, (u32)batch)));
  LAUNCH(c, "dec_pick", dec_pick_kernel, dim3((unsigned)((N / 2 + 255) / 256), (unsigned)batch), dim3(256), (cplx *)dslots, (const cplx *)w, (const u32 *)c->d_enc_map, N);
  return st.finish();
  } catch (...) { return abi_exception(); }
}

int fhe_hip_debug_violations(fhe_hip_ctx *c, uint64_t *count, uint32_t *first_site, int *enabled) {
  TRY(check_ctx(c));
  if (count) *count = 0;
  if (first_site) *first_site = 0;
#ifdef FHE_HIP_DEBUG
  if (enabled) *enabled = 1;
  HIPCHK(c, hipStreamSynchronize(c->stream));
  unsigned long long n = 0; unsigned int site = 0;
  HIPCHK(c, hipMemcpyFromSymbol(&n, HIP_SYMBOL(fhe_hip_dbg_count), sizeof(n)));
  HIPCHK(c, hipMemcpyFromSymbol(&site, HIP_SYMBOL(fhe_hip_dbg_first_site), sizeof(site)));
  if (count) *count = n;
  if (first_site) *first_site = site;
  n = 0; site = 0;  // reading resets
  HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(fhe_hip_dbg_count), &n, sizeof(n)));
  HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(fhe_hip_dbg_first_site), &site, sizeof(site)));
#else
  if (enabled) *enabled = 0;
#endif
  return FHE_HIP_OK;
}

int fhe_hip_profile_enable(fhe_hip_ctx *c, int on) { TRY(check_ctx(c)); c->prof = on != 0; return FHE_HIP_OK; }
int fhe_hip_profile_reset(fhe_hip_ctx *c) {
  try {
  TRY(check_ctx(c));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  prof_drain(c);
  for (auto &e : c->prof_entries) { e.launches = 0; e.total_ms = 0; }
  return FHE_HIP_OK;
  } catch (...) { return abi_exception(); }
}
int fhe_hip_profile_get(fhe_hip_ctx *c, const char **names, uint64_t *launches, double *total_ms, int cap) {
  try {
  if (!c) return FHE_HIP_ERR_ARG;
  hipSetDevice(c->device);
  hipStreamSynchronize(c->stream);
  prof_drain(c);
  int n = 0;
  for (auto &e : c->prof_entries) {
    if (n < cap) { if (names) names[n] = e.name.c_str(); if (launches) launches[n] = e.launches; if (total_ms) total_ms[n] = e.total_ms; }
    ++n;
  }
  return n;
  } catch (...) { return abi_exception(); }
}

}  // extern "C"

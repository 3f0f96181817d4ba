/*
 * fhe_hip.h — C-ABI of the MI355X (gfx950) backend for fHE's RNS-CKKS polynomial hot path.
 *
 * The reference (fionser/fHE) has no plugin/FFI interface; its seams are the pimpl classes
 * fHE::Evaluator (include/fHE/evaluator.hpp:13-15, Impl src/evaluator.cpp:12-68) and
 * fHE::BaseConverter (include/fHE/base_converter.hpp:5-9, Impl src/base_converter.cpp:29-78) and the
 * raw-pointer calls yell::ntt<N>::forward/backward(T*, size_t moduli_idx).  Each entry point below
 * cites the reference interface it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; 0 = OK, negative = error (fhe_hip_last_error()).
 *   - limb word = uint64_t, little endian; polynomial = limb-major [nlimbs][N] contiguous
 *     (yell::poly layout, src/base_converter.cpp:199); ciphertext batch = [ct][half][limb][N] with
 *     half 0 = bx, half 1 = ax (include/fHE/cipher.hpp:9-10).
 *   - moduli are addressed by GLOBAL index into the context's prime table: normal primes 0..L-1,
 *     special primes L..L+K-1 (include/fHE/context.hpp:16-19).
 *   - mem_kind: FHE_HIP_MEM_DEVICE pointers are device memory of the context's GPU (no copies);
 *     FHE_HIP_MEM_HOST pointers are host memory, staged through device buffers by the library.
 *   - caller owns every buffer; one context per device; a context is not thread-safe; work is
 *     enqueued on the context's stream (fhe_hip_set_stream) and host-memory calls synchronise.
 *   - NTT-domain values are canonical residues in [0,p) on output; forward-NTT inputs may be any
 *     value < 4p (residues of another prime of the table: src/evaluator.cpp:667-668); backward-NTT
 *     and pointwise inputs must be < 2p.
 */
#ifndef FHE_HIP_H
#define FHE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fhe_hip_ctx fhe_hip_ctx;

enum { FHE_HIP_MEM_DEVICE = 0, FHE_HIP_MEM_HOST = 1 };

enum {
  FHE_HIP_OK = 0,
  FHE_HIP_ERR_ARG = -1,      /* null pointer / shape mismatch (reference: return false + std::cerr) */
  FHE_HIP_ERR_RANGE = -2,    /* limb count out of range (reference: std::invalid_argument) */
  FHE_HIP_ERR_NOKEY = -3,    /* key not loaded */
  FHE_HIP_ERR_HIP = -4,      /* HIP runtime error */
  FHE_HIP_ERR_NODEVICE = -5  /* no usable GPU: the library never falls back to the CPU */
};

/* fHE::context (include/fHE/context.hpp:6-25) made runtime: N = 2^logN, L normal + K special primes,
 * primes[L+K] (< 2^62, 2N | p-1) and psi[L+K] (primitive 2N-th roots) as in yell::params::P. */
int fhe_hip_ctx_create(fhe_hip_ctx **out, uint32_t logN, uint32_t L, uint32_t K, const uint64_t *primes,
                       const uint64_t *psi, int device);
void fhe_hip_ctx_destroy(fhe_hip_ctx *ctx);
const char *fhe_hip_last_error(const fhe_hip_ctx *ctx); /* ctx may be NULL: last create() error */
int fhe_hip_set_stream(fhe_hip_ctx *ctx, void *hip_stream);
int fhe_hip_sync(fhe_hip_ctx *ctx);

/* yell::ntt<N>::forward / backward(T* limb, size_t moduli_idx) (call sites src/evaluator.cpp:365,390,
 * 668,707; src/base_converter.cpp:224,260), batched: limbs = [n_limbs][N] in place, moduli_idx[n_limbs]
 * is a HOST array.  Natural order in, bit-reversed out at psi^(2*brv(d)+1) (src/yell.cpp:16-21);
 * backward includes N^-1. */
int fhe_hip_ntt_forward(fhe_hip_ctx *ctx, uint64_t *limbs, const uint32_t *moduli_idx, size_t n_limbs, int mem_kind);
int fhe_hip_ntt_backward(fhe_hip_ctx *ctx, uint64_t *limbs, const uint32_t *moduli_idx, size_t n_limbs, int mem_kind);

/* fHE::apply_galois / apply_galois_non_ntt (src/yell.cpp:4-25, 29-53): polys = [batch][nlimbs][N] in
 * place, limb k of a polynomial uses modulus k. */
int fhe_hip_apply_galois(fhe_hip_ctx *ctx, uint64_t *polys, uint32_t nlimbs, uint64_t galois, size_t batch, int mem_kind);
int fhe_hip_apply_galois_non_ntt(fhe_hip_ctx *ctx, uint64_t *polys, uint32_t nlimbs, uint64_t galois, size_t batch, int mem_kind);

/* BaseConverter::approximated_mod_up (src/base_converter.cpp:189-203): op [batch][Li][N] power basis
 * -> rop [batch][Li+K][N].  Li < 2 -> FHE_HIP_ERR_RANGE (:284-285). */
int fhe_hip_mod_up(fhe_hip_ctx *ctx, uint64_t *rop, const uint64_t *op, uint32_t Li, size_t batch, int mem_kind);
/* BaseConverter::approximated_mod_down (:205-242): op [batch][L0+K][N] (normal limbs NTT basis, special
 * limbs power basis) -> rop [batch][L0][N] NTT basis. */
int fhe_hip_mod_down(fhe_hip_ctx *ctx, uint64_t *rop, const uint64_t *op, uint32_t L0, size_t batch, int mem_kind);
/* BaseConverter::rescale (:244-274): op [batch][n][N] (last limb power basis, rest NTT basis) ->
 * rop [batch][n-1][N] NTT basis; floor-style, no rounding term. */
int fhe_hip_rescale(fhe_hip_ctx *ctx, uint64_t *rop, const uint64_t *op, uint32_t n, size_t batch, int mem_kind);
/* BaseConverter::neg_approx_convert_to_normal_basis for the targets j = 0..L0-1 (src/base_converter.cpp:297-311 over
 * :313-349): sp = [batch][K][N], the K special limbs in the power basis -> rop = [batch][L0][N] with
 * rop_j = sum_k [sp_k * hat_pi_invmod_pi_k]_{p_k} * neg_hat_pi_mod_qj[j][k] mod q_j, power basis (the conversion inside
 * approximated_mod_down before its NTT, :220-223).  approx_convert_to_special_basis (:276-295) is the special-limb part
 * of fhe_hip_mod_up. */
int fhe_hip_conv_to_normal(fhe_hip_ctx *ctx, uint64_t *rop, const uint64_t *sp, uint32_t L0, size_t batch, int mem_kind);
/* BaseConverter::approx_convert_to_special_basis (include/fHE/base_converter.hpp:30-32, src/base_converter.cpp:276-295) for ONE
 * special prime: op = [batch][Li][N] power basis -> rop = [batch][N], the residues under P[L + sp_index] (uncorrected, as
 * fhe_hip_mod_up computes for all K special primes at once).  Li <= 1 -> FHE_HIP_ERR_RANGE (:284-285). */
int fhe_hip_conv_to_special(fhe_hip_ctx *ctx, uint64_t *rop, const uint64_t *op, uint32_t Li, uint32_t sp_index, size_t batch, int mem_kind);

/* Device memory for host-language callers that keep polynomials on the GPU between calls: the device side of the storage
 * behind yell::poly / fHE::Cipher (reference include/fHE/cipher.hpp:5-50; the reference, a CPU library,
 * has host vectors only).  Blocks are plain device memory of the context's GPU, usable as FHE_HIP_MEM_DEVICE operands.
 * fhe_hip_mem_copy is ordered on the context's stream; it returns once a host buffer involved is the caller's again
 * (host <-> device: after completion; device -> device: enqueued).  fhe_hip_mem_free waits for the context's stream first. */
#define FHE_HIP_COPY_H2D 1
#define FHE_HIP_COPY_D2H 2
#define FHE_HIP_COPY_D2D 3
int fhe_hip_mem_alloc(fhe_hip_ctx *ctx, uint64_t **dptr, size_t words);
int fhe_hip_mem_free(fhe_hip_ctx *ctx, uint64_t *dptr);
int fhe_hip_mem_copy(fhe_hip_ctx *ctx, void *dst, const void *src, size_t bytes, int direction);

/* MultKey (include/fHE/mult_key.hpp:5-38): beta/alpha = [L][L][N] (key j, limb i), NTT basis.
 * RotKey (include/fHE/rot_key.hpp:9-39): beta/alpha = [L+K][N], NTT basis, selected later by galois.
 * Keys are copied to (and re-encoded on) the device; the caller's buffers are not retained. */
int fhe_hip_load_mult_key(fhe_hip_ctx *ctx, const uint64_t *beta, const uint64_t *alpha, int mem_kind);
int fhe_hip_load_rot_key(fhe_hip_ctx *ctx, uint64_t galois, const uint64_t *beta, const uint64_t *alpha, int mem_kind);

/* Evaluator::add (src/evaluator.cpp:270-297, poly operator+=): rop, op = [batch][nlimbs][N]. */
int fhe_hip_add(fhe_hip_ctx *ctx, uint64_t *rop, const uint64_t *op, uint32_t nlimbs, size_t batch, int mem_kind);
/* Evaluator::multiply_plain (:88-111): ct [batch][2][nlimbs][N] *= pt [nlimbs][N] on both halves. */
int fhe_hip_multiply_plain(fhe_hip_ctx *ctx, uint64_t *ct, const uint64_t *pt, uint32_t nlimbs, size_t batch, int mem_kind);
/* Evaluator::multiply = Mul::cross_mult (:621-647): c0, c1 = [batch][2][Li][N]; out3 = [batch][3][Li][N]
 * holding (bx, ax, mul_aux). */
int fhe_hip_cross_mult(fhe_hip_ctx *ctx, uint64_t *out3, const uint64_t *c0, const uint64_t *c1, uint32_t Li, size_t batch, int mem_kind);
/* Mul::reLinearize only (:649-693): ct3 = [batch][3][Li][N] -> out = [batch][2][Li][N] (no rescale). */
int fhe_hip_relin_only(fhe_hip_ctx *ctx, uint64_t *out, const uint64_t *ct3, uint32_t Li, size_t batch, int mem_kind);
/* Evaluator::relinearize = Impl::relinerize_inplace (:599-619): reLinearize + rescale_ntt;
 * ct3 = [batch][3][Li][N] -> out = [batch][2][Li-1][N]. */
int fhe_hip_relinearize(fhe_hip_ctx *ctx, uint64_t *out, const uint64_t *ct3, uint32_t Li, size_t batch, int mem_kind);
/* Evaluator::multiply_and_relin (:119-126): c0, c1 = [batch][2][Li][N] -> out = [batch][2][Li-1][N]. */
int fhe_hip_mul_relin_rescale(fhe_hip_ctx *ctx, uint64_t *out, const uint64_t *c0, const uint64_t *c1, uint32_t Li, size_t batch, int mem_kind);
/* Evaluator::rotate_slots(Cipher*, RotKey const&) (:128-136, 299-311, 343-399): ct = [batch][2][Li][N]
 * in place with the RotKey loaded for `galois`; galois == 0 is a no-op (:303-304). */
int fhe_hip_rotate(fhe_hip_ctx *ctx, uint64_t galois, uint64_t *ct, uint32_t Li, size_t batch, int mem_kind);
/* The same rotation out of place: in, out = [batch][2][Li][N], out != in, in is left untouched (the reference's method works on
 * its argument; a batch on the device usually wants a separate result).  The device pipeline is out of place by nature (it
 * reads the input through the galois index table until its last kernel), so this form saves the copy fhe_hip_rotate makes for
 * device buffers.  galois == 0 copies in to out. */
int fhe_hip_rotate_to(fhe_hip_ctx *ctx, uint64_t galois, uint64_t *out, const uint64_t *in, uint32_t Li, size_t batch, int mem_kind);

/* Evaluator::rotate_slots_non_ntt(Cipher*, RotKey const&) (src/evaluator.cpp:138-146, 327-339, 520-574): ct = [batch][2][Li][N]
 * in the POWER basis, in place; apply_galois_non_ntt on both halves, then the hybrid key switch with every limb taken
 * through NTT -> key product -> INTT and approximated_mod_down applied to the result.  The reference's mod_down adds an
 * NTT-domain correction to those power-basis limbs (base_converter.cpp:220-239 under evaluator.cpp:563-572); this entry
 * point reproduces that arithmetic literally, so its output equals the reference's, not a valid rotation.
 *  */
int fhe_hip_rotate_non_ntt(fhe_hip_ctx *ctx, uint64_t galois, uint64_t *ct, uint32_t Li, size_t batch, int mem_kind);

/* MixedRotKey (include/fHE/rot_key.hpp:41-82, src/rot_key.cpp:81-120): beta/alpha = [L][L+1][N] (key i, limb j; limb L
 * lives under the special prime P[L]), NTT basis.   */
int fhe_hip_load_mixed_rot_key(fhe_hip_ctx *ctx, uint64_t galois, const uint64_t *beta, const uint64_t *alpha, int mem_kind);
/* Evaluator::rotate_slots(Cipher*, MixedRotKey const&) (src/evaluator.cpp:148-156, 313-325, 403-516): digit-decomposed
 * key switch over Li+1 limbs with the fused rescale by P[L]; ct = [batch][2][Li][N] in place. */
int fhe_hip_rotate_mixed(fhe_hip_ctx *ctx, uint64_t galois, uint64_t *ct, uint32_t Li, size_t batch, int mem_kind);

/* Decryptor::decrypt(poly_t*, Cipher const&, SK const&) inner loops (src/decryptor.cpp:24-36): ct = [batch][2][Li][N]
 * canonical NTT basis, sx = [>=Li][N] secret key (NTT basis) -> out = [batch][Li][N] = INTT(bx + sx*ax), power basis.
 * (the step right after the path.) */
int fhe_hip_decrypt(fhe_hip_ctx *ctx, uint64_t *out, const uint64_t *ct, const uint64_t *sx, uint32_t Li, size_t batch, int mem_kind);
/* Encryptor::encrypt inner loops (src/encryptor.cpp:15-35) with the sampled polynomials supplied by the caller (the
 * samplers stay on the host): u, e0m (= e0 + msg), e1 = [batch][Li][N] power basis; pk_b, pk_a = [>=Li][N] NTT basis
 * -> ct = [batch][2][Li][N] with bx = NTT(e0m) + NTT(u)*pk_b, ax = NTT(e1) + NTT(u)*pk_a.  (§8(f) rank 3.) */
int fhe_hip_encrypt(fhe_hip_ctx *ctx, uint64_t *ct, const uint64_t *u, const uint64_t *e0m, const uint64_t *e1,
                    const uint64_t *pk_b, const uint64_t *pk_a, uint32_t Li, size_t batch, int mem_kind);

/* Single-process multi-GPU mode ("one ctx + stream set per device", batch index split contiguously).  No
 * counterpart in the reference, which is single-threaded on one CPU; the split is the rule of fhe_amd/batching.py
 * (shard_bounds: context g gets batch / n_ctx operations, the first batch % n_ctx contexts one more).  ctxs = n_ctx contexts,
 * normally one per GPU, created with the same (logN, L, K, primes, psi) and each already holding the keys the op needs.
 * Every context runs its slice on a host thread of its own as a pipeline over chunks of at most 32 operations and three
 * device staging slots: upload of chunk k+1 and download of chunk k-1 (two more threads and streams per context) run under
 * the computation of chunk k.  Host pointers only (mem_kind must be FHE_HIP_MEM_HOST; pageable or pinned: pinned memory
 * doubles the copy rate); device-resident callers drive the per-context entry points from their own threads.  Independent
 * ciphertexts, no collective.  Returns the first non-zero status of any slice; fhe_hip_last_error(ctxs[g]) is empty for the
 * contexts whose slice succeeded and carries the message for the ones that failed.
 *   FHE_HIP_OP_MUL_RELIN_RESCALE: out = [batch][2][Li-1][N], in0, in1 = [batch][2][Li][N]   (fhe_hip_mul_relin_rescale)
 *   FHE_HIP_OP_ROTATE / _ROTATE_MIXED / _ROTATE_NON_NTT: out = [batch][2][Li][N] in place, in0 = in1 = NULL, galois
 *                                 (fhe_hip_rotate, fhe_hip_rotate_mixed, fhe_hip_rotate_non_ntt; Li >= 1 for _ROTATE_MIXED) */
#define FHE_HIP_OP_MUL_RELIN_RESCALE 1
#define FHE_HIP_OP_ROTATE 2
#define FHE_HIP_OP_ROTATE_MIXED 3
#define FHE_HIP_OP_ROTATE_NON_NTT 4
int fhe_hip_batch_run(fhe_hip_ctx *const *ctxs, uint32_t n_ctx, int op, uint64_t *out, const uint64_t *in0, const uint64_t *in1,
                      uint64_t galois, uint32_t Li, size_t batch, int mem_kind);

/* fhe_hip_mul_relin_rescale for callers that cannot allocate.  No counterpart in the reference (a CPU library with no device
 * memory); it exists for callers that cannot call hipMalloc: code under HIP graph capture, tight serving loops, harnesses
 * that forbid allocation.  Same kernels and results as fhe_hip_mul_relin_rescale, device pointers only; every workspace is
 * carved out of a static device array of FHE_HIP_NOALLOC_SCRATCH_BYTES that each GPU the library runs on holds until the
 * process ends.  The call performs no hipMalloc / hipFree / event creation.  Needs batch * (9 Li - 2) * N * 8 bytes of scratch
 * (N = 2^16, Li = 30: 1 op per call; N = 2^15, Li = 16: 7 ops); larger requests fail with FHE_HIP_ERR_RANGE and no side
 * effect.  Every call on a GPU, from any context, uses that GPU's one array: calls that overlap in time on one GPU race;
 * serialise them (one stream, or events between streams). */
#define FHE_HIP_NOALLOC_SCRATCH_BYTES (256u << 20)
int fhe_hip_mul_relin_rescale_noalloc(fhe_hip_ctx *ctx, uint64_t *out, const uint64_t *c0, const uint64_t *c1, uint32_t Li, size_t batch);

/* Encoder::encode (include/fHE/encoder.hpp:31-37, src/encoder.cpp:84-157): values = [batch][nvals] complex doubles (re, im
 * interleaved; real inputs have im = 0), the first min(nvals, N/2) of each entry are used like the reference's ("Just handle the
 * first nr_slots values") -> msg = [batch][nlimbs][N], power basis, limb l under P[l]:
 * coefficient = trunc(Re(iFFT(slots through the generator-3 index map))[d] * scale / N) mod P[l], non-negative remainder
 * (mpz_set_d + mpz_mod_ui).  Floating point (complex double, like the reference): results agree with a CPU run to rounding.
 * Encoder::decode (encoder.hpp:39-45, encoder.cpp:159-200): msg = [batch][nlimbs][N] power basis -> slots = [batch][N/2] complex
 * doubles: CRT lift to [0, Q), centred around Q/2, divided by scale, FFT, slots picked through the index map.  Residues are expected
 * canonical (any value below 4p costs nothing extra; larger words are reduced exactly on a slow path). */
int fhe_hip_encode(fhe_hip_ctx *ctx, uint64_t *msg, const double *values, size_t nvals, uint32_t nlimbs, double scale, size_t batch, int mem_kind);
int fhe_hip_decode(fhe_hip_ctx *ctx, double *slots, const uint64_t *msg, uint32_t nlimbs, double scale, size_t batch, int mem_kind);

/* Runtime tuning (the reference has compile-time constants only, include/fHE/context.hpp:7-14).
 *   FHE_HIP_TUNE_RELIN_GROUP: output limbs per relinearisation pass (0 = all at once, the default).  The column-pass output
 *   of the Li^2 digit NTTs is produced and consumed value limbs at a time, which bounds its footprint to
 *   batch * value * Li * N * 8 bytes; results do not depend on it. */
#define FHE_HIP_TUNE_RELIN_GROUP 1
/*   FHE_HIP_TUNE_NTT_CHUNK: limbs per column-pass / row-pass launch pair of fhe_hip_ntt_forward / _backward (0 = the whole call,
 *   the default): with a chunk whose limbs fit the 256 MiB Infinity Cache the second pass reads what the first just wrote. */
#define FHE_HIP_TUNE_NTT_CHUNK 2
#define FHE_HIP_TUNE_NTT_STREAMS 4 /* 2 (default): fhe_hip_ntt_forward / _backward at N >= 2^15 with at least two chunks of limbs (chunk =
                                    * FHE_HIP_TUNE_NTT_CHUNK, default 128) issue the chunks alternately on two library-owned streams forked from
                                    * and joined to the context's stream, so one chunk's column pass overlaps another's row pass; 1: one stream */
#define FHE_HIP_TUNE_RELIN_KERNEL 5 /* 1 (default): digit key switch (relinearisation, MixedRotKey rotation) with unreduced key-MAC accumulators;
                                    * 0: the Montgomery-MAC kernel of rounds 1-2 (A/B).  Set BEFORE loading the keys it applies to: the two kernels
                                    * keep the key copies in different forms, and a key loaded for the other one is refused (FHE_HIP_ERR_NOKEY). */
#define FHE_HIP_TUNE_BCONV 6        /* base conversions whose digits arrive ready-made (the rotation pipelines).  0 (default): on the matrix cores
                                    * (byte-decomposed integer product on v_mfma_i32_16x16x64_i8) when the conversion has <= 32 source limbs, its
                                    * byte image fits one CU's LDS and the launch fills persistent workgroups, else on the vector ALU;
                                    * 1: vector-ALU kernel only; 2: matrix cores whenever the shape allows.  Identical results. */
#define FHE_HIP_TUNE_ONEPASS 3     /* 1 (default): plain NTTs at N = 2^12 .. 2^14 run as one kernel with the limb in LDS; 0: pass pair */
int fhe_hip_set_tuning(fhe_hip_ctx *ctx, int knob, uint64_t value);

/* Debug build support (reference analogue: the non-release -DCHECK_STRICTMOD range asserts, CMakeCompilers.txt:26).
 * libfhe_hip_debug.so (make -C fhe_amd/csrc DEBUG=1) compiles residue-range checks into the kernels: < p on canonical
 * outputs, < 4p / < 2p on lazy hand-offs and inputs.  Violations are counted on the device, never trapped.  This call
 * returns how many were seen since the last call and the site of the first one (list in fhe_amd/csrc/modarith.h), and
 * resets both; *enabled = 0 and a zero count in the ordinary build. */
int fhe_hip_debug_violations(fhe_hip_ctx *ctx, uint64_t *count, uint32_t *first_site, int *enabled);

/* Measurement support (no reference counterpart): when enabled every kernel launch is bracketed by
 * HIP events on the context's stream; stats are summed per kernel name. */
int fhe_hip_profile_enable(fhe_hip_ctx *ctx, int on);
int fhe_hip_profile_reset(fhe_hip_ctx *ctx);
/* returns number of kernels; fills up to cap entries (name pointers stay valid for the ctx lifetime) */
int fhe_hip_profile_get(fhe_hip_ctx *ctx, const char **names, uint64_t *launches, double *total_ms, int cap);

#ifdef __cplusplus
}
#endif
#endif

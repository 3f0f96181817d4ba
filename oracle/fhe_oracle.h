/*
 * fhe_oracle.h — CPU restatement (plain C) of fHE's RNS-CKKS polynomial hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (fhe_amd/, the C-ABI library) may link,
 * import or call this file.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * use it, and only as the checker / CPU baseline.
 *
 * PARITY UNPINNED at the yell boundary: the reference's arithmetic library `yell`
 * (git@github.com:fionser/yell.git, un-vendored submodule, pinned commit unknown) is absent from
 * /root/reference and the reference ships no tests, golden vectors or KATs.  The prime table, the
 * choice of psi and the PRNG streams below are therefore fixed by THIS file (docs: DESIGN.md
 * "Arithmetic contract"); everything above that boundary follows the reference's own sources,
 * cited per function as /root/reference-relative file:line.  The restatement is pinned instead by
 * independent mathematics in tests/ (schoolbook negacyclic evaluation, big-integer CRT, decrypt
 * round trips).
 *
 * Word size: T = uint64_t, gT = unsigned __int128, primes < 2^62.
 * Polynomial layout: limb-major contiguous [nlimbs][N] (reference: base_converter.cpp:199).
 */
#ifndef FHE_ORACLE_H
#define FHE_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_ctx orc_ctx;

/* ---- arithmetic contract ------------------------------------------------------------------ */
/* Descending primes p < 2^62 with p = 1 (mod 2^32); psi = x^((p-1)/2N) for the smallest
 * quadratic non-residue x >= 2.  Index 0..L-1 normal, L..L+K-1 special (context.hpp:16-19). */
void orc_gen_primes(uint32_t logN, uint32_t count, uint64_t *primes, uint64_t *psi);
int orc_is_prime(uint64_t n);

orc_ctx *orc_ctx_create(uint32_t logN, uint32_t L, uint32_t K, const uint64_t *primes, const uint64_t *psi);
void orc_ctx_destroy(orc_ctx *c);
uint64_t orc_prime(const orc_ctx *c, uint32_t idx);

/* yell::ops (absent; call sites SURVEY.md §8 a2) */
uint64_t orc_mulmod(uint64_t a, uint64_t b, uint64_t p);
uint64_t orc_addmod(uint64_t a, uint64_t b, uint64_t p);
uint64_t orc_powmod(uint64_t a, uint64_t e, uint64_t p);
uint64_t orc_invmod(uint64_t a, uint64_t p);
uint64_t orc_shoupify(uint64_t y, uint64_t p);
uint64_t orc_mulmod_shoup(uint64_t x, uint64_t y, uint64_t yshoup, uint64_t p);

/* ---- a1: yell::ntt<N>::forward/backward(T*, moduli_idx) ----------------------------------- */
/* natural order in, bit-reversed out at psi^(2*brv(d)+1) (src/yell.cpp:16-21); inputs may be any
 * 64-bit value (residues of another prime: evaluator.cpp:667-668); output canonical [0,p). */
void orc_ntt_forward(const orc_ctx *c, uint64_t *limb, uint32_t moduli_idx);
void orc_ntt_backward(const orc_ctx *c, uint64_t *limb, uint32_t moduli_idx);
/* O(N^2) definition, used by tests to pin the fast transform (small N only) */
void orc_ntt_forward_schoolbook(const orc_ctx *c, uint64_t *out, const uint64_t *in, uint32_t moduli_idx);

/* ---- a5/a6: src/yell.cpp:4-53 -------------------------------------------------------------- */
void orc_apply_galois(const orc_ctx *c, uint64_t *poly, uint32_t nlimbs, uint64_t galois);
/* limb_idx[k] = global modulus index of limb k (needed for the sign flip, src/yell.cpp:47) */
void orc_apply_galois_non_ntt(const orc_ctx *c, uint64_t *poly, uint32_t nlimbs, uint64_t galois);
uint64_t orc_rotation_galois(const orc_ctx *c, int offset, int dir_right); /* rot_key.cpp:6-24 */

/* ---- a7-a11: BaseConverter (src/base_converter.cpp) ---------------------------------------- */
void orc_approx_convert_to_special_basis(const orc_ctx *c, uint64_t *rop, const uint64_t *op, uint32_t Li, uint32_t k);
void orc_neg_approx_convert_to_normal_basis(const orc_ctx *c, uint64_t *rop, const uint64_t *sp_part, uint32_t j);
int orc_mod_up(const orc_ctx *c, uint64_t *rop, const uint64_t *op, uint32_t Li);     /* :189-203 */
int orc_mod_down(const orc_ctx *c, uint64_t *rop, const uint64_t *op, uint32_t L0);   /* :205-242 */
int orc_rescale(const orc_ctx *c, uint64_t *rop, const uint64_t *op, uint32_t n);     /* :244-274 */
/* table accessors (for upload-parity tests): which: 0 hat_pi_invmod_pi[K], 1 neg_hat_pi_mod_qj[L][K],
 * 2 P_invmod_qj[L], 3 P_mod_qj[L], 4 hat_qj_invmod_qj[L-1][L], 5 hat_qj_mod_pi[L-1][K][L], 6 Q_mod_pi[L-1][K] */
const uint64_t *orc_bconv_table(const orc_ctx *c, int which, size_t *count);

/* ---- a12-a16: Evaluator (src/evaluator.cpp) ------------------------------------------------- */
void orc_cross_mult(const orc_ctx *c, uint64_t *d0, uint64_t *d1, uint64_t *d2, const uint64_t *b0,
                    const uint64_t *a0, const uint64_t *b1, const uint64_t *a1, uint32_t Li); /* :621-647 */
/* keys: beta/alpha as [L][L][N] (key j, limb i); in-place on d0,d1 like the reference (:649-693) */
void orc_relinearize(const orc_ctx *c, uint64_t *d0, uint64_t *d1, const uint64_t *d2, const uint64_t *beta,
                     const uint64_t *alpha, uint32_t Li);
/* :695-710; ct0/ct1 are modified (last limb goes to the power basis) like the reference */
void orc_rescale_ntt(const orc_ctx *c, uint64_t *r0, uint64_t *r1, uint64_t *ct0, uint64_t *ct1, uint32_t Li);
/* Evaluator::multiply_and_relin (:119-126): out_b/out_a get Li-1 limbs */
int orc_mul_relin_rescale(const orc_ctx *c, uint64_t *out_b, uint64_t *out_a, const uint64_t *c0b,
                          const uint64_t *c0a, const uint64_t *c1b, const uint64_t *c1a, uint32_t Li,
                          const uint64_t *beta, const uint64_t *alpha);
void orc_add(const orc_ctx *c, uint64_t *rop, const uint64_t *op, uint32_t nlimbs);              /* :284-285 */
void orc_multiply_plain(const orc_ctx *c, uint64_t *bx, uint64_t *ax, const uint64_t *pt, uint32_t nlimbs); /* :88-111 */
/* rotate_slots(RotKey) (:299-311, 343-399): rk_beta/rk_alpha are [(L+K)][N]; in place on bx, ax */
int orc_rotate(const orc_ctx *c, uint64_t *bx, uint64_t *ax, uint32_t Li, uint64_t galois,
               const uint64_t *rk_beta, const uint64_t *rk_alpha);
/* Evaluator::rotate_slots_non_ntt (src/evaluator.cpp:327-339, 520-574), reference arithmetic reproduced literally */
int orc_rotate_non_ntt(const orc_ctx *c, uint64_t *bx, uint64_t *ax, uint32_t Li, uint64_t galois,
               const uint64_t *rk_beta, const uint64_t *rk_alpha);

/* MixedRotKey variant: rotate_slots(MixedRotKey) evaluator.cpp:313-325 + apply_rotation_key
 * :403-516; keys beta/alpha = [L][L+1][N] (key i, limb j; limb L is the special prime P[L]); needs K >= 1 */
int orc_rotate_mixed(const orc_ctx *c, uint64_t *bx, uint64_t *ax, uint32_t Li, uint64_t galois,
                     const uint64_t *mk_beta, const uint64_t *mk_alpha);
void orc_keygen_mixed_rot(const orc_ctx *c, uint64_t *beta, uint64_t *alpha, const uint64_t *sx, const uint64_t *ext_sx,
                          uint64_t galois, uint64_t seed);                                          /* rot_key.cpp:81-120 */

/* ---- plumbing either side of the path (host only; seeds are this repo's) --------------------- */
uint64_t orc_splitmix64(uint64_t *state);
void orc_uniform_poly(const orc_ctx *c, uint64_t *poly, uint32_t first_idx, uint32_t nlimbs, uint64_t seed);
/* keys.hpp:9-24: sx[L][N], ext_sx[K][N], NTT form, hamming weight hwt */
void orc_keygen_sk(const orc_ctx *c, uint64_t *sx, uint64_t *ext_sx, uint32_t hwt, uint64_t seed);
void orc_keygen_pk(const orc_ctx *c, uint64_t *pk_b, uint64_t *pk_a, const uint64_t *sx, uint64_t seed); /* keys.hpp:39-47 */
void orc_keygen_mult(const orc_ctx *c, uint64_t *beta, uint64_t *alpha, const uint64_t *sx, uint64_t seed); /* mult_key.cpp:3-30 */
void orc_keygen_rot(const orc_ctx *c, uint64_t *beta, uint64_t *alpha, const uint64_t *sx, const uint64_t *ext_sx,
                    uint64_t galois, uint64_t seed);                                                  /* rot_key.cpp:23-79 */
/* encryptor.cpp:9-37; msg is an Li-limb power-basis plaintext */
void orc_encrypt(const orc_ctx *c, uint64_t *bx, uint64_t *ax, const uint64_t *msg, uint32_t Li,
                 const uint64_t *pk_b, const uint64_t *pk_a, uint64_t seed);
/* the arithmetic of encryptor.cpp:15-35 with the sampled polynomials given (power basis, Li limbs):
 * bx = NTT(e0m) + NTT(u)*pk_b, ax = NTT(e1) + NTT(u)*pk_a, where e0m = e0 + msg */
void orc_encrypt_core(const orc_ctx *c, uint64_t *bx, uint64_t *ax, const uint64_t *u, const uint64_t *e0m,
                      const uint64_t *e1, uint32_t Li, const uint64_t *pk_b, const uint64_t *pk_a);
/* decryptor.cpp:11-38: rop = INTT(bx + sx*ax), Li limbs, power basis */
void orc_decrypt(const orc_ctx *c, uint64_t *rop, const uint64_t *bx, const uint64_t *ax, uint32_t Li, const uint64_t *sx);

#ifdef __cplusplus
}
#endif
#endif

/*
 * oracle/tfhe_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C11) of the encrypted functional-bootstrap path that
 * tfhe_fbs_map_amd runs on the GPU.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product never does.
 *
 * PARITY STATUS
 *   * cleartext semantics (table lookup / linear combination, reference
 *     fbs_mapper/fbs_exec_env.py:208-229, negacyclic table contract
 *     fbs_mapper/map_to_fbs.py:81-121): PINNED -- decrypt(oracle(enc(x))) is
 *     checked against golden vectors captured from the reference by import
 *     (tests/golden/ *.json.gz files, made by tests/golden/capture_reference.py).
 *   * ciphertext-level arithmetic: PARITY UNPINNED against any third party.
 *     The reference never executes a homomorphic bootstrap; its would-be
 *     provider (zama-ai/concrete @ nightly-2024.04.17, README.md:17, used only
 *     as a cost-model CLI at experiments/add_exec_estimates.py:14) is Rust,
 *     un-vendored and unbuildable here.  This file restates the published TFHE
 *     algorithms (CGGI20 programmable bootstrap: key switch -> modulus switch
 *     -> blind rotation by CMUX/external product -> sample extraction) over the
 *     prime modulus q = 2^46 - 62*2^13 + 1 (64-bit words) with an exact NTT, so "bit-exact" in
 *     this project means GPU == this oracle, word for word.
 *
 *     ROUNDING CONVENTIONS ARE THE BUILDER'S OWN.  Where the literature leaves a choice -- which representative is
 *     decomposed (signed, centred accumulator), how a value is rounded to l*beta or t*gamma bits (q treated as 2^46),
 *     balanced digits with carries in both the blind rotation and the key switch, Delta = 2*round(q/4p), the test-vector
 *     box boundaries round(j*p/N) -- this file and the GPU kernels were specified TOGETHER by the same author, and
 *     some of those choices were revised during kernel work (round 1: centred decomposition; round 2: balanced
 *     key-switch digits, mean-compensated modulus switch).  Word-for-word agreement between the two is therefore agreement with that specification, not
 *     with any external implementation.  What ties the path to the reference is the DECRYPTED level only: every change
 *     of convention is committed together with a green run of tests/test_oracle_tfhe.py (decrypt == the reference's
 *     cleartext goldens, all table modes) -- a change that breaks decryption cannot hide behind GPU == oracle.
 *
 *     Several tables on one blind rotation (orc_tv0 / orc_build_tv_diff / orc_multi_extract, round 2) restate the
 *     multi-value bootstrap of Carpov, Izabachene, Mollimard (CT-RSA 2019, section 3) with schoolbook products; the
 *     identity TV_F = TV_0 * D_F is checked as polynomial arithmetic and the decrypted results against the same goldens
 *     (tests/test_oracle_tfhe.py).
 *
 * All vectors are canonical residues in [0, q).
 */
#ifndef TFHE_ORACLE_H
#define TFHE_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_Q 0x3FFFFFF84001ULL /* 2^46 - 62*2^13 + 1, prime; 2^14 divides q-1 */
#define ORC_QBITS 46

typedef struct {
    uint32_t n;          /* small LWE dimension                          */
    uint32_t log_n_poly; /* log2 N                                       */
    uint32_t k;          /* GLWE dimension                               */
    uint32_t l_bsk;      /* blind-rotate gadget levels                   */
    uint32_t beta_bsk;   /* log2 of blind-rotate gadget base             */
    uint32_t t_ksk;      /* key-switch levels                            */
    uint32_t gamma_ksk;  /* log2 of key-switch base                      */
    uint32_t p_msg;      /* plaintext modulus p ("fbs_size"): Delta=q/2p */
    uint64_t sigma_lwe;  /* noise std-dev (absolute, units of 1/q) KSK   */
    uint64_t sigma_glwe; /* noise std-dev for BSK rows and fresh inputs  */
    uint32_t bsk_group;  /* key bits per blind-rotation step: 0/1 = one, 2 = two (n even) */
    uint32_t reserved;
} orc_params;

typedef struct orc_ctx orc_ctx;

/* --- field + NTT unit-test hooks ---------------------------------------- */
uint64_t orc_gl_mul(uint64_t a, uint64_t b);
uint64_t orc_gl_mul_slow(uint64_t a, uint64_t b);
uint64_t orc_gl_pow(uint64_t a, uint64_t e);
/* c = a*b mod (X^N+1, q): O(N^2) schoolbook and NTT versions */
void orc_negacyclic_mul_schoolbook(const uint64_t *a, const uint64_t *b, uint64_t *c, uint32_t N);
void orc_negacyclic_mul_ntt(const uint64_t *a, const uint64_t *b, uint64_t *c, uint32_t log_n);

/* --- deterministic randomness (ChaCha20, see DESIGN.md "Randomness") ---- */
uint64_t orc_rand64(uint64_t seed, uint64_t stream, uint64_t idx);
int64_t  orc_noise(uint64_t seed, uint64_t stream, uint64_t idx, uint64_t sigma);

/* --- context / keys ------------------------------------------------------ */
orc_ctx *orc_create(const orc_params *p, uint64_t seed);
void     orc_destroy(orc_ctx *c);
/* generate keys from the seed (same derivation as the product's fbs_keygen) */
void     orc_keygen(orc_ctx *c);
/* or install keys exported by the product (coefficient-domain BSK):
 *   sk_lwe[n], sk_glwe[k*N] (0/1), bsk[G][(k+1)l][k+1][N], ksk[kN][t][n+1]
 * G GGSW samples: n of them (one per key bit) when bsk_group <= 1; with bsk_group = 2 three per PAIR of key bits
 * (s0, s1) = (sk[2i], sk[2i+1]), encrypting s0(1-s1), (1-s0)s1, s0 s1 in this order: G = 3n/2 */
void     orc_set_keys(orc_ctx *c, const uint64_t *sk_lwe, const uint64_t *sk_glwe,
                      const uint64_t *bsk, const uint64_t *ksk);
const uint64_t *orc_sk_lwe(const orc_ctx *c);
const uint64_t *orc_sk_glwe(const orc_ctx *c);
const uint64_t *orc_bsk(const orc_ctx *c);
const uint64_t *orc_ksk(const orc_ctx *c);

uint64_t orc_delta_half(const orc_ctx *c); /* round(q/4p); Delta = 2*that */

/* --- encrypt / decrypt under the big key (dimension kN) ----------------- */
/* ct i uses randomness streams indexed by (nonce0 + i) */
void orc_encrypt(const orc_ctx *c, const int64_t *msgs, size_t count, uint64_t nonce0, uint64_t *cts);
void orc_decrypt(const orc_ctx *c, const uint64_t *cts, size_t count, int64_t *msgs);
/* raw phase b - <a,s> */
void orc_phase(const orc_ctx *c, const uint64_t *cts, size_t count, uint64_t *phases);

/* --- test vector --------------------------------------------------------- */
/* table[len] -> tv[N] and the post-add constant; returns 0, or -1 when the
 * table violates the negacyclic contract for p (map_to_fbs.py:81-98) */
int orc_build_tv(const orc_ctx *c, const int32_t *table, uint32_t len, uint64_t *tv, uint64_t *post_add);

/* --- several tables on one blind rotation (multi-value bootstrap, CIM19; see tfhe_oracle.c) --- */
void orc_tv0(const orc_ctx *c, uint64_t *tv);                    /* delta_half * (1 + X + .. + X^(N-1)) */
/* diff[N]: the integer polynomial D_F with TV_F = TV_0 * D_F; post_add and errors as orc_build_tv */
int orc_build_tv_diff(const orc_ctx *c, const int32_t *table, uint32_t len, int32_t *diff, uint64_t *post_add);
/* ct_out = SampleExtract_0(acc * D_F) + post_add for acc = orc_blind_rotate(ms, TV_0) */
void orc_multi_extract(const orc_ctx *c, const uint64_t *acc, const int32_t *diff, uint64_t post_add, uint64_t *ct_big);

/* --- the path, stage by stage ------------------------------------------- */
void orc_lincomb(const orc_ctx *c, const uint64_t *const *srcs, const int64_t *coefs, uint32_t n_src,
                 int64_t const_coef, uint64_t *out);
void orc_keyswitch(const orc_ctx *c, const uint64_t *ct_big, uint64_t *ct_small);
void orc_modswitch(const orc_ctx *c, const uint64_t *ct_small, uint32_t *ms);
void orc_blind_rotate(const orc_ctx *c, const uint32_t *ms, const uint64_t *tv, uint64_t *acc /* (k+1)*N */);
void orc_sample_extract(const orc_ctx *c, const uint64_t *acc, uint64_t post_add, uint64_t *ct_big);
/* one functional bootstrap = KS + MS + BR + SE (+post-add) */
void orc_bootstrap(const orc_ctx *c, const uint64_t *ct_in, const uint64_t *tv, uint64_t post_add, uint64_t *ct_out);
/* batch of `count` bootstraps, tv_idx[i] selects tvs[tv_idx[i]*N..];
 * OpenMP over the batch with `threads` threads (<=0: all). returns threads used */
int orc_bootstrap_batch(const orc_ctx *c, const uint64_t *cts_in, const uint32_t *tv_idx, const uint64_t *tvs,
                        const uint64_t *post_adds, size_t count, uint64_t *cts_out, int threads);

#ifdef __cplusplus
}
#endif
#endif

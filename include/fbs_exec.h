/*
 * fbs_exec.h -- C ABI of libfbsexec.so, the MI355X (gfx950) executor for the
 * "linear combination + functional bootstrap" programs of ssmiler/tfhe_fbs_map.
 *
 * The reference has NO native/FFI boundary: its executor is the Python method
 * `LutExecEnv.eval` (fbs_mapper/fbs_exec_env.py:208-229) and the only call
 * site is fbs_mapper/map_circuit.py:174.  This header is therefore the
 * boundary a maintainer would bind with ctypes from that method (stub in
 * INTEGRATION.md).  Each entry point names the reference construct it stands
 * behind.
 *
 * Conventions: every function returns 0 on success and a negative FBS_E_* code
 * on failure (never throws, never aborts: every entry point is an exception
 * barrier -- a host allocation that fails inside the library comes back as
 * FBS_E_NOMEM, any other internal exception as FBS_E_INVALID -- and counts are
 * checked against FBS_MAX_* before anything is sized by them; what the library
 * cannot check is that a caller's array is as long as its count says);
 * `fbs_last_error` gives the text.
 * Host buffers are caller-allocated, C-contiguous, 64-bit words unless stated;
 * the library owns device memory and keys behind opaque handles.  A context is
 * bound to one GPU and must be driven by one host thread at a time.  There is
 * no CPU fallback: without a usable gfx950 device `fbs_ctx_create` fails.
 *
 * Ciphertexts are LWE samples over Z_q, q = 2^46 - 62*2^13 + 1 = 0x3FFFFFF84001
 * (prime; the ciphertext modulus and the NTT modulus are the same), one residue
 * per 64-bit word, under the "big" key of dimension D = k*N: D mask words then
 * the body, all canonical (< q).  A message m in [0, 2p) is encoded as
 * m * Delta, Delta = 2*round(q/4p), with p = `p_msg` the reference's `fbs_size`
 * (map_circuit.py:97,117-122).
 *
 * Streams.  Entry points that take a `stream` queue their work on it and return
 * without waiting -- except a call that needs more scratch than any earlier one,
 * which blocks while the scratch grows (fbs_ctx_reserve sizes it up front).  All
 * calls on a context share its scratch buffers: calls on one stream are ordered
 * by the stream, and a call on a different stream first waits (on the device)
 * for the previous call that used the scratch.
 */
#ifndef FBS_EXEC_H
#define FBS_EXEC_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FBS_OK 0
#define FBS_E_INVALID (-1)   /* bad argument / unsupported parameter set          */
#define FBS_E_DEVICE (-2)    /* HIP error (no GPU, OOM, launch failure ...)       */
#define FBS_E_STATE (-3)     /* call out of order (e.g. eval before keygen)       */
#define FBS_E_TABLE (-4)     /* table violates the negacyclic contract for p      */
#define FBS_E_POLY_SIZE (-5) /* polynomial size rejected (see fbs_poly_size_check) */
#define FBS_E_NOMEM (-6)     /* the HOST could not allocate what the arguments ask for */

typedef struct fbs_params {
    uint32_t n;          /* small LWE dimension (P1024: 630)                      */
    uint32_t log_n_poly; /* log2 of the GLWE polynomial size N (P1024: 10)        */
    uint32_t k;          /* GLWE dimension: 1 (N = 256 .. 4096); 2, 3, 4 at N = 256 / 512; 2, 3 at N = 1024 (any l_bsk, bsk_group 1 or 2) */
    uint32_t l_bsk;      /* blind-rotation gadget levels (3)                      */
    uint32_t beta_bsk;   /* log2 blind-rotation gadget base (7)                   */
    uint32_t t_ksk;      /* key-switch levels (8)                                 */
    uint32_t gamma_ksk;  /* log2 key-switch base (2)                              */
    uint32_t p_msg;      /* plaintext modulus p = fbs_size                        */
    uint64_t sigma_lwe;  /* std-dev of key-switch-key noise, in units of 1/q       */
    uint64_t sigma_glwe; /* std-dev of bootstrap-key and fresh-input noise, same  */
    uint32_t bsk_group;  /* key bits consumed per blind-rotation step: 0 or 1 = one   */
                         /* (n CMUX steps); 2 = two ("multi-bit": n/2 steps on a     */
                         /* bundle of three GGSW samples per pair of key bits; n even)*/
    uint32_t reserved;   /* 0                                                       */
} fbs_params;

/* Polynomial sizes.  fbs_params carries log2 N: the ring is Z_q[X]/(X^N + 1) with N a power of two (this build:
 * 256 .. 4096).  BASELINE config 5 also names "non-power-of-two N".  That is rejected on purpose, not for lack of a
 * transform (2N | q - 1 holds for N = 3 * 2^k under this modulus): for N = m * 2^k with m odd > 1, X^N + 1 is not
 * cyclotomic -- y^m + 1 is divisible by y + 1, so X^N + 1 has the factor X^(2^k) + 1 and every GLWE sample maps onto
 * the power-of-two ring of degree N/m, whose (smaller) dimension then bounds the security: N = 1536 is no safer than
 * N = 512 and costs three times as much.  The reference's patch generalises the plaintext modulus p, not N
 * (experiments/concrete.patch:85-90); odd p at power-of-two N is what this library runs for that config.
 * Returns FBS_OK, FBS_E_POLY_SIZE for a non-power-of-two (text via fbs_last_error(NULL)), FBS_E_INVALID for a power
 * of two outside the supported range. */
int fbs_poly_size_check(uint32_t poly_size);

typedef struct fbs_ctx fbs_ctx;
typedef struct fbs_tvset fbs_tvset;
typedef struct fbs_prog fbs_prog;

/* ---- context ------------------------------------------------------------ */
/* device: HIP ordinal.  seed: all key material and encryption randomness is
 * derived from it (ChaCha20 streams, DESIGN.md "Randomness").
 *
 * RANDOMNESS GRADE.  fbs_ctx_create is the REPRODUCIBLE form: 64 bits of seed, the
 * same keys for the same seed whatever the parameter set -- for tests, benchmarks
 * and checkers that must be keyed identically (oracle/).  It is not a way to make
 * production keys, whatever noise the parameter set carries.  fbs_ctx_create_seeded
 * keys the generator with 32 caller-supplied bytes (e.g. from the OS) and mixes the
 * parameter set into the derivation, so that two parameter sets under one seed
 * share no key material.  In BOTH forms the noise sampler is an integer
 * Irwin-Hall(12) stand-in for a discrete Gaussian (bounded at 6 sigma): test-grade.
 * A deployment that needs more brings its own keys with fbs_import_keys. */
int fbs_ctx_create(const fbs_params *params, uint64_t seed, int device, fbs_ctx **out);
int fbs_ctx_create_seeded(const fbs_params *params, const uint8_t seed[32], int device, fbs_ctx **out);
void fbs_ctx_destroy(fbs_ctx *ctx);
/* Scratch (modulus-switched rows, the key switch's digit/limb buffers, accumulators of shared rotations, the wire
 * buffer of fbs_eval) grows on demand, and growing BLOCKS: the call that needs more than any earlier one waits for
 * the context's queued work, frees and reallocates.  A host that wants every *_dev call to be kernel launches and
 * nothing else (several streams, collectives between levels) sizes it once: max_keyswitches = the largest number
 * of key switches one call will ask for (fbs_bootstrap_batch_dev: count; fbs_level_bootstrap_dev: the slice's
 * sources x samples; fbs_eval_dev: fbs_layout.max_sources x T), max_shared_rows = shared rotations x samples of a
 * fused program's widest level (0 otherwise), wire_words = n_slots x T x (D+1) for fbs_eval_dev (0 otherwise).
 * Zero leaves a buffer as it is.  After it, the only blocking case left is a call that exceeds what was reserved. */
int fbs_ctx_reserve(fbs_ctx *ctx, size_t max_keyswitches, size_t max_shared_rows, size_t wire_words);
/* Launcher knobs -- which kernel shape a launch takes.  The defaults are the measured choices; tests set them to reach
 * every launcher branch in one process, tools/ to time one shape against another.  Results never depend on them.
 *   "ks_gemm_min" (1)  key switches per launch from which the int8 GEMM on the matrix cores is used
 *   "ks_mfma" (1), "ks_fp" (1), "ks_cols_major" (1), "ks_split" (0 = automatic)   key-switch fallbacks / grid order
 *   "br_whole_cu" (1)       whole rounds of a launch as one four-bootstrap workgroup per CU
 *   "br_cu_kernel" (1)      launches that leave most of the chip empty as ONE bootstrap per CU (eight waves)
 *   "br_cu_max_per_cu" (2)  ... up to this many bootstraps per CU
 *   "br_cu_lean" (1)        ... and between one and two per CU as two 128-register workgroups per CU (2: always, 0: never) */
int fbs_ctx_tune(fbs_ctx *ctx, const char *knob, int64_t value);
/* counters: "scratch_growths" (how often a call (re)allocated scratch, i.e. blocked), "ms_capacity", "acc_capacity",
 * "wires_capacity", "next_nonce", "cu_count" */
int fbs_ctx_stat(const fbs_ctx *ctx, const char *name, int64_t *value);
/* text of the last failure on `ctx` (or of the last failed fbs_ctx_create when ctx == NULL) */
const char *fbs_last_error(const fbs_ctx *ctx);
/* "gfx950 <device name> CUs=<n>" of the bound device */
const char *fbs_device_info(const fbs_ctx *ctx);

/* ---- keys ----------------------------------------------------------------
 * Secret keys stay on the host; the bootstrapping key (n GGSW samples) is
 * uploaded, transformed to the NTT domain on the GPU and kept resident, as is
 * the key-switching key. */
int fbs_keygen(fbs_ctx *ctx);
/* word counts of { sk_lwe, sk_glwe, bsk, ksk } in the standard (coefficient) layout:
 *   sk_lwe[n], sk_glwe[k][N] (bit c N + j = coefficient j of key polynomial S_c; read flat it is the key of the
 *     extracted LWE ciphertexts, dimension k N);
 *   bsk[G][(k+1) l][k+1][N]: GGSW sample g, row rr = comp l + lv (comp = 0 .. k: the GLWE component the gadget
 *     factor sits on, k = the body; lv = gadget level), then the row's k + 1 polynomials: columns 0 .. k-1 the mask
 *     A_0 .. A_(k-1), column k the body B = sum_c A_c S_c + e -- and the message: bit g_lv added to coefficient 0
 *     of column comp, g_lv = round(q / 2^(beta (lv+1)));
 *     bsk_group <= 1: G = n, sample g encrypts key bit sk_lwe[g];
 *     bsk_group = 2 (the default 128-bit sets): G = 3 n / 2, samples 3 i, 3 i + 1, 3 i + 2 belong to the pair
 *     (s0, s1) = (sk_lwe[2 i], sk_lwe[2 i + 1]) and encrypt s0 (1 - s1), (1 - s0) s1 and s0 s1, in that order;
 *   ksk[k N][t][n+1]: row (j, v) = LWE under sk_lwe (n mask words, then the body) of sk_glwe[j] h_v,
 *     h_v = round(q / 2^(gamma (v+1))), j = c N + coefficient. */
int fbs_key_sizes(const fbs_ctx *ctx, size_t sizes[4]);
/* test hook: copy keys out (any pointer may be NULL) so a checker can be keyed identically */
int fbs_export_keys(const fbs_ctx *ctx, uint64_t *sk_lwe, uint64_t *sk_glwe, uint64_t *bsk, uint64_t *ksk);
/* The mirror: keys made elsewhere (a caller's own CSPRNG and Gaussian sampler, or a checker's) in the layout of
 * fbs_key_sizes, instead of fbs_keygen.  Secret keys are binary, every other word a canonical residue (< q); the
 * evaluation keys must encrypt the secrets under this library's gadget conventions (GGSW row
 * (c, l) = GLWE(0) + s g_l on component c, g_l = round(q / 2^(beta (l+1))); key-switching row (j, v) = LWE(s_j h_v)).
 * The secret keys stay on the host and serve fbs_encrypt / fbs_decrypt only.  Besides the ranges, the call DECRYPTS a few
 * GGSW samples (first, middle, last; every row) and key-switching rows with the supplied secrets and refuses
 * (FBS_E_INVALID) keys whose phases are not what this layout says they encrypt, within 16 standard deviations of the
 * parameter set's noises: a key in another order fails here, not as garbage after the first bootstrap. */
int fbs_import_keys(fbs_ctx *ctx, const uint64_t *sk_lwe, const uint64_t *sk_glwe, const uint64_t *bsk, const uint64_t *ksk);

/* ---- encrypt / decrypt (host side, big key) ------------------------------
 * Stand behind the Input arm of eval (fbs_exec_env.py:213-214) and the final
 * read-out (:225-229).  Ciphertext i draws its randomness from stream
 * nonce0 + i, so a run is reproducible. cts: [count][D+1]. */
int fbs_encrypt(const fbs_ctx *ctx, const int64_t *msgs, size_t count, uint64_t nonce0, uint64_t *cts);
/* The same on streams nobody has used: the context keeps a counter over [2^55, 2^56) (fbs_encrypt's explicit nonces stay
 * below 2^55), so two calls never share mask or noise.  *nonce0 (may be NULL) = the first stream this call took. */
int fbs_encrypt_fresh(fbs_ctx *ctx, const int64_t *msgs, size_t count, uint64_t *cts, uint64_t *nonce0);
/* msgs[i] = round(phase * 2p / q) mod 2p */
int fbs_decrypt(const fbs_ctx *ctx, const uint64_t *cts, size_t count, int64_t *msgs);

/* ---- tables -> test vectors ----------------------------------------------
 * One entry per distinct `Bootstrap.table` (fbs_exec_env.py:51-61).  Table t is
 * table_vals[table_off[t] .. table_off[t+1]); length <= 2p, and where it
 * exceeds p it must satisfy table[i] + table[i+p] == const (the three modes of
 * map_to_fbs.py:81-98), else FBS_E_TABLE. */
int fbs_tvset_create(fbs_ctx *ctx, const int32_t *table_vals, const uint32_t *table_off, uint32_t n_tables,
                     fbs_tvset **out);
void fbs_tvset_destroy(fbs_tvset *tv);

/* ---- batch of independent functional bootstraps (BASELINE config 2) -------
 * One FBS = key switch (kN -> n), modulus switch (q -> 2N), blind rotation
 * (n CMUX), sample extraction: the encrypted form of `table[v]`
 * (fbs_exec_env.py:218-220).  Host buffers: cts_in/out [count][D+1]. */
int fbs_bootstrap_batch(fbs_ctx *ctx, const fbs_tvset *tv, const uint64_t *cts_in, const uint32_t *table_ids,
                        size_t count, uint64_t *cts_out);
/* same on device-resident buffers, asynchronous on `stream` (a hipStream_t; NULL = the
 * context's own stream).  d_table_ids is a device array of `count` uint32; being device
 * memory it cannot be checked by the host: an id >= the set's size selects table 0. */
int fbs_bootstrap_batch_dev(fbs_ctx *ctx, const fbs_tvset *tv, const uint64_t *d_cts_in,
                            const uint32_t *d_table_ids, size_t count, uint64_t *d_cts_out, void *stream);

/* ---- linear combination (LinearProd, fbs_exec_env.py:37-49, :215-217) ------
 * out[g][s] = sum_i coefs[off[g]+i] * wires[srcs[off[g]+i]][s] + consts[g]*Delta  for g < n_out,
 * s < T.  `d_wires` is a device array laid out [wire][T][D+1]; outputs are written to wire
 * slots dst[g] of the same array.  term_off has n_out+1 entries.  HOST index arrays: this call
 * and fbs_bootstrap_wires_dev stage them through a per-context buffer and wait for that copy
 * (a convenience for tests and one-off calls; to step a program use fbs_level_* below). */
int fbs_lincomb_dev(fbs_ctx *ctx, uint64_t *d_wires, size_t T, uint32_t n_out, const uint32_t *dst,
                    const uint32_t *term_off, const uint32_t *srcs, const int64_t *coefs, const int64_t *consts,
                    void *stream);
/* bootstraps over wire slots: wire dst[g] = FBS(wire src[g], table table_ids[g]) for samples
 * [s_begin, s_end) of each gate (the slice a rank owns in gate-sharded multi-GPU mode). */
int fbs_bootstrap_wires_dev(fbs_ctx *ctx, const fbs_tvset *tv, uint64_t *d_wires, size_t T, uint32_t n_gates,
                            const uint32_t *src, const uint32_t *dst, const uint32_t *table_ids, size_t s_begin,
                            size_t s_end, void *stream);

/* ---- whole program (LutExecEnv.eval, fbs_exec_env.py:208-229) ---------------
 * Flat description of `LutExecEnv.instructions` after the Input entries:
 *   wire ids: 0..n_inputs-1 are the inputs in program order, n_inputs+i is
 *   instruction i.  kind[i]: 0 = LinearProd, 1 = Bootstrap.
 *   LinearProd i: terms [arg0[i], arg0[i]+arg1[i]) of (term_coef, term_src), constant const_coef[i].
 *   Bootstrap  i: source wire arg0[i], table id arg1[i] (into the fbs_tvset).
 *   outputs: out_wire[o] >= 0 is a wire id; a constant output c is encoded as -1-c. */
#define FBS_MAX_WIRES (1u << 28)   /* n_inputs + n_instr, and n_outputs, of one program */
#define FBS_MAX_TERMS (1u << 30)   /* n_terms of one program */
#define FBS_MAX_TABLES (1u << 20)  /* tables of one fbs_tvset */
typedef struct fbs_program_desc {
    uint32_t n_inputs, n_instr, n_terms, n_outputs;
    const uint8_t *kind;
    const uint32_t *arg0, *arg1;
    const int64_t *const_coef;
    const int64_t *term_coef;
    const uint32_t *term_src;
    const int64_t *out_wire;
} fbs_program_desc;

int fbs_program_load(fbs_ctx *ctx, const fbs_program_desc *desc, const fbs_tvset *tv, fbs_prog **out);
/* The same with options.  FBS_LOAD_FUSE_TABLES: several tables on ONE blind rotation (the reference's
 * one-gate-one-bootstrap lowering puts several tables on one linear combination, fbs_mapper/map_to_fbs.py:41-45, and its
 * CSE merges identical tables only, fbs_exec_env.py:93-100).  A source wire that two or more Bootstraps read is rotated
 * ONCE, from the table-independent test vector TV_0 = Delta/2 (1 + X + .. + X^(N-1)); each table F is then cut out of that
 * accumulator by a product with the small integer polynomial D_F (TV_F = TV_0 * D_F; multi-value bootstrap, Carpov,
 * Izabachene, Mollimard, CT-RSA 2019) and a sample extraction.  Same decrypted results; such an output carries more noise
 * than an ordinary bootstrap's (fbs_table_fusion_norms: the caller's parameter choice must carry it), and the ciphertexts
 * differ from the unfused program's.  Into the wire slots a level of a fused program runs whole (fbs_eval, fbs_eval_dev,
 * fbs_level_bootstrap_dev over the full range without d_rows).  Across GPUs the unit dealt out is the ROTATION:
 * fbs_level_bootstrap_dev with d_rows takes any slice of the level's (rotation, sample) grid, its rows are
 * fbs_layout.row_words = (k + 1) N words -- an ordinary gate leaves its ciphertext there, a shared rotation its whole accumulator --
 * and fbs_level_scatter_dev, given all rows of the level, files the ciphertexts and cuts every table out of the gathered
 * accumulators. */
#define FBS_LOAD_FUSE_TABLES 1u
int fbs_program_load_ex(fbs_ctx *ctx, const fbs_program_desc *desc, const fbs_tvset *tv, uint32_t flags, fbs_prog **out);
/* What sharing a rotation does to the noise of table `table`'s output, with TV_F = Delta/2 G_F(X) (G_j = +-(2 f - c)):
 * *d_norm2 = |D_F|^2, *g_norm2 = |G_F|^2 (sums over the N coefficients).  The part of the blind-rotation noise that comes
 * from the bootstrapping key's noise has independent coefficients and grows by |D_F|^2; the part that comes from rounding
 * the accumulator is seen through the BINARY key S, whose mean 1/2 makes S D_F = G_F / 2 + (centred part) D_F: it grows
 * by |D_F|^2 / 2 + |G_F|^2 / (2N).  max of the two bounds the variance factor for any mix. */
int fbs_table_fusion_norms(const fbs_tvset *tv, uint32_t table, uint64_t *d_norm2, uint64_t *g_norm2);
void fbs_program_destroy(fbs_prog *prog);
/* depth (number of bootstrap levels) and the widest level, as scheduled */
int fbs_program_info(const fbs_prog *prog, uint32_t *n_levels, uint32_t *max_width, uint32_t *n_bootstrap);
/* in_cts: host [n_inputs][T][D+1]; out_cts: host [n_outputs][T][D+1] (constant outputs are
 * written as trivial ciphertexts).  Levels are batched over (gate, sample); samples are
 * evaluated in chunks whose wire slots fit in HBM. */
int fbs_eval(fbs_ctx *ctx, fbs_prog *prog, const uint64_t *in_cts, size_t T, uint64_t *out_cts);
/* the same on device-resident buffers, asynchronous on `stream` */
int fbs_eval_dev(fbs_ctx *ctx, fbs_prog *prog, const uint64_t *d_in, size_t T, uint64_t *d_out, void *stream);

/* ---- a loaded program, one level at a time (multi-GPU hosts) -----------------
 * The two independent axes of the reference's eval loop (fbs_exec_env.py:211-223) are the gates
 * of a bootstrap level and the samples.  A host that shards the GATES of a level over several
 * GPUs keeps a replicated wire buffer per GPU, [n_slots][T][D+1] words of its own device
 * memory, and steps the program with the calls below; every index array they need was
 * uploaded by fbs_program_load, so each call is a few kernel launches on `stream` and
 * nothing else.  Wires live in SLOTS: a wire's slot is reused once its last reader has run,
 * so n_slots is the peak number of live wires, not the number of wires.
 *   level L in [0, n_levels]:  fbs_level_lincomb_dev    the LinearProds of level L
 *   level L in [0, n_levels):  fbs_level_bootstrap_dev  bootstraps f in [f_begin, f_end) of the
 *        level's [n_gates][s_count] grid (gate-major), each preceded by the key switch of its
 *        source -- one key switch per distinct (source wire, sample), shared by the gates that
 *        read it.  d_rows == NULL: results go to their wire slots; else to row f - f_begin of
 *        d_rows ([f_end - f_begin][fbs_layout.row_words], e.g. the send buffer of an all-gather), and
 *   fbs_level_scatter_dev copies rows of such an array (after the all-gather) into the slots.
 * Samples [s_begin, s_begin + s_count) of every wire are processed; T is the sample stride. */
typedef struct fbs_layout {
    uint32_t n_slots;      /* wire slots a wire buffer needs                                   */
    uint32_t n_levels;     /* bootstrap levels                                                 */
    uint32_t max_width;    /* bootstraps in the widest level                                   */
    uint32_t max_sources;  /* key switches in the level that has most                          */
    uint32_t n_bootstrap;  /* bootstraps in the program                                        */
    uint32_t n_keyswitch;  /* key switches in the program (<= n_bootstrap: shared sources)     */
    uint32_t n_inputs, n_outputs;
    uint32_t n_rotations;  /* blind rotations per sample (< n_bootstrap when tables share them)       */
    uint32_t row_words;    /* words per row of the d_rows arrays below: D + 1, or (k + 1) N for a fused program */
} fbs_layout;
int fbs_program_layout(const fbs_prog *prog, fbs_layout *out);
int fbs_program_level(const fbs_prog *prog, uint32_t level, uint32_t *n_gates, uint32_t *n_sources);
/* in_slot[n_inputs]: where input i is expected; out_slot[n_outputs]: slot of output o, or -1-c for a constant c */
int fbs_program_io_slots(const fbs_prog *prog, uint32_t *in_slot, int64_t *out_slot);
int fbs_level_lincomb_dev(fbs_ctx *ctx, const fbs_prog *prog, uint32_t level, uint64_t *d_wires, size_t T,
                          size_t s_begin, size_t s_count, void *stream);
int fbs_level_bootstrap_dev(fbs_ctx *ctx, const fbs_prog *prog, uint32_t level, uint64_t *d_wires, size_t T,
                            size_t s_begin, size_t s_count, size_t f_begin, size_t f_end, uint64_t *d_rows,
                            void *stream);
int fbs_level_scatter_dev(fbs_ctx *ctx, const fbs_prog *prog, uint32_t level, uint64_t *d_wires, size_t T,
                          size_t s_begin, size_t s_count, const uint64_t *d_rows, size_t f_begin, size_t f_end,
                          void *stream);

/* ---- measurement hooks ------------------------------------------------------
 * When enabled, every kernel launch is bracketed by HIP events on its own
 * stream; fbs_profile_read synchronises and returns per-kernel totals since the
 * last reset: ms[0]=keyswitch+modswitch, ms[1]=blind-rotate+extract, ms[2]=lincomb;
 * launches[i] = number of launches. */
int fbs_profile_enable(fbs_ctx *ctx, int on);
int fbs_profile_read(fbs_ctx *ctx, double ms[3], uint64_t launches[3], int reset);
/* name of the kernel instantiation the most recent launch of kind `which` (0, 1, 2 as above) used: the launcher
 * picks the shape by parameter set and batch size */
const char *fbs_profile_kernel(const fbs_ctx *ctx, int which);
/* The same totals per kernel instantiation since the last reset (fbs_profile_read with reset clears them too): lines
 * "<kind>\t<kernel>\t<launches>\t<ms>\n", kind = 0, 1, 2 as above.  A launch the launcher cuts into a whole-round part and
 * a remainder shows as two entries.  *needed (may be NULL) = bytes the text takes; buf == NULL only asks for that. */
int fbs_profile_kernels(fbs_ctx *ctx, char *buf, size_t cap, size_t *needed);
/* newline-separated names of every key-switch and blind-rotation kernel instantiation the launchers can pick, by the
 * rules of their dispatch: what tests/test_gpu_dispatch.py drives one by one against the oracle */
const char *fbs_kernel_catalog(void);
/* block until all work queued on the context's stream (or `stream`) has finished */
int fbs_sync(fbs_ctx *ctx, void *stream);

/* ---- mapper: coefficient search ------------------------------------
 * Stands behind MapToFBSHeur._find_lincomb_coefs_search (fbs_mapper/map_to_fbs.py:363-392): x, y are the multi-value
 * columns of two cones over the `rows` rows of their joint truth table (xy_mvt[:, 0], xy_mvt[:, 1]), tt the merged output
 * bit per row (r_tt).  On success *found says whether a legal (a, b) exists; then ab = {a, b} and, if mvt != NULL,
 * mvt[r] = a x[r] + b y[r] -- the reference's (r_ab, r_mvt), chosen by the reference's rule among the reference's
 * candidates.  Needs no keys: a searcher is bound to a device only.  max_fbs_size <= 128. */
typedef struct fbs_searcher fbs_searcher;
int fbs_searcher_create(int device, fbs_searcher **out);
void fbs_searcher_destroy(fbs_searcher *s);
const char *fbs_searcher_last_error(const fbs_searcher *s);
/* device time of the most recent search kernel (HIP events on the searcher's stream), ms */
double fbs_searcher_last_kernel_ms(const fbs_searcher *s);
int fbs_search_lincomb_coefs(fbs_searcher *s, const int32_t *x, const int32_t *y, const uint8_t *tt, uint32_t rows,
                             uint32_t fbs_size, uint32_t max_fbs_size, int32_t ab[2], int64_t *mvt, int *found);

/* ---- debug hook: negacyclic product of two polynomials on the device NTT ---- */
int fbs_debug_polymul(fbs_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *c);
/* ---- test hook: raises a C++ exception INSIDE the library (kind 0 std::bad_alloc, 1 std::length_error, 2 std::runtime_error,
 * 3 a non-standard one; else nothing) to show that none crosses this boundary: returns FBS_E_NOMEM, FBS_E_NOMEM,
 * FBS_E_INVALID, FBS_E_INVALID, FBS_OK, with the text in fbs_last_error(ctx) (ctx may be NULL: no device is touched). */
int fbs_debug_raise(fbs_ctx *ctx, int kind);

#ifdef __cplusplus
}
#endif
#endif /* FBS_EXEC_H */

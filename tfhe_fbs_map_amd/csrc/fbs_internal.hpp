// Internal definitions shared by the host side (fbs_host.cpp, fbs_capi.cpp) and the HIP side
// (fbs_kernels.hip) of libfbsexec.so.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <map>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/fbs_exec.h"
#include "fbs_field.hpp"

namespace fbs {

// ---- randomness: ChaCha20 keyed by the context seed (spec in DESIGN.md) -----------------------
enum Domain : uint64_t {
    DOM_SK_LWE = 1, DOM_SK_GLWE = 2, DOM_BSK_MASK = 3, DOM_BSK_NOISE = 4,
    DOM_KSK_MASK = 5, DOM_KSK_NOISE = 6, DOM_ENC_MASK = 7, DOM_ENC_NOISE = 8
};
inline uint64_t stream_id(Domain d, uint64_t sub) { return ((uint64_t)d << 56) | (sub & 0x00FFFFFFFFFFFFFFull); }
// the 256-bit ChaCha key of a context: a 64-bit seed followed by a fixed tail (fbs_ctx_create: reproducible, test-grade), or
// derived from 32 caller-supplied bytes and the parameter set (fbs_ctx_create_seeded)
struct RandKey {
    uint32_t w[8];
};
RandKey rand_key_from_seed64(uint64_t seed);
RandKey rand_key_derive(const uint8_t seed[32], const fbs_params &p);
void rand_words(const RandKey &key, uint64_t stream, uint64_t idx0, uint64_t *dst, size_t count);
int64_t noise_sample(const RandKey &key, uint64_t stream, uint64_t idx, uint64_t sigma);

// launcher knobs (fbs_ctx_tune): which kernel shape a launch takes.  Defaults are the measured choices; tests use them to reach
// every launcher branch in one process, tools/ to measure one against another.
struct Tune {
    int64_t ks_gemm_min = 1;        // key switches per launch from which the int8 GEMM on the matrix cores is used
    int64_t ks_mfma = 1;            // 0: never the GEMM
    int64_t ks_fp = 1;              // 0: integer kernels instead of the FP64 one (fallback path)
    int64_t ks_cols_major = 1;      // 0: round-1 grid order of the vector key-switch kernels
    int64_t ks_split = 0;           // > 0: k slices of the GEMM
    int64_t br_whole_cu = 1;        // 0: no whole-CU workgroups (four bootstraps per workgroup) in the blind rotation
    int64_t br_cu_kernel = 1;       // 0: no one-bootstrap-per-CU kernel (the generic kernel on the four-wave transform instead)
    int64_t br_cu_max_per_cu = 2;   // bootstraps per CU up to which a launch takes the one-bootstrap-per-CU kernel
    int64_t br_cu_lean = 1;         // 1: between one and two per CU, its 128-register variant (two workgroups per CU); 2: always; 0: never
    int64_t br_k2_shape = 0;        // k = 2: 0 by launch size; 3 always three waves per bootstrap; 12 always the twelve-wave latency shape
    int64_t br_glwe_fpw = 0;        // k_blind_rotate_glwe: bootstraps per workgroup -- 0 by launch size; 1, 2; anything larger = the throughput shape
};

// ---- launch descriptors ------------------------------------------------------------------------
// A "gate" is one Bootstrap instruction applied to `s_count` samples of its source wire.  The bootstraps of a launch
// are the flattened indices f in [f_begin, f_begin + count) of the [n_gates][s_count] grid: gate g = f / s_count,
// sample s = s_begin + f % s_count.  Wire slot w, sample s lives at base + (w*T + s)*ct_words words.  Slot arrays may
// be null (identity), which is the plain batch API.
struct GateView {
    const uint64_t *in_base;
    uint64_t *out_base;
    const uint32_t *src_slot;   // [n_sources] or null: wire slot of key-switch source u
    const uint32_t *dst_slot;   // [n_gates] or null
    const uint32_t *table_ids;  // [n_gates] or null (table 0)
    // Gates that read the same wire share ONE key switch + modulus switch (reference fbs_mapper/map_to_fbs.py:41-45
    // emits several tables per linear combination): source_of[g] = index of gate g's source in src_slot, null = g.
    const uint32_t *source_of;  // [n_gates] or null
    uint64_t *out_rows;         // non-null: bootstrap f writes row (f - f_begin) of this contiguous array instead of its slot
    uint32_t row_words;         // words per row of out_rows; 0 = one ciphertext (D + 1).  2N for a fused level cut across GPUs:
                                // a shared rotation then leaves its whole accumulator in its row (it travels with the gather)
    // Several tables on ONE blind rotation (fused programs, fbs_program_load_ex): a gate whose dst_slot has bit 31 set is
    // the rotation of the table-independent test vector TV_0 for a source that several tables read; it leaves its whole
    // accumulator in row (dst & 0x7fffffff) * s_count + sample of acc_rows ([row][k + 1][N]) for k_multi_extract.
    uint64_t *acc_rows;
    size_t T;                   // samples per wire in the buffers (stride)
    size_t s_begin, s_count;    // sample window of the launch
    size_t f_begin, count;      // bootstraps of this launch (flattened gate-major over the window)
    size_t ks_begin, ks_count;  // key switches of this launch, flattened the same way over [n_sources][s_count];
                                // row r of the modulus-switched scratch holds flattened source index ks_begin + r
    uint32_t n_gates;
};

struct Profile {
    struct Pending {
        hipEvent_t begin, end;
        std::string kernel;   // instantiation the bracketed launch used
    };
    struct PerKernel {
        double ms = 0;
        uint64_t launches = 0;
    };
    bool on = false;
    std::vector<Pending> pending[3];
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    double ms[3] = {0, 0, 0};
    uint64_t launches[3] = {0, 0, 0};
    std::string kernel[3];   // instantiation of the most recent launch of each kind
    // the same totals per kernel instantiation: a launch that is cut into a whole-round part and a remainder shows as two entries
    std::map<std::string, PerKernel> by_kernel[3];
};

}  // namespace fbs

struct fbs_ctx {
    fbs_params p{};
    uint64_t seed = 0;
    fbs::RandKey rkey{};               // what all randomness of the context is expanded from
    std::atomic<uint64_t> next_nonce{1ull << 55};  // fbs_encrypt_fresh: first unused encryption stream of [2^55, 2^56); reserved by compare-exchange
    fbs::Tune tune;
    int64_t scratch_growths = 0;       // how often a call had to (re)allocate scratch, i.e. blocked (fbs_ctx_stat)
    int device = 0;
    uint32_t N = 0, D = 0, rows = 0;   // D = k*N, rows = (k+1)*l
    uint32_t group = 1;                // key bits per blind-rotation step (1 or 2)
    size_t n_ggsw = 0;                 // GGSW samples in the bootstrapping key: n, or 3n/2 for group 2
    uint32_t ksk_stride = 0;           // padded n+1
    uint64_t delta_half = 0;
    uint64_t g[16]{};                  // round(q / B^(lv+1))
    uint64_t h[64]{};                  // round(q / 2^(gamma (v+1)))
    mutable std::string err;
    std::string devinfo;
    int cu_count = 0;
    hipStream_t stream = nullptr;

    bool have_keys = false;
    std::vector<uint64_t> sk_lwe, sk_glwe, bsk, ksk;   // host copies, standard layout

    uint64_t *d_bsk_hat = nullptr;   // [n][rows][k+1][N]  NTT domain, lane-interleaved, x N^-1
    uint64_t *d_bsk_hat_small = nullptr;   // the same in the evaluation order of the small-launch shape (fbs_ntt.hpp), or null
    uint64_t *d_ksk = nullptr;       // [D*t][ksk_stride]
    uint64_t *d_ksk_f = nullptr;     // the same key as centred doubles (bit patterns), for the FP64 key-switch kernel
    int8_t *d_ks_b = nullptr;        // the key as balanced base-256 limbs in MFMA fragment order (k_ks_gemm, fbs_kernels.hip)
    int8_t *d_ks_a = nullptr;        // scratch: digit fragments of one pass of ciphertexts
    int *d_ks_c = nullptr;           // scratch: limb sums [rows][6][cols_pad], zero between launches
    size_t ks_rows_capacity = 0;
    uint64_t *d_ks_corr = nullptr;   // [ksk_stride]  (B/2) * sum of all key-switching-key rows: balanced digits from unsigned fields
    uint64_t *d_tw_fwd = nullptr;    // [N]  psi^bitrev(i)
    uint64_t *d_tw_inv = nullptr;    // [N]  psi^-bitrev(i)
    uint64_t *d_psi_pow = nullptr;    // [N] psi^x, centred doubles: what the transforms of the monomials X^e - 1 are made from (group 2)
    uint32_t *d_ms = nullptr;        // scratch: mod-switched small ciphertexts [capacity][n+1]
    size_t ms_capacity = 0;
    unsigned long long *d_ms_eps = nullptr;   // [capacity] sums of the mask words' rounding errors (zero between launches)
    uint64_t *d_ms_body = nullptr;            // [capacity] bodies before their rounding (mean-compensated modulus switch)
    uint64_t *d_acc = nullptr;       // scratch: whole accumulators [capacity][2][N] of rotations several tables share
    size_t acc_capacity = 0;         // in rows
    uint64_t *d_stage_in = nullptr, *d_stage_out = nullptr;   // device staging of the host-buffer batch call
    uint32_t *d_stage_ids = nullptr;
    size_t stage_capacity = 0;       // in ciphertexts
    uint32_t *d_idx = nullptr;       // scratch for index arrays of the host-index wires API
    size_t idx_capacity = 0;
    uint64_t *d_wires = nullptr;     // wire slots of fbs_eval, shared by every program of the context
    size_t wires_capacity = 0;       // in words
    // The scratch buffers above are shared by every call on the context.  Calls on ONE stream are ordered by the
    // stream; a call on another stream first waits for the last user of the scratch (scratch_wait / scratch_done).
    hipStream_t scratch_stream = nullptr;
    hipEvent_t scratch_event = nullptr;
    bool scratch_used = false;

    fbs::Profile prof;
};

struct fbs_tvset {
    fbs_ctx *ctx = nullptr;
    uint32_t n_tables = 0;
    uint64_t *d_tvs = nullptr;        // [n_tables + 1][N]: the tables, then TV_0 = delta_half (1 + X + .. + X^(N-1))
    uint64_t *d_post = nullptr;       // [n_tables + 1]
    std::vector<uint64_t> post;       // host copy
    // TV_F = TV_0 * D_F (host_build_tv_diff): the non-zero coefficients of the small integer polynomial D_F per table
    uint32_t diff_cap = 0;            // entries per table in the two arrays below
    uint32_t *d_diff_pos = nullptr;   // [n_tables][diff_cap]
    int32_t *d_diff_val = nullptr;    // [n_tables][diff_cap]
    uint32_t *d_diff_n = nullptr;     // [n_tables]
    std::vector<uint64_t> diff_norm2; // |D_F|^2 and |G_F|^2 (TV_F = delta_half G_F): what fusing does to the output noise
    std::vector<uint64_t> g_norm2;    //   (fbs_table_fusion_norms)
    std::vector<uint8_t> fusable;     // 0: the coefficients of D_F are too large for the 64-bit sums of k_multi_extract
};

namespace fbs {

int set_error(const fbs_ctx *ctx, int code, const std::string &msg);
#define FBS_HIP(ctx, call)                                                                       \
    do {                                                                                         \
        hipError_t e__ = (call);                                                                 \
        if (e__ != hipSuccess)                                                                   \
            return fbs::set_error(ctx, FBS_E_DEVICE, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

// The ABI promises "never throws, never aborts" (include/fbs_exec.h): every extern "C" entry point is a function-try-block that
// ends in FBS_API_CATCH(owner of the error text).  translate_exception re-throws the exception in flight and maps it: an
// allocation the host cannot serve (a std::vector or std::thread sized by caller input) -> FBS_E_NOMEM, anything else ->
// FBS_E_INVALID with its what().  `report` is set_error or the searcher's equivalent; it may itself fail to allocate the text.
template <class Report>
int translate_exception(Report &&report) noexcept {
    int code = FBS_E_INVALID;
    const char *text = "unknown internal error";
    std::string what;
    try {
        throw;
    } catch (const std::bad_alloc &) {
        code = FBS_E_NOMEM, text = "out of host memory";
    } catch (const std::length_error &) {
        code = FBS_E_NOMEM, text = "out of host memory (a size computed from the arguments exceeds what a container can hold)";
    } catch (const std::exception &e) {
        try {
            what = std::string("internal error: ") + e.what();
            text = what.c_str();
        } catch (...) {
        }
    } catch (...) {
    }
    try {
        report(code, text);
    } catch (...) {
    }
    return code;
}
#define FBS_API_CATCH(owner)                                                                                          \
    catch (...) {                                                                                                     \
        const fbs_ctx *owner__ = (owner);                                                                             \
        return fbs::translate_exception([&](int code, const char *text) { fbs::set_error(owner__, code, text); });    \
    }

// host side (fbs_host.cpp)
int host_ctx_init(fbs_ctx *ctx, const fbs_params *params, uint64_t seed, const uint8_t *seed32);   // errors: text in ctx->err
void host_keygen(fbs_ctx *ctx);
void host_encrypt(const fbs_ctx *ctx, const int64_t *msgs, size_t count, uint64_t nonce0, uint64_t *cts);
void host_decrypt(const fbs_ctx *ctx, const uint64_t *cts, size_t count, int64_t *msgs);
int host_build_tv(const fbs_ctx *ctx, const int32_t *table, uint32_t len, uint64_t *tv, uint64_t *post_add);
// D_F with TV_F = TV_0 * D_F as (position, value) pairs of its non-zero coefficients, at most p + 1 of them (`pos`, `val`
// sized for that); *norm2 = |D_F|^2, *g_norm2 = |G_F|^2 (TV_F = delta_half G_F), *abs_sum = sum |d|.  Errors as host_build_tv.
int host_build_tv_diff(const fbs_ctx *ctx, const int32_t *table, uint32_t len, uint32_t *pos, int32_t *val, uint32_t *count,
                       uint64_t *norm2, uint64_t *g_norm2, uint64_t *abs_sum);
void host_twiddles(uint32_t log_n, std::vector<uint64_t> &fwd, std::vector<uint64_t> &inv);

// device side (fbs_kernels.hip); all asynchronous on `stream`
int dev_supported(const fbs_ctx *ctx);   // FBS_OK or error if no kernel instance for the params
bool glwe_shape_built(uint32_t log_n, uint32_t k);   // fbs_blind_rotate_glwe.hip: is there a kernel for GLWE dimension k >= 2 at N = 2^log_n?
int dev_upload_keys(fbs_ctx *ctx);       // BSK -> NTT domain, KSK padded
int dev_keyswitch_gemm_setup(fbs_ctx *ctx);   // limb fragments of the key-switching key for the int8 MFMA key switch
int dev_keyswitch(fbs_ctx *ctx, const GateView &gv, uint32_t *d_ms, hipStream_t stream);
int dev_keyswitch_reserve(fbs_ctx *ctx, size_t count);          // scratch of the GEMM key switch for launches of `count` (blocks when it grows)
int dev_keyswitch_rezero(fbs_ctx *ctx, hipStream_t stream);     // puts its "zero between launches" scratch back after a failed call
int dev_blind_rotate(fbs_ctx *ctx, const fbs_tvset *tv, const GateView &gv, const uint32_t *d_ms, hipStream_t stream);
// `T` = sample stride of the wire buffer, samples [s_begin, s_begin + s_count) are computed
int dev_lincomb(fbs_ctx *ctx, uint64_t *d_wires, size_t T, size_t s_begin, size_t s_count, uint32_t n_out,
                const uint32_t *d_dst, const uint32_t *d_term_off, const uint32_t *d_srcs, const uint64_t *d_coefs,
                const uint64_t *d_consts, hipStream_t stream);
// rows[f - f_begin] -> wire slot dst_slot[f / s_count], sample s_begin + f % s_count, for f in [f_begin, f_begin + count)
// (row_words = 0: rows are ciphertexts; else the row stride, and gates whose dst_slot has bit 31 set -- shared rotations -- are skipped)
int dev_scatter_rows(fbs_ctx *ctx, uint64_t *d_wires, size_t T, size_t s_begin, size_t s_count, const uint32_t *d_dst_slot,
                     const uint64_t *d_rows, size_t f_begin, size_t count, uint32_t row_words, hipStream_t stream);
// rows of `count` ciphertexts: out[i] = wires[slot][s_begin + i] (slot >= 0) or the trivial ciphertext of `body`
int dev_copy_out(fbs_ctx *ctx, const uint64_t *d_wires, size_t T, size_t s_begin, size_t count, int64_t slot, uint64_t body,
                 uint64_t *d_out, hipStream_t stream);
// out[slot x_dst[e]][s] = SampleExtract_0(acc_rows[x_row[e] * s_count + s - s_begin] * D_table) + post, for the n_extract
// tables e that share rotations of TV_0, samples [s_begin, s_begin + s_count)
int dev_multi_extract(fbs_ctx *ctx, const fbs_tvset *tv, const uint64_t *d_acc_rows, uint64_t *d_wires, size_t T, size_t s_begin,
                      size_t s_count, uint32_t n_extract, const uint32_t *d_x_row, const uint32_t *d_x_table,
                      const uint32_t *d_x_dst, hipStream_t stream);
// names of every kernel instantiation the launchers can pick (fbs_kernel_catalog)
void blind_rotate_catalog(std::vector<std::string> *out);
void keyswitch_catalog(std::vector<std::string> *out);
int dev_polymul(fbs_ctx *ctx, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_c, hipStream_t stream);

// profiling helpers
void prof_begin(fbs_ctx *ctx, int which, hipStream_t s, hipEvent_t *e0, hipEvent_t *e1);
void prof_end(fbs_ctx *ctx, int which, hipStream_t s, hipEvent_t e0, hipEvent_t e1);

}  // namespace fbs

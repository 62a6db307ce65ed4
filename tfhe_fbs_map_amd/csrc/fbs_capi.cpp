// extern "C" surface of libfbsexec.so (declared in include/fbs_exec.h) and the level-scheduled
// program executor that stands behind `LutExecEnv.eval` (reference fbs_mapper/fbs_exec_env.py:208-229).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>

#include "fbs_internal.hpp"
#include "fbs_plan.hpp"

using namespace fbs;

static thread_local std::string g_create_error;

namespace fbs {
int set_error(const fbs_ctx *ctx, int code, const std::string &msg) {
    if (ctx) ctx->err = msg;
    else g_create_error = msg;
    return code;
}
}  // namespace fbs

// ---------------------------------------------------------------------------------------------
// program representation
// ---------------------------------------------------------------------------------------------
struct LincombStage {
    uint32_t n_out = 0;
    uint32_t *d_dst = nullptr, *d_term_off = nullptr, *d_srcs = nullptr;   // wire SLOTS
    uint64_t *d_coefs = nullptr, *d_consts = nullptr;
};
// Bootstraps of one level, sorted by source wire.  Gates that read the same wire (the reference's one-gate-one-bootstrap
// lowering emits several tables per linear combination, fbs_mapper/map_to_fbs.py:41-45, and its CSE only merges
// identical tables, fbs_mapper/fbs_exec_env.py:93-100) share one key switch + modulus switch.
struct BootStage {
    uint32_t n_gates = 0, n_sources = 0;
    uint32_t *d_src_slot = nullptr;    // [n_sources] wire slot of each distinct source
    uint32_t *d_source_of = nullptr;   // [n_gates]   index into d_src_slot
    uint32_t *d_dst = nullptr, *d_table = nullptr;   // [n_gates]
    std::vector<uint32_t> source_of;   // host copy: which key switches a slice of the level needs
    // Fused programs (FBS_LOAD_FUSE_TABLES): the tables of a source that several read are served by ONE gate of the list
    // above -- the rotation of TV_0 (table id = the set's n_tables, dst = 0x80000000 | shared index) -- and one entry each
    // of the extraction list below (k_multi_extract)
    uint32_t n_shared = 0, n_extract = 0;
    uint32_t *d_x_row = nullptr, *d_x_table = nullptr, *d_x_dst = nullptr;   // [n_extract] shared index, table, wire slot
    uint32_t *d_x_gate = nullptr;      // [n_extract] position of the shared rotation in the gate list (a level cut across GPUs
                                       // finds the accumulator in that gate's gathered row)
};
struct fbs_prog {
    fbs_ctx *ctx = nullptr;
    const fbs_tvset *tv = nullptr;
    uint32_t n_inputs = 0, n_instr = 0, n_outputs = 0, n_wires = 0, n_slots = 0;
    uint32_t depth = 0, max_width = 0, max_sources = 0, n_bootstrap = 0, n_keyswitch = 0;
    uint32_t n_rotations = 0, max_shared = 0;   // blind rotations per sample (= n_bootstrap unless fused); shared rotations of the widest level
    bool fused = false;
    std::vector<uint32_t> in_slot;    // [n_inputs]
    std::vector<int64_t> out_slot;    // [n_outputs]  slot, or -1-c for the constant c
    // schedule: for level L = 0..depth: lincomb stages (dependency order), then the bootstraps of level L+1
    std::vector<std::vector<LincombStage>> lin;   // [depth+1][sub]
    std::vector<BootStage> boot;                  // [depth]  (boot[L] = bootstraps of level L+1)
    std::vector<void *> allocations;
};

template <typename T>
static int to_device(fbs_ctx *ctx, fbs_prog *prog, const std::vector<T> &v, T **out) {
    *out = nullptr;
    if (v.empty()) return FBS_OK;
    void *d = nullptr;
    FBS_HIP(ctx, hipMalloc(&d, v.size() * sizeof(T)));
    prog->allocations.push_back(d);
    FBS_HIP(ctx, hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (T *)d;
    return FBS_OK;
}

// a LinearProd coefficient as the kernel wants it: centred residue, as the bit pattern of a double
static uint64_t coef_bits(int64_t c) {
    const double d = fq_centered(fq_from_i64(c));
    uint64_t u;
    std::memcpy(&u, &d, sizeof u);
    return u;
}

static hipStream_t pick(fbs_ctx *ctx, void *stream) { return stream ? (hipStream_t)stream : ctx->stream; }

// Scratch is grown on demand, and growing it BLOCKS (the old buffers may still be read by queued kernels): a host that wants
// its *_dev calls to be nothing but kernel launches sizes everything up front with fbs_ctx_reserve.
static int ensure_ms(fbs_ctx *ctx, size_t count) {
    if (int rc = dev_keyswitch_reserve(ctx, count)) return rc;   // (the int8-GEMM key switch's digit and limb-sum scratch)
    if (count <= ctx->ms_capacity) return FBS_OK;
    ctx->scratch_growths++;
    if (ctx->scratch_used) FBS_HIP(ctx, hipStreamSynchronize(ctx->scratch_stream));   // kernels may still read the old buffer
    for (void *p : {(void *)ctx->d_ms, (void *)ctx->d_ms_eps, (void *)ctx->d_ms_body})
        if (p) (void)hipFree(p);
    ctx->d_ms = nullptr;
    ctx->d_ms_eps = nullptr;
    ctx->d_ms_body = nullptr;
    ctx->ms_capacity = 0;
    FBS_HIP(ctx, hipMalloc(&ctx->d_ms, count * (ctx->p.n + 1) * sizeof(uint32_t)));
    FBS_HIP(ctx, hipMalloc(&ctx->d_ms_eps, count * 8));
    FBS_HIP(ctx, hipMalloc(&ctx->d_ms_body, count * 8));
    FBS_HIP(ctx, hipMemset(ctx->d_ms_eps, 0, count * 8));   // every launch leaves it zero again (k_ms_body)
    FBS_HIP(ctx, hipDeviceSynchronize());
    ctx->ms_capacity = count;
    return FBS_OK;
}

static int ensure_acc(fbs_ctx *ctx, size_t rows) {
    if (rows <= ctx->acc_capacity) return FBS_OK;
    ctx->scratch_growths++;
    if (ctx->scratch_used) FBS_HIP(ctx, hipStreamSynchronize(ctx->scratch_stream));
    if (ctx->d_acc) (void)hipFree(ctx->d_acc);
    ctx->d_acc = nullptr;
    ctx->acc_capacity = 0;
    FBS_HIP(ctx, hipMalloc(&ctx->d_acc, rows * (size_t)(ctx->p.k + 1) * ctx->N * 8));   // a row = a whole GLWE accumulator
    ctx->acc_capacity = rows;
    return FBS_OK;
}

static int ensure_wires(fbs_ctx *ctx, size_t words) {
    if (words <= ctx->wires_capacity) return FBS_OK;
    ctx->scratch_growths++;
    if (ctx->scratch_used) FBS_HIP(ctx, hipStreamSynchronize(ctx->scratch_stream));
    if (ctx->d_wires) (void)hipFree(ctx->d_wires);
    ctx->d_wires = nullptr;
    ctx->wires_capacity = 0;
    FBS_HIP(ctx, hipMalloc(&ctx->d_wires, words * 8));
    ctx->wires_capacity = words;
    return FBS_OK;
}

// Cross-stream ordering of the per-context scratch (d_ms, d_idx, d_wires): a call on stream `s` first waits for the last
// call that used the scratch on ANOTHER stream; calls on one stream are ordered by the stream itself.
static int scratch_wait(fbs_ctx *ctx, hipStream_t s) {
    if (ctx->scratch_used && ctx->scratch_stream != s) FBS_HIP(ctx, hipStreamWaitEvent(s, ctx->scratch_event, 0));
    return FBS_OK;
}
static int scratch_done(fbs_ctx *ctx, hipStream_t s) {
    FBS_HIP(ctx, hipEventRecord(ctx->scratch_event, s));
    ctx->scratch_stream = s;
    ctx->scratch_used = true;
    return FBS_OK;
}

// A call that fails between scratch_wait and scratch_done may have left the "zero between launches" scratch (rounding-error
// sums, limb sums of the GEMM key switch) half used: put it back, so that the next call on the context starts clean.
static int scratch_fail(fbs_ctx *ctx, hipStream_t s, int rc) {
    if (ctx->d_ms_eps && ctx->ms_capacity) (void)hipMemsetAsync(ctx->d_ms_eps, 0, ctx->ms_capacity * 8, s);
    (void)dev_keyswitch_rezero(ctx, s);
    (void)scratch_done(ctx, s);
    return rc;
}

// the plain batch: gate g = ciphertext g, one sample each
static GateView batch_view(const uint64_t *in, uint64_t *out, const uint32_t *table_ids, size_t count) {
    GateView gv{};
    gv.in_base = in;
    gv.out_base = out;
    gv.table_ids = table_ids;
    gv.T = 1;
    gv.s_begin = 0;
    gv.s_count = 1;
    gv.f_begin = 0;
    gv.count = count;
    gv.ks_begin = 0;
    gv.ks_count = count;
    gv.n_gates = (uint32_t)count;
    return gv;
}

extern "C" {

// ---------------------------------------------------------------------------------------------
int fbs_poly_size_check(uint32_t poly_size) try {
    if (poly_size == 0) return set_error(nullptr, FBS_E_INVALID, "polynomial size 0");
    if (poly_size & (poly_size - 1)) {
        uint32_t pow2 = poly_size & (~poly_size + 1);   // largest power of two dividing N
        return set_error(nullptr, FBS_E_POLY_SIZE,
                         "N = " + std::to_string(poly_size) + " is not a power of two: X^N + 1 then has the factor X^" +
                             std::to_string(pow2) + " + 1, so a GLWE sample over it is no harder than one of degree " +
                             std::to_string(pow2) + "; use a power-of-two N (256 .. 4096) and any plaintext modulus p");
    }
    if (poly_size < 256 || poly_size > 4096)
        return set_error(nullptr, FBS_E_INVALID, "supported polynomial sizes are N = 256, 512, 1024, 2048, 4096");
    return FBS_OK;
} FBS_API_CATCH(nullptr)

static int64_t env_knob(const char *name, int64_t dflt) {
    const char *e = getenv(name);
    return e && *e ? atoll(e) : dflt;
}

static int ctx_create(const fbs_params *params, uint64_t seed, const uint8_t *seed32, int device, fbs_ctx **out) {
    if (!params || !out) return set_error(nullptr, FBS_E_INVALID, "null argument");
    *out = nullptr;
    std::unique_ptr<fbs_ctx> ctx(new fbs_ctx);
    // everything that is arithmetic on the parameter set (ranges, derived sizes, Delta, gadget factors, the random key): host code
    // with no device in it (fbs_host.cpp: the sanitizer harness of tests/c/ runs the same function)
    int rc = host_ctx_init(ctx.get(), params, seed, seed32);
    if (rc != FBS_OK) return set_error(nullptr, rc, ctx->err);
    ctx->device = device;
    // A/B switches of the launchers, settable per context with fbs_ctx_tune; the environment gives the defaults of a process
    ctx->tune.ks_gemm_min = env_knob("FBS_KS_GEMM_MIN", ctx->tune.ks_gemm_min);
    ctx->tune.ks_mfma = env_knob("FBS_KS_NO_MFMA", 0) ? 0 : 1;
    ctx->tune.ks_fp = env_knob("FBS_KS_INTEGER", 0) ? 0 : 1;
    ctx->tune.ks_cols_major = env_knob("FBS_KS_TILES_MAJOR", 0) ? 0 : 1;
    ctx->tune.ks_split = env_knob("FBS_KS_SPLIT", 0);
    ctx->tune.br_whole_cu = getenv("FBS_BR_SMALL_WORKGROUPS") ? 0 : 1;
    ctx->tune.br_cu_kernel = getenv("FBS_BR_NO_CU_KERNEL") ? 0 : 1;
    ctx->tune.br_cu_max_per_cu = env_knob("FBS_BR_CU_MAX_PER_CU", ctx->tune.br_cu_max_per_cu);
    ctx->tune.br_cu_lean = env_knob("FBS_BR_CU_LEAN", ctx->tune.br_cu_lean);
    ctx->tune.br_k2_shape = env_knob("FBS_BR_K2_SHAPE", ctx->tune.br_k2_shape);
    ctx->tune.br_glwe_fpw = env_knob("FBS_BR_GLWE_FPW", ctx->tune.br_glwe_fpw);
    rc = dev_supported(ctx.get());   // (is there a kernel instantiation for this shape?)
    if (rc != FBS_OK) return set_error(nullptr, rc, ctx->err);

    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0)
        return set_error(nullptr, FBS_E_DEVICE, "no HIP device: libfbsexec has no CPU path (" + std::string(hipGetErrorString(e)) + ")");
    if (device < 0 || device >= n_dev) return set_error(nullptr, FBS_E_INVALID, "device ordinal out of range");
    e = hipSetDevice(device);
    if (e != hipSuccess) return set_error(nullptr, FBS_E_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) return set_error(nullptr, FBS_E_DEVICE, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
    if (std::string(prop.gcnArchName).rfind("gfx950", 0) != 0)
        return set_error(nullptr, FBS_E_DEVICE, std::string("device is ") + prop.gcnArchName + "; this library carries gfx950 code only");
    ctx->cu_count = prop.multiProcessorCount;
    ctx->devinfo = std::string(prop.gcnArchName) + " " + prop.name + " CUs=" + std::to_string(prop.multiProcessorCount);
    // a blocking stream: ordered with the legacy null stream, which is what PyTorch's default stream is --
    // a caller that passes stream = NULL while using torch tensors still gets correct ordering
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamDefault);
    if (e != hipSuccess) return set_error(nullptr, FBS_E_DEVICE, std::string("hipStreamCreate: ") + hipGetErrorString(e));
    e = hipEventCreateWithFlags(&ctx->scratch_event, hipEventDisableTiming);
    if (e != hipSuccess) {
        (void)hipStreamDestroy(ctx->stream);
        return set_error(nullptr, FBS_E_DEVICE, std::string("hipEventCreate: ") + hipGetErrorString(e));
    }
    *out = ctx.release();
    return FBS_OK;
}

int fbs_ctx_create(const fbs_params *params, uint64_t seed, int device, fbs_ctx **out) try {
    return ctx_create(params, seed, nullptr, device, out);
} FBS_API_CATCH(nullptr)

int fbs_ctx_create_seeded(const fbs_params *params, const uint8_t seed[32], int device, fbs_ctx **out) try {
    if (!seed) return set_error(nullptr, FBS_E_INVALID, "null seed");
    return ctx_create(params, 0, seed, device, out);
} FBS_API_CATCH(nullptr)

int fbs_ctx_tune(fbs_ctx *ctx, const char *knob, int64_t value) try {
    if (!ctx || !knob) return FBS_E_INVALID;
    const std::string k(knob);
    Tune &t = ctx->tune;
    int64_t *slot = k == "ks_gemm_min" ? &t.ks_gemm_min : k == "ks_mfma" ? &t.ks_mfma : k == "ks_fp" ? &t.ks_fp :
                    k == "ks_cols_major" ? &t.ks_cols_major : k == "ks_split" ? &t.ks_split : k == "br_whole_cu" ? &t.br_whole_cu :
                    k == "br_cu_kernel" ? &t.br_cu_kernel : k == "br_cu_max_per_cu" ? &t.br_cu_max_per_cu : k == "br_cu_lean" ? &t.br_cu_lean :
                    k == "br_k2_shape" ? &t.br_k2_shape : k == "br_glwe_fpw" ? &t.br_glwe_fpw : nullptr;
    if (!slot) return set_error(ctx, FBS_E_INVALID, "unknown knob '" + k + "'");
    if (value < 0) return set_error(ctx, FBS_E_INVALID, "knob values are non-negative");
    *slot = value;
    return FBS_OK;
} FBS_API_CATCH(ctx)

int fbs_ctx_stat(const fbs_ctx *ctx, const char *name, int64_t *value) try {
    if (!ctx || !name || !value) return FBS_E_INVALID;
    const std::string k(name);
    if (k == "scratch_growths") *value = ctx->scratch_growths;
    else if (k == "ms_capacity") *value = (int64_t)ctx->ms_capacity;
    else if (k == "acc_capacity") *value = (int64_t)ctx->acc_capacity;
    else if (k == "wires_capacity") *value = (int64_t)ctx->wires_capacity;
    else if (k == "next_nonce") *value = (int64_t)ctx->next_nonce.load();
    else if (k == "cu_count") *value = ctx->cu_count;
    else return set_error(ctx, FBS_E_INVALID, "unknown statistic '" + k + "'");
    return FBS_OK;
} FBS_API_CATCH(ctx)

int fbs_ctx_reserve(fbs_ctx *ctx, size_t max_keyswitches, size_t max_shared_rows, size_t wire_words) try {
    if (!ctx) return FBS_E_INVALID;
    FBS_HIP(ctx, hipSetDevice(ctx->device));
    int rc;
    if (max_keyswitches && (rc = ensure_ms(ctx, max_keyswitches)) != FBS_OK) return rc;
    if (max_shared_rows && (rc = ensure_acc(ctx, max_shared_rows)) != FBS_OK) return rc;
    if (wire_words && (rc = ensure_wires(ctx, wire_words)) != FBS_OK) return rc;
    return FBS_OK;
} FBS_API_CATCH(ctx)

void fbs_ctx_destroy(fbs_ctx *ctx) try {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->scratch_used) (void)hipStreamSynchronize(ctx->scratch_stream);
    for (void *p : {(void *)ctx->d_bsk_hat, (void *)ctx->d_bsk_hat_small, (void *)ctx->d_ksk, (void *)ctx->d_ksk_f, (void *)ctx->d_ks_corr, (void *)ctx->d_ks_a, (void *)ctx->d_ks_b, (void *)ctx->d_ks_c, (void *)ctx->d_tw_fwd, (void *)ctx->d_tw_inv, (void *)ctx->d_psi_pow, (void *)ctx->d_ms, (void *)ctx->d_ms_eps, (void *)ctx->d_ms_body, (void *)ctx->d_acc, (void *)ctx->d_stage_in, (void *)ctx->d_stage_out, (void *)ctx->d_stage_ids,
                    (void *)ctx->d_idx, (void *)ctx->d_wires})
        if (p) (void)hipFree(p);
    if (ctx->scratch_event) (void)hipEventDestroy(ctx->scratch_event);
    for (auto &v : ctx->prof.pending)
        for (auto &pr : v) {
            (void)hipEventDestroy(pr.begin);
            (void)hipEventDestroy(pr.end);
        }
    for (auto &pr : ctx->prof.pool) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
} catch (...) {   // (nothing here allocates; the promise of the ABI is kept anyway)
}

const char *fbs_last_error(const fbs_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }
const char *fbs_device_info(const fbs_ctx *ctx) { return ctx ? ctx->devinfo.c_str() : ""; }

// ---------------------------------------------------------------------------------------------
int fbs_keygen(fbs_ctx *ctx) try {
    if (!ctx) return FBS_E_INVALID;
    FBS_HIP(ctx, hipSetDevice(ctx->device));
    host_keygen(ctx);
    int rc = dev_upload_keys(ctx);
    if (rc != FBS_OK) return rc;
    ctx->have_keys = true;
    return FBS_OK;
} FBS_API_CATCH(ctx)

int fbs_key_sizes(const fbs_ctx *ctx, size_t sizes[4]) try {
    if (!ctx || !sizes) return FBS_E_INVALID;
    sizes[0] = ctx->p.n;
    sizes[1] = ctx->D;
    sizes[2] = ctx->n_ggsw * ctx->rows * (ctx->p.k + 1) * ctx->N;
    sizes[3] = (size_t)ctx->D * ctx->p.t_ksk * (ctx->p.n + 1);
    return FBS_OK;
} FBS_API_CATCH(ctx)

int fbs_export_keys(const fbs_ctx *ctx, uint64_t *sk_lwe, uint64_t *sk_glwe, uint64_t *bsk, uint64_t *ksk) try {
    if (!ctx) return FBS_E_INVALID;
    if (!ctx->have_keys) return set_error(ctx, FBS_E_STATE, "fbs_keygen has not run");
    if (sk_lwe) std::memcpy(sk_lwe, ctx->sk_lwe.data(), ctx->sk_lwe.size() * 8);
    if (sk_glwe) std::memcpy(sk_glwe, ctx->sk_glwe.data(), ctx->sk_glwe.size() * 8);
    if (bsk) std::memcpy(bsk, ctx->bsk.data(), ctx->bsk.size() * 8);
    if (ksk) std::memcpy(ksk, ctx->ksk.data(), ctx->ksk.size() * 8);
    return FBS_OK;
} FBS_API_CATCH(ctx)

// Do the evaluation keys decrypt under the secrets they came with?  A handful of GGSW samples (every row) and key-switching rows,
// each phase compared with what the layout of fbs_key_sizes says it encrypts: a key in another sample / row / column order has
// uniform phases and fails here instead of bootstrapping to garbage.  Tolerance: 16 standard deviations of the set's noise.
static const char *imported_keys_mismatch(const fbs_ctx *ctx, const uint64_t *sk_lwe, const uint64_t *sk_glwe, const uint64_t *bsk,
                                          const uint64_t *ksk) {
    const fbs_params &p = ctx->p;
    const uint32_t N = ctx->N, D = ctx->D, n = p.n, k = p.k, l = p.l_bsk, t = p.t_ksk, rows = ctx->rows;
    auto far = [](uint64_t got, uint64_t want, double tol) { return std::fabs(fq_centered(fq_sub(got, want))) > tol; };
    const double tol_glwe = 1024.0 + 16.0 * (double)p.sigma_glwe, tol_lwe = 1024.0 + 16.0 * (double)p.sigma_lwe;
    std::vector<size_t> samples = {0, 1, 2, ctx->n_ggsw / 2, ctx->n_ggsw - 1};
    std::vector<uint64_t> phase(N);
    for (size_t g : samples) {
        if (g >= ctx->n_ggsw) continue;
        uint64_t bit = sk_lwe[g];
        if (ctx->group == 2) {
            const uint64_t s0 = sk_lwe[2 * (g / 3)], s1 = sk_lwe[2 * (g / 3) + 1];
            bit = g % 3 == 0 ? (s0 & (1 - s1)) : g % 3 == 1 ? ((1 - s0) & s1) : (s0 & s1);
        }
        for (uint32_t rr = 0; rr < rows; rr++) {
            const uint32_t comp = rr / l, lv = rr % l;
            const uint64_t *row = bsk + (g * rows + rr) * (size_t)(k + 1) * N;
            for (uint32_t j = 0; j < N; j++) phase[j] = row[(size_t)k * N + j];
            for (uint32_t c = 0; c < k; c++)                            // phase -= A_c * S_c (negacyclic, binary S)
                for (uint32_t sh = 0; sh < N; sh++) {
                    if (!sk_glwe[(size_t)c * N + sh]) continue;
                    const uint64_t *a = row + (size_t)c * N;
                    for (uint32_t j = 0; j < N - sh; j++) phase[j + sh] = fq_sub(phase[j + sh], a[j]);
                    for (uint32_t j = N - sh; j < N; j++) phase[j + sh - N] = fq_add(phase[j + sh - N], a[j]);
                }
            // row (comp, lv) = GLWE(0) + bit g_lv on component comp: the phase is bit g_lv at X^0 (body row), -bit g_lv S_comp (mask rows)
            for (uint32_t j = 0; j < N; j++) {
                uint64_t want = 0;
                if (bit && comp == k && j == 0) want = ctx->g[lv];
                if (bit && comp < k && sk_glwe[(size_t)comp * N + j]) want = fq_sub(0, ctx->g[lv]);
                if (far(phase[j], want, tol_glwe)) return "bootstrapping key does not decrypt under the supplied secrets (sample / row / column order of fbs_key_sizes?)";
            }
        }
    }
    const size_t ksk_rows = (size_t)D * t;
    for (size_t r : {(size_t)0, (size_t)1, (size_t)2, (size_t)3, ksk_rows / 2, ksk_rows - 4, ksk_rows - 3, ksk_rows - 2, ksk_rows - 1}) {
        if (r >= ksk_rows) continue;
        const uint32_t j = (uint32_t)(r / t), v = (uint32_t)(r % t);
        const uint64_t *row = ksk + r * (size_t)(n + 1);
        uint64_t ph = row[n];
        for (uint32_t i = 0; i < n; i++)
            if (sk_lwe[i]) ph = fq_sub(ph, row[i]);
        if (far(ph, sk_glwe[j] ? ctx->h[v] : 0, tol_lwe)) return "key-switching key does not decrypt under the supplied secrets (row order [kN][t][n+1]?)";
    }
    return nullptr;
}

int fbs_import_keys(fbs_ctx *ctx, const uint64_t *sk_lwe, const uint64_t *sk_glwe, const uint64_t *bsk, const uint64_t *ksk) try {
    if (!ctx) return FBS_E_INVALID;
    if (!sk_lwe || !sk_glwe || !bsk || !ksk) return set_error(ctx, FBS_E_INVALID, "null argument");
    FBS_HIP(ctx, hipSetDevice(ctx->device));
    size_t sizes[4];
    fbs_key_sizes(ctx, sizes);
    for (size_t i = 0; i < sizes[0]; i++)
        if (sk_lwe[i] > 1) return set_error(ctx, FBS_E_INVALID, "secret keys are binary");
    for (size_t i = 0; i < sizes[1]; i++)
        if (sk_glwe[i] > 1) return set_error(ctx, FBS_E_INVALID, "secret keys are binary");
    for (size_t i = 0; i < sizes[2]; i++)
        if (bsk[i] >= FQ) return set_error(ctx, FBS_E_INVALID, "bootstrapping-key word is not a canonical residue");
    for (size_t i = 0; i < sizes[3]; i++)
        if (ksk[i] >= FQ) return set_error(ctx, FBS_E_INVALID, "key-switching-key word is not a canonical residue");
    if (const char *why = imported_keys_mismatch(ctx, sk_lwe, sk_glwe, bsk, ksk)) return set_error(ctx, FBS_E_INVALID, why);
    if (ctx->scratch_used) FBS_HIP(ctx, hipStreamSynchronize(ctx->scratch_stream));   // kernels may still read the old keys
    ctx->sk_lwe.assign(sk_lwe, sk_lwe + sizes[0]);
    ctx->sk_glwe.assign(sk_glwe, sk_glwe + sizes[1]);
    ctx->bsk.assign(bsk, bsk + sizes[2]);
    ctx->ksk.assign(ksk, ksk + sizes[3]);
    ctx->have_keys = false;
    int rc = dev_upload_keys(ctx);
    if (rc != FBS_OK) return rc;
    ctx->have_keys = true;
    return FBS_OK;
} FBS_API_CATCH(ctx)

int fbs_encrypt_fresh(fbs_ctx *ctx, const int64_t *msgs, size_t count, uint64_t *cts, uint64_t *nonce0) try {
    if (!ctx || (count && (!msgs || !cts))) return FBS_E_INVALID;
    if (!ctx->have_keys) return set_error(ctx, FBS_E_STATE, "fbs_keygen has not run");
    // the range [first, first + count) is reserved atomically: two threads encrypting on one context never share a stream (the
    // bound is checked BEFORE the counter moves, so a refused call leaves it where it was)
    uint64_t first = ctx->next_nonce.load(std::memory_order_relaxed);
    do {
        if (count > (1ull << 56) || first + count > (1ull << 56)) return set_error(ctx, FBS_E_STATE, "encryption streams of this context are used up");
    } while (!ctx->next_nonce.compare_exchange_weak(first, first + count, std::memory_order_relaxed));
    if (nonce0) *nonce0 = first;
    host_encrypt(ctx, msgs, count, first, cts);
    return FBS_OK;
} FBS_API_CATCH(ctx)

int fbs_encrypt(const fbs_ctx *ctx, const int64_t *msgs, size_t count, uint64_t nonce0, uint64_t *cts) try {
    if (!ctx || (count && (!msgs || !cts))) return FBS_E_INVALID;
    if (!ctx->have_keys) return set_error(ctx, FBS_E_STATE, "fbs_keygen has not run");
    // streams [2^55, 2^56) belong to fbs_encrypt_fresh: an explicit nonce can never repeat one the context handed out itself
    if (nonce0 >= (1ull << 55) || count > (1ull << 55) - nonce0) return set_error(ctx, FBS_E_INVALID, "nonce0 + count must stay below 2^55");
    host_encrypt(ctx, msgs, count, nonce0, cts);
    return FBS_OK;
} FBS_API_CATCH(ctx)

int fbs_decrypt(const fbs_ctx *ctx, const uint64_t *cts, size_t count, int64_t *msgs) try {
    if (!ctx || (count && (!msgs || !cts))) return FBS_E_INVALID;
    if (!ctx->have_keys) return set_error(ctx, FBS_E_STATE, "fbs_keygen has not run");
    host_decrypt(ctx, cts, count, msgs);
    return FBS_OK;
} FBS_API_CATCH(ctx)

// ---------------------------------------------------------------------------------------------
int fbs_tvset_create(fbs_ctx *ctx, const int32_t *table_vals, const uint32_t *table_off, uint32_t n_tables, fbs_tvset **out) try {
    if (!ctx || !out || (n_tables && (!table_vals || !table_off))) return FBS_E_INVALID;
    *out = nullptr;
    // the count is checked BEFORE anything is sized by it: more than 2^20 tables (8 GB of test vectors at N = 1024) is a
    // corrupted count, not a program
    if (n_tables > FBS_MAX_TABLES) return set_error(ctx, FBS_E_INVALID, "more than FBS_MAX_TABLES tables");
    FBS_HIP(ctx, hipSetDevice(ctx->device));
    std::unique_ptr<fbs_tvset> tv(new fbs_tvset);
    tv->ctx = ctx;
    tv->n_tables = n_tables;
    const uint32_t N = ctx->N;
    // the tables, then TV_0 (index n_tables): what a rotation shared by several tables starts from
    std::vector<uint64_t> host((size_t)(n_tables + 1) * N, 0);
    tv->post.assign((size_t)n_tables + 1, 0);
    tv->diff_cap = ctx->p.p_msg + 1;
    std::vector<uint32_t> dpos((size_t)std::max(1u, n_tables) * tv->diff_cap, 0), dn(std::max(1u, n_tables), 0);
    std::vector<int32_t> dval((size_t)std::max(1u, n_tables) * tv->diff_cap, 0);
    tv->diff_norm2.assign(n_tables, 0);
    tv->g_norm2.assign(n_tables, 0);
    tv->fusable.assign(n_tables, 0);
    for (uint32_t t = 0; t < n_tables; t++) {
        if (table_off[t + 1] < table_off[t]) return set_error(ctx, FBS_E_INVALID, "table offsets must be non-decreasing");
        const int32_t *vals = table_vals + table_off[t];
        const uint32_t len = table_off[t + 1] - table_off[t];
        int rc = host_build_tv(ctx, vals, len, host.data() + (size_t)t * N, &tv->post[t]);
        uint64_t abs_sum = 0;
        if (rc == FBS_OK)
            rc = host_build_tv_diff(ctx, vals, len, dpos.data() + (size_t)t * tv->diff_cap, dval.data() + (size_t)t * tv->diff_cap, &dn[t],
                                    &tv->diff_norm2[t], &tv->g_norm2[t], &abs_sum);
        if (rc != FBS_OK)
            return set_error(ctx, rc, "table " + std::to_string(t) + " of length " + std::to_string(len) +
                                          " is not evaluable by one bootstrap at p = " + std::to_string(ctx->p.p_msg));
        tv->fusable[t] = abs_sum < (1ull << 16);   // k_multi_extract sums d * word (< 2^46) in 64 bits
    }
    for (uint32_t j = 0; j < N; j++) host[(size_t)n_tables * N + j] = ctx->delta_half;
    hipError_t e = hipMalloc(&tv->d_tvs, host.size() * 8);
    if (e == hipSuccess) e = hipMalloc(&tv->d_post, tv->post.size() * 8);
    if (e == hipSuccess) e = hipMalloc(&tv->d_diff_pos, dpos.size() * 4);
    if (e == hipSuccess) e = hipMalloc(&tv->d_diff_val, dval.size() * 4);
    if (e == hipSuccess) e = hipMalloc(&tv->d_diff_n, dn.size() * 4);
    if (e == hipSuccess) e = hipMemcpy(tv->d_tvs, host.data(), host.size() * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(tv->d_post, tv->post.data(), tv->post.size() * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(tv->d_diff_pos, dpos.data(), dpos.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(tv->d_diff_val, dval.data(), dval.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(tv->d_diff_n, dn.data(), dn.size() * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        fbs_tvset_destroy(tv.release());   // frees whichever buffer was obtained
        return set_error(ctx, FBS_E_DEVICE, std::string("test-vector upload: ") + hipGetErrorString(e));
    }
    *out = tv.release();
    return FBS_OK;
} FBS_API_CATCH(ctx)

int fbs_table_fusion_norms(const fbs_tvset *tv, uint32_t table, uint64_t *d_norm2, uint64_t *g_norm2) try {
    if (!tv || table >= tv->n_tables) return FBS_E_INVALID;
    if (d_norm2) *d_norm2 = tv->diff_norm2[table];
    if (g_norm2) *g_norm2 = tv->g_norm2[table];
    return FBS_OK;
} FBS_API_CATCH(tv ? tv->ctx : nullptr)

void fbs_tvset_destroy(fbs_tvset *tv) try {
    if (!tv) return;
    if (tv->ctx) (void)hipSetDevice(tv->ctx->device);
    if (tv->d_tvs) (void)hipFree(tv->d_tvs);
    if (tv->d_post) (void)hipFree(tv->d_post);
    if (tv->d_diff_pos) (void)hipFree(tv->d_diff_pos);
    if (tv->d_diff_val) (void)hipFree(tv->d_diff_val);
    if (tv->d_diff_n) (void)hipFree(tv->d_diff_n);
    delete tv;
} catch (...) {   // (nothing here allocates; the promise of the ABI is kept anyway)
}

// ---------------------------------------------------------------------------------------------
static int check_ready(fbs_ctx *ctx, const fbs_tvset *tv) {
    if (!ctx) return FBS_E_INVALID;
    if (!ctx->have_keys) return set_error(ctx, FBS_E_STATE, "fbs_keygen has not run");
    if (tv && tv->ctx != ctx) return set_error(ctx, FBS_E_INVALID, "test-vector set belongs to another context");
    FBS_HIP(ctx, hipSetDevice(ctx->device));
    return FBS_OK;
}

int fbs_bootstrap_batch_dev(fbs_ctx *ctx, const fbs_tvset *tv, const uint64_t *d_cts_in, const uint32_t *d_table_ids,
                            size_t count, uint64_t *d_cts_out, void *stream) try {
    int rc = check_ready(ctx, tv);
    if (rc != FBS_OK) return rc;
    if (!tv || (count && (!d_cts_in || !d_cts_out))) return set_error(ctx, FBS_E_INVALID, "null argument");
    if (count == 0) return FBS_OK;
    if (count > 0x7FFFFFFFull) return set_error(ctx, FBS_E_INVALID, "batch too large");
    rc = ensure_ms(ctx, count);
    if (rc != FBS_OK) return rc;
    const GateView gv = batch_view(d_cts_in, d_cts_out, d_table_ids, count);
    hipStream_t s = pick(ctx, stream);
    if ((rc = scratch_wait(ctx, s)) != FBS_OK) return rc;
    rc = dev_keyswitch(ctx, gv, ctx->d_ms, s);
    if (rc != FBS_OK) return scratch_fail(ctx, s, rc);
    rc = dev_blind_rotate(ctx, tv, gv, ctx->d_ms, s);
    if (rc != FBS_OK) return scratch_fail(ctx, s, rc);
    return scratch_done(ctx, s);
} FBS_API_CATCH(ctx)

int fbs_bootstrap_batch(fbs_ctx *ctx, const fbs_tvset *tv, const uint64_t *cts_in, const uint32_t *table_ids, size_t count,
                        uint64_t *cts_out) try {
    int rc = check_ready(ctx, tv);
    if (rc != FBS_OK) return rc;
    if (!tv || (count && (!cts_in || !cts_out))) return set_error(ctx, FBS_E_INVALID, "null argument");
    if (count == 0) return FBS_OK;
    if (table_ids)
        for (size_t i = 0; i < count; i++)
            if (table_ids[i] >= tv->n_tables) return set_error(ctx, FBS_E_INVALID, "table id out of range");
    // device staging owned by the context (grown on demand, reused from call to call: no allocation on the steady path)
    const size_t words = count * (ctx->D + 1);
    if (ctx->stage_capacity < count) {
        if (ctx->scratch_used) FBS_HIP(ctx, hipStreamSynchronize(ctx->scratch_stream));
        for (void *p : {(void *)ctx->d_stage_in, (void *)ctx->d_stage_out, (void *)ctx->d_stage_ids})
            if (p) (void)hipFree(p);
        ctx->d_stage_in = ctx->d_stage_out = nullptr;
        ctx->d_stage_ids = nullptr;
        ctx->stage_capacity = 0;
        FBS_HIP(ctx, hipMalloc(&ctx->d_stage_in, words * 8));
        FBS_HIP(ctx, hipMalloc(&ctx->d_stage_out, words * 8));
        FBS_HIP(ctx, hipMalloc(&ctx->d_stage_ids, count * 4));
        ctx->stage_capacity = count;
    }
    hipStream_t s = ctx->stream;
    if ((rc = scratch_wait(ctx, s)) != FBS_OK) return rc;   // the staging buffers are per-context scratch too
    FBS_HIP(ctx, hipMemcpyAsync(ctx->d_stage_in, cts_in, words * 8, hipMemcpyHostToDevice, s));
    if (table_ids) FBS_HIP(ctx, hipMemcpyAsync(ctx->d_stage_ids, table_ids, count * 4, hipMemcpyHostToDevice, s));
    rc = fbs_bootstrap_batch_dev(ctx, tv, ctx->d_stage_in, table_ids ? ctx->d_stage_ids : nullptr, count, ctx->d_stage_out, nullptr);
    if (rc != FBS_OK) {
        (void)hipStreamSynchronize(s);
        return rc;
    }
    FBS_HIP(ctx, hipMemcpyAsync(cts_out, ctx->d_stage_out, words * 8, hipMemcpyDeviceToHost, s));
    FBS_HIP(ctx, hipStreamSynchronize(s));
    return FBS_OK;
} FBS_API_CATCH(ctx)

// ---------------------------------------------------------------------------------------------
// wire-slot building blocks with HOST index arrays (convenience: each call stages its indices through a
// per-context buffer and waits for the copy; hosts that step a loaded program use the fbs_level_* calls below,
// whose index arrays were uploaded once by fbs_program_load)
// ---------------------------------------------------------------------------------------------
static int ensure_idx(fbs_ctx *ctx, size_t words) {
    if (words <= ctx->idx_capacity) return FBS_OK;
    if (ctx->scratch_used) FBS_HIP(ctx, hipStreamSynchronize(ctx->scratch_stream));
    if (ctx->d_idx) (void)hipFree(ctx->d_idx);
    ctx->d_idx = nullptr;
    ctx->idx_capacity = 0;
    size_t cap = std::max<size_t>(words, 1 << 16);
    FBS_HIP(ctx, hipMalloc(&ctx->d_idx, cap * 4));
    ctx->idx_capacity = cap;
    return FBS_OK;
}

int fbs_lincomb_dev(fbs_ctx *ctx, uint64_t *d_wires, size_t T, uint32_t n_out, const uint32_t *dst, const uint32_t *term_off,
                    const uint32_t *srcs, const int64_t *coefs, const int64_t *consts, void *stream) try {
    int rc = check_ready(ctx, nullptr);
    if (rc != FBS_OK) return rc;
    if (n_out == 0 || T == 0) return FBS_OK;
    if (!d_wires || !dst || !term_off || !consts) return set_error(ctx, FBS_E_INVALID, "null argument");
    const uint32_t n_terms = term_off[n_out];
    if (n_terms && (!srcs || !coefs)) return set_error(ctx, FBS_E_INVALID, "null argument");
    // staging: [dst n_out][term_off n_out+1][srcs n_terms] as u32, then coefs and consts as u64
    size_t u32_words = (size_t)n_out + (n_out + 1) + n_terms;
    u32_words = (u32_words + 1) & ~(size_t)1;
    size_t total = u32_words + 2 * ((size_t)n_terms + n_out);
    hipStream_t s = pick(ctx, stream);
    if (ctx->scratch_used) FBS_HIP(ctx, hipStreamSynchronize(ctx->scratch_stream));   // the staging buffer may still be read
    rc = ensure_idx(ctx, total);
    if (rc != FBS_OK) return rc;
    std::vector<uint32_t> stage(total, 0);
    std::memcpy(stage.data(), dst, (size_t)n_out * 4);
    std::memcpy(stage.data() + n_out, term_off, (size_t)(n_out + 1) * 4);
    if (n_terms) std::memcpy(stage.data() + 2 * (size_t)n_out + 1, srcs, (size_t)n_terms * 4);
    uint64_t *f = reinterpret_cast<uint64_t *>(stage.data() + u32_words);
    for (uint32_t i = 0; i < n_terms; i++) f[i] = coef_bits(coefs[i]);
    for (uint32_t g = 0; g < n_out; g++) f[n_terms + g] = fq_mul(fq_from_i64(consts[g]), 2 * ctx->delta_half);
    FBS_HIP(ctx, hipMemcpyAsync(ctx->d_idx, stage.data(), total * 4, hipMemcpyHostToDevice, s));
    FBS_HIP(ctx, hipStreamSynchronize(s));   // `stage` is pageable host memory going out of scope
    const uint32_t *d_dst = ctx->d_idx, *d_off = ctx->d_idx + n_out, *d_srcs = ctx->d_idx + 2 * (size_t)n_out + 1;
    const uint64_t *d_f = reinterpret_cast<const uint64_t *>(ctx->d_idx + u32_words);
    rc = dev_lincomb(ctx, d_wires, T, 0, T, n_out, d_dst, d_off, d_srcs, d_f, d_f + n_terms, s);
    if (rc != FBS_OK) return rc;
    return scratch_done(ctx, s);
} FBS_API_CATCH(ctx)

int fbs_bootstrap_wires_dev(fbs_ctx *ctx, const fbs_tvset *tv, uint64_t *d_wires, size_t T, uint32_t n_gates,
                            const uint32_t *src, const uint32_t *dst, const uint32_t *table_ids, size_t s_begin, size_t s_end,
                            void *stream) try {
    int rc = check_ready(ctx, tv);
    if (rc != FBS_OK) return rc;
    if (!tv || !d_wires || !src || !dst || !table_ids) return set_error(ctx, FBS_E_INVALID, "null argument");
    if (s_end > T || s_begin > s_end) return set_error(ctx, FBS_E_INVALID, "bad sample range");
    if (n_gates == 0 || s_begin == s_end) return FBS_OK;
    for (uint32_t g = 0; g < n_gates; g++)
        if (table_ids[g] >= tv->n_tables) return set_error(ctx, FBS_E_INVALID, "table id out of range");
    const size_t count = (size_t)n_gates * (s_end - s_begin);
    hipStream_t s = pick(ctx, stream);
    if (ctx->scratch_used) FBS_HIP(ctx, hipStreamSynchronize(ctx->scratch_stream));
    rc = ensure_idx(ctx, 3 * (size_t)n_gates);
    if (rc != FBS_OK) return rc;
    rc = ensure_ms(ctx, count);
    if (rc != FBS_OK) return rc;
    std::vector<uint32_t> stage(3 * (size_t)n_gates);
    std::memcpy(stage.data(), src, (size_t)n_gates * 4);
    std::memcpy(stage.data() + n_gates, dst, (size_t)n_gates * 4);
    std::memcpy(stage.data() + 2 * (size_t)n_gates, table_ids, (size_t)n_gates * 4);
    FBS_HIP(ctx, hipMemcpyAsync(ctx->d_idx, stage.data(), stage.size() * 4, hipMemcpyHostToDevice, s));
    FBS_HIP(ctx, hipStreamSynchronize(s));
    GateView gv{};
    gv.in_base = d_wires;
    gv.out_base = d_wires;
    gv.src_slot = ctx->d_idx;
    gv.dst_slot = ctx->d_idx + n_gates;
    gv.table_ids = ctx->d_idx + 2 * (size_t)n_gates;
    gv.T = T;
    gv.s_begin = s_begin;
    gv.s_count = s_end - s_begin;
    gv.f_begin = 0;
    gv.count = count;
    gv.ks_begin = 0;
    gv.ks_count = count;
    gv.n_gates = n_gates;
    rc = dev_keyswitch(ctx, gv, ctx->d_ms, s);
    if (rc != FBS_OK) return rc;
    rc = dev_blind_rotate(ctx, tv, gv, ctx->d_ms, s);
    if (rc != FBS_OK) return rc;
    return scratch_done(ctx, s);
} FBS_API_CATCH(ctx)

// ---------------------------------------------------------------------------------------------
// whole-program executor
// ---------------------------------------------------------------------------------------------
int fbs_program_load(fbs_ctx *ctx, const fbs_program_desc *d, const fbs_tvset *tv, fbs_prog **out) try {
    return fbs_program_load_ex(ctx, d, tv, 0, out);
} FBS_API_CATCH(ctx)

int fbs_program_load_ex(fbs_ctx *ctx, const fbs_program_desc *d, const fbs_tvset *tv, uint32_t flags, fbs_prog **out) try {
    int rc = check_ready(ctx, tv);
    if (rc != FBS_OK) return rc;
    if (!d || !out) return set_error(ctx, FBS_E_INVALID, "null argument");
    *out = nullptr;
    if (flags & ~(uint32_t)FBS_LOAD_FUSE_TABLES) return set_error(ctx, FBS_E_INVALID, "unknown load flag");
    // the schedule: levels, wire slots by liveness, stage tables -- host arithmetic with no device in it (fbs_plan.cpp; the same
    // function runs under the sanitizers in tests/c/host_harness.cpp)
    ProgramPlan plan;
    std::string why;
    const bool fuse = (flags & FBS_LOAD_FUSE_TABLES) != 0;
    if (plan_program(d, tv ? tv->n_tables : 0u, fuse && tv ? tv->fusable.data() : nullptr, &plan, &why) != FBS_OK)
        return set_error(ctx, FBS_E_INVALID, why);
    std::unique_ptr<fbs_prog, void (*)(fbs_prog *)> prog(new fbs_prog, fbs_program_destroy);
    prog->fused = fuse;
    prog->ctx = ctx;
    prog->tv = tv;
    prog->n_inputs = d->n_inputs;
    prog->n_instr = d->n_instr;
    prog->n_outputs = d->n_outputs;
    prog->n_wires = plan.n_wires;
    prog->n_slots = plan.n_slots;
    prog->depth = plan.depth;
    prog->max_width = plan.max_width;
    prog->max_sources = plan.max_sources;
    prog->max_shared = plan.max_shared;
    prog->n_bootstrap = plan.n_bootstrap;
    prog->n_keyswitch = plan.n_keyswitch;
    prog->n_rotations = plan.n_rotations;
    prog->in_slot = plan.in_slot;
    prog->out_slot = plan.out_slot;
    // ---- upload the stage tables (coefficients and constants mapped into the field) ----------------------------------
    prog->lin.resize(plan.depth + 1);
    prog->boot.resize(plan.depth);
    for (uint32_t L = 0; L <= plan.depth; L++) {
        for (const LinPlan &h : plan.lin[L]) {
            LincombStage st;
            st.n_out = (uint32_t)h.dst.size();
            std::vector<uint64_t> coefs(h.coefs.size()), consts(h.consts.size());
            for (size_t i = 0; i < coefs.size(); i++) coefs[i] = coef_bits(h.coefs[i]);
            for (size_t i = 0; i < consts.size(); i++) consts[i] = fq_mul(fq_from_i64(h.consts[i]), 2 * ctx->delta_half);
            if (st.n_out &&
                ((rc = to_device(ctx, prog.get(), h.dst, &st.d_dst)) || (rc = to_device(ctx, prog.get(), h.off, &st.d_term_off)) ||
                 (rc = to_device(ctx, prog.get(), h.srcs, &st.d_srcs)) || (rc = to_device(ctx, prog.get(), coefs, &st.d_coefs)) ||
                 (rc = to_device(ctx, prog.get(), consts, &st.d_consts))))
                return rc;
            prog->lin[L].push_back(st);   // kept even when empty: the plan's stage times count every sub-stage
        }
    }
    for (uint32_t L = 0; L < plan.depth; L++) {
        const BootPlan &h = plan.boot[L];
        BootStage st;
        st.n_gates = (uint32_t)h.dst.size();
        st.n_sources = (uint32_t)h.src_slot.size();
        st.n_shared = h.n_shared;
        st.n_extract = (uint32_t)h.x_row.size();
        st.source_of = h.source_of;
        if ((rc = to_device(ctx, prog.get(), h.src_slot, &st.d_src_slot)) || (rc = to_device(ctx, prog.get(), h.source_of, &st.d_source_of)) ||
            (rc = to_device(ctx, prog.get(), h.dst, &st.d_dst)) || (rc = to_device(ctx, prog.get(), h.table, &st.d_table)))
            return rc;
        if (st.n_extract && ((rc = to_device(ctx, prog.get(), h.x_row, &st.d_x_row)) || (rc = to_device(ctx, prog.get(), h.x_table, &st.d_x_table)) ||
                             (rc = to_device(ctx, prog.get(), h.x_dst, &st.d_x_dst)) || (rc = to_device(ctx, prog.get(), h.x_gate, &st.d_x_gate))))
            return rc;
        prog->boot[L] = std::move(st);
    }
    *out = prog.release();
    return FBS_OK;
} FBS_API_CATCH(ctx)

void fbs_program_destroy(fbs_prog *prog) try {
    if (!prog) return;
    if (prog->ctx) (void)hipSetDevice(prog->ctx->device);
    for (void *p : prog->allocations) (void)hipFree(p);
    delete prog;
} catch (...) {   // (nothing here allocates; the promise of the ABI is kept anyway)
}

int fbs_program_info(const fbs_prog *prog, uint32_t *n_levels, uint32_t *max_width, uint32_t *n_bootstrap) try {
    if (!prog) return FBS_E_INVALID;
    if (n_levels) *n_levels = prog->depth;
    if (max_width) *max_width = prog->max_width;
    if (n_bootstrap) *n_bootstrap = prog->n_bootstrap;
    return FBS_OK;
} FBS_API_CATCH(prog ? prog->ctx : nullptr)

int fbs_program_layout(const fbs_prog *prog, fbs_layout *out) try {
    if (!prog || !out) return FBS_E_INVALID;
    out->n_slots = prog->n_slots;
    out->n_levels = prog->depth;
    out->max_width = prog->max_width;
    out->max_sources = prog->max_sources;
    out->n_bootstrap = prog->n_bootstrap;
    out->n_keyswitch = prog->n_keyswitch;
    out->n_rotations = prog->n_rotations;
    out->row_words = prog->fused ? (prog->ctx->p.k + 1) * prog->ctx->N : prog->ctx->D + 1;
    out->n_inputs = prog->n_inputs;
    out->n_outputs = prog->n_outputs;
    return FBS_OK;
} FBS_API_CATCH(prog ? prog->ctx : nullptr)

int fbs_program_level(const fbs_prog *prog, uint32_t level, uint32_t *n_gates, uint32_t *n_sources) try {
    if (!prog || level >= prog->depth) return FBS_E_INVALID;
    if (n_gates) *n_gates = prog->boot[level].n_gates;
    if (n_sources) *n_sources = prog->boot[level].n_sources;
    return FBS_OK;
} FBS_API_CATCH(prog ? prog->ctx : nullptr)

int fbs_program_io_slots(const fbs_prog *prog, uint32_t *in_slot, int64_t *out_slot) try {
    if (!prog) return FBS_E_INVALID;
    if (in_slot) std::copy(prog->in_slot.begin(), prog->in_slot.end(), in_slot);
    if (out_slot) std::copy(prog->out_slot.begin(), prog->out_slot.end(), out_slot);
    return FBS_OK;
} FBS_API_CATCH(prog ? prog->ctx : nullptr)

// ---- one level at a time, device-resident wires, nothing but kernel launches on `stream` ------------------------
static int check_level_call(fbs_ctx *ctx, const fbs_prog *prog, const uint64_t *d_wires, size_t T, size_t s_begin, size_t s_count) {
    int rc = check_ready(ctx, prog ? prog->tv : nullptr);
    if (rc != FBS_OK) return rc;
    if (!prog || prog->ctx != ctx) return set_error(ctx, FBS_E_INVALID, "program belongs to another context");
    if (!d_wires) return set_error(ctx, FBS_E_INVALID, "null argument");
    if (s_begin + s_count > T) return set_error(ctx, FBS_E_INVALID, "bad sample range");
    return FBS_OK;
}

int fbs_level_lincomb_dev(fbs_ctx *ctx, const fbs_prog *prog, uint32_t level, uint64_t *d_wires, size_t T, size_t s_begin,
                          size_t s_count, void *stream) try {
    int rc = check_level_call(ctx, prog, d_wires, T, s_begin, s_count);
    if (rc != FBS_OK) return rc;
    if (level > prog->depth) return set_error(ctx, FBS_E_INVALID, "level out of range");
    hipStream_t s = pick(ctx, stream);
    for (const LincombStage &st : prog->lin[level]) {
        rc = dev_lincomb(ctx, d_wires, T, s_begin, s_count, st.n_out, st.d_dst, st.d_term_off, st.d_srcs, st.d_coefs, st.d_consts, s);
        if (rc != FBS_OK) return rc;
    }
    return FBS_OK;
} FBS_API_CATCH(ctx)

int fbs_level_bootstrap_dev(fbs_ctx *ctx, const fbs_prog *prog, uint32_t level, uint64_t *d_wires, size_t T, size_t s_begin,
                            size_t s_count, size_t f_begin, size_t f_end, uint64_t *d_rows, void *stream) try {
    int rc = check_level_call(ctx, prog, d_wires, T, s_begin, s_count);
    if (rc != FBS_OK) return rc;
    if (level >= prog->depth) return set_error(ctx, FBS_E_INVALID, "level out of range");
    const BootStage &b = prog->boot[level];
    if (f_begin > f_end || f_end > (size_t)b.n_gates * s_count) return set_error(ctx, FBS_E_INVALID, "bad bootstrap range");
    if (f_begin == f_end) return FBS_OK;   // (covers s_count == 0)
    // A level with shared rotations: into the wire slots it runs whole (the tables of one source are cut from one accumulator
    // right after the rotations).  Into rows it can be SLICED: rows are then (k + 1) N words (fbs_layout.row_words), an ordinary gate
    // leaves its ciphertext in its row and a shared rotation its whole accumulator -- the unit dealt out across GPUs is the
    // rotation -- and fbs_level_scatter_dev cuts the tables out once every row is there.
    if (b.n_shared && !d_rows && (f_begin != 0 || f_end != (size_t)b.n_gates * s_count))
        return set_error(ctx, FBS_E_INVALID, "a level of a fused program runs whole when it writes to the wire slots (slice it into rows)");
    // the key switches this slice needs: the (source, sample) pairs of its gates, as ONE flattened range.  Gates are
    // sorted by source, so only the first and the last gate of the slice can be cut short in the sample direction, and
    // only while no other gate of the slice shares their source.
    const size_t g0 = f_begin / s_count, s0 = f_begin % s_count;
    const size_t g1 = (f_end - 1) / s_count, s1 = (f_end - 1) % s_count + 1;
    const size_t u0 = b.source_of[g0], u1 = b.source_of[g1];
    const bool first_shared = g1 > g0 && b.source_of[g0 + 1] == u0;
    const bool last_shared = g1 > g0 && b.source_of[g1 - 1] == u1;
    GateView gv{};
    gv.in_base = d_wires;
    gv.out_base = d_wires;
    gv.src_slot = b.d_src_slot;
    gv.dst_slot = b.d_dst;
    gv.table_ids = b.d_table;
    gv.source_of = b.d_source_of;
    gv.out_rows = d_rows;
    gv.row_words = (d_rows && prog->fused) ? (ctx->p.k + 1) * ctx->N : 0;
    gv.T = T;
    gv.s_begin = s_begin;
    gv.s_count = s_count;
    gv.f_begin = f_begin;
    gv.count = f_end - f_begin;
    gv.ks_begin = u0 * s_count + (first_shared ? 0 : s0);
    gv.ks_count = u1 * s_count + (last_shared ? s_count : s1) - gv.ks_begin;
    gv.n_gates = b.n_gates;
    rc = ensure_ms(ctx, gv.ks_count);
    if (rc != FBS_OK) return rc;
    if (b.n_shared && !d_rows) {
        if ((rc = ensure_acc(ctx, (size_t)b.n_shared * s_count)) != FBS_OK) return rc;
        gv.acc_rows = ctx->d_acc;
    }
    hipStream_t s = pick(ctx, stream);
    if ((rc = scratch_wait(ctx, s)) != FBS_OK) return rc;
    if ((rc = dev_keyswitch(ctx, gv, ctx->d_ms, s)) != FBS_OK) return scratch_fail(ctx, s, rc);
    if ((rc = dev_blind_rotate(ctx, prog->tv, gv, ctx->d_ms, s)) != FBS_OK) return scratch_fail(ctx, s, rc);
    if (b.n_shared && !d_rows &&
        (rc = dev_multi_extract(ctx, prog->tv, ctx->d_acc, d_wires, T, s_begin, s_count, b.n_extract, b.d_x_row, b.d_x_table, b.d_x_dst, s)) != FBS_OK)
        return scratch_fail(ctx, s, rc);
    return scratch_done(ctx, s);
} FBS_API_CATCH(ctx)

int fbs_level_scatter_dev(fbs_ctx *ctx, const fbs_prog *prog, uint32_t level, uint64_t *d_wires, size_t T, size_t s_begin,
                          size_t s_count, const uint64_t *d_rows, size_t f_begin, size_t f_end, void *stream) try {
    int rc = check_level_call(ctx, prog, d_wires, T, s_begin, s_count);
    if (rc != FBS_OK) return rc;
    if (level >= prog->depth) return set_error(ctx, FBS_E_INVALID, "level out of range");
    const BootStage &b = prog->boot[level];
    if (!d_rows || f_begin > f_end || f_end > (size_t)b.n_gates * s_count) return set_error(ctx, FBS_E_INVALID, "bad row range");
    if (!prog->fused) return dev_scatter_rows(ctx, d_wires, T, s_begin, s_count, b.d_dst, d_rows, f_begin, f_end - f_begin, 0, pick(ctx, stream));
    // fused: rows of (k + 1) N words; the tables of shared rotations are cut out of the gathered accumulators, which takes every row
    if (b.n_shared && (f_begin != 0 || f_end != (size_t)b.n_gates * s_count))
        return set_error(ctx, FBS_E_INVALID, "scattering a level of a fused program takes all of its rows");
    rc = dev_scatter_rows(ctx, d_wires, T, s_begin, s_count, b.d_dst, d_rows, f_begin, f_end - f_begin, (ctx->p.k + 1) * ctx->N, pick(ctx, stream));
    if (rc != FBS_OK || !b.n_shared) return rc;
    return dev_multi_extract(ctx, prog->tv, d_rows, d_wires, T, s_begin, s_count, b.n_extract, b.d_x_gate, b.d_x_table, b.d_x_dst,
                             pick(ctx, stream));
} FBS_API_CATCH(ctx)

static int run_levels(fbs_ctx *ctx, const fbs_prog *prog, uint64_t *d_wires, size_t T, size_t s_count, hipStream_t s) {
    int rc;
    for (uint32_t L = 0; L <= prog->depth; L++) {
        if ((rc = fbs_level_lincomb_dev(ctx, prog, L, d_wires, T, 0, s_count, s)) != FBS_OK) return rc;
        if (L == prog->depth) break;
        const size_t total = (size_t)prog->boot[L].n_gates * s_count;
        if ((rc = fbs_level_bootstrap_dev(ctx, prog, L, d_wires, T, 0, s_count, 0, total, nullptr, s)) != FBS_OK) return rc;
    }
    return FBS_OK;
}

// Samples are independent through the whole program: evaluate in chunks whose wire slots fit in HBM.  The wire buffer
// belongs to the context and is shared by all of its programs (it only ever grows).
static int reserve_wires(fbs_ctx *ctx, const fbs_prog *prog, size_t T, size_t *chunk) {
    const size_t ctw = ctx->D + 1;
    const size_t per_sample = (size_t)prog->n_slots * ctw * 8 + (size_t)std::max(1u, prog->max_sources) * (ctx->p.n + 1) * 4 +
                              (size_t)prog->max_shared * (ctx->p.k + 1) * ctx->N * 8;
    size_t free_b = 0, total_b = 0;
    FBS_HIP(ctx, hipMemGetInfo(&free_b, &total_b));
    size_t have = free_b + ctx->wires_capacity * 8 + ctx->ms_capacity * (ctx->p.n + 1) * 4 + ctx->acc_capacity * (size_t)(ctx->p.k + 1) * ctx->N * 8;
    // test hook: FBS_WIRE_BUDGET_MB caps what the wire slots may take, so that the chunked path runs at small sizes
    if (const char *cap = getenv("FBS_WIRE_BUDGET_MB")) have = std::min<size_t>(have, (size_t)std::max(1, atoi(cap)) << 20);
    const size_t Tc = std::min<size_t>(T, std::max<size_t>(1, (size_t)(0.6 * (double)have) / per_sample));
    const size_t words = Tc * (size_t)prog->n_slots * ctw;
    int rc = ensure_wires(ctx, words);
    if (rc != FBS_OK) return rc;
    rc = ensure_ms(ctx, (size_t)std::max(1u, prog->max_sources) * Tc);
    if (rc != FBS_OK) return rc;
    if (prog->max_shared && (rc = ensure_acc(ctx, (size_t)prog->max_shared * Tc)) != FBS_OK) return rc;
    *chunk = Tc;
    return FBS_OK;
}

static uint64_t trivial_body(const fbs_ctx *ctx, int64_t out_slot) { return fq_mul(fq_from_i64(-1 - out_slot), 2 * ctx->delta_half); }

int fbs_eval_dev(fbs_ctx *ctx, fbs_prog *prog, const uint64_t *d_in, size_t T, uint64_t *d_out, void *stream) try {
    int rc = check_ready(ctx, prog ? prog->tv : nullptr);
    if (rc != FBS_OK) return rc;
    if (!prog || prog->ctx != ctx) return set_error(ctx, FBS_E_INVALID, "program belongs to another context");
    if (T == 0) return FBS_OK;
    if ((prog->n_inputs && !d_in) || (prog->n_outputs && !d_out)) return set_error(ctx, FBS_E_INVALID, "null argument");
    const size_t ctw = ctx->D + 1;
    hipStream_t s = pick(ctx, stream);
    size_t Tc = 0;
    if ((rc = reserve_wires(ctx, prog, T, &Tc)) != FBS_OK) return rc;
    if ((rc = scratch_wait(ctx, s)) != FBS_OK) return rc;
    for (size_t s0 = 0; s0 < T; s0 += Tc) {
        const size_t tc = std::min(Tc, T - s0);
        for (uint32_t i = 0; i < prog->n_inputs; i++)
            FBS_HIP(ctx, hipMemcpyAsync(ctx->d_wires + (size_t)prog->in_slot[i] * Tc * ctw, d_in + ((size_t)i * T + s0) * ctw, tc * ctw * 8,
                                        hipMemcpyDeviceToDevice, s));
        if ((rc = run_levels(ctx, prog, ctx->d_wires, Tc, tc, s)) != FBS_OK) return rc;
        for (uint32_t o = 0; o < prog->n_outputs; o++) {
            const int64_t w = prog->out_slot[o];
            rc = dev_copy_out(ctx, ctx->d_wires, Tc, 0, tc, w, w < 0 ? trivial_body(ctx, w) : 0, d_out + ((size_t)o * T + s0) * ctw, s);
            if (rc != FBS_OK) return rc;
        }
    }
    return scratch_done(ctx, s);
} FBS_API_CATCH(ctx)

int fbs_eval(fbs_ctx *ctx, fbs_prog *prog, const uint64_t *in_cts, size_t T, uint64_t *out_cts) try {
    int rc = check_ready(ctx, prog ? prog->tv : nullptr);
    if (rc != FBS_OK) return rc;
    if (!prog || prog->ctx != ctx) return set_error(ctx, FBS_E_INVALID, "program belongs to another context");
    if (T == 0) return FBS_OK;
    if ((prog->n_inputs && !in_cts) || (prog->n_outputs && !out_cts)) return set_error(ctx, FBS_E_INVALID, "null argument");
    const size_t ctw = ctx->D + 1;
    hipStream_t s = ctx->stream;
    size_t Tc = 0;
    if ((rc = reserve_wires(ctx, prog, T, &Tc)) != FBS_OK) return rc;
    if ((rc = scratch_wait(ctx, s)) != FBS_OK) return rc;
    for (size_t s0 = 0; s0 < T; s0 += Tc) {
        const size_t tc = std::min(Tc, T - s0);
        for (uint32_t i = 0; i < prog->n_inputs; i++)
            FBS_HIP(ctx, hipMemcpyAsync(ctx->d_wires + (size_t)prog->in_slot[i] * Tc * ctw, in_cts + ((size_t)i * T + s0) * ctw, tc * ctw * 8,
                                        hipMemcpyHostToDevice, s));
        if ((rc = run_levels(ctx, prog, ctx->d_wires, Tc, tc, s)) != FBS_OK) return rc;
        for (uint32_t o = 0; o < prog->n_outputs; o++) {
            uint64_t *dst = out_cts + ((size_t)o * T + s0) * ctw;
            const int64_t w = prog->out_slot[o];
            if (w >= 0) {
                FBS_HIP(ctx, hipMemcpyAsync(dst, ctx->d_wires + (size_t)w * Tc * ctw, tc * ctw * 8, hipMemcpyDeviceToHost, s));
            } else {
                const uint64_t body = trivial_body(ctx, w);   // trivial ciphertext of the constant
                for (size_t q = 0; q < tc; q++) {
                    std::memset(dst + q * ctw, 0, ctx->D * 8);
                    dst[q * ctw + ctx->D] = body;
                }
            }
        }
        FBS_HIP(ctx, hipStreamSynchronize(s));
    }
    return scratch_done(ctx, s);
} FBS_API_CATCH(ctx)

// ---------------------------------------------------------------------------------------------
int fbs_profile_enable(fbs_ctx *ctx, int on) try {
    if (!ctx) return FBS_E_INVALID;
    ctx->prof.on = on != 0;
    return FBS_OK;
} FBS_API_CATCH(ctx)

static int profile_collect(fbs_ctx *ctx) {
    FBS_HIP(ctx, hipSetDevice(ctx->device));
    for (int k = 0; k < 3; k++) {
        for (auto &pr : ctx->prof.pending[k]) {
            FBS_HIP(ctx, hipEventSynchronize(pr.end));
            float t = 0;
            FBS_HIP(ctx, hipEventElapsedTime(&t, pr.begin, pr.end));
            ctx->prof.ms[k] += t;
            ctx->prof.launches[k]++;
            Profile::PerKernel &pk = ctx->prof.by_kernel[k][pr.kernel];
            pk.ms += t;
            pk.launches++;
            ctx->prof.pool.push_back({pr.begin, pr.end});
        }
        ctx->prof.pending[k].clear();
    }
    return FBS_OK;
}

int fbs_profile_read(fbs_ctx *ctx, double ms[3], uint64_t launches[3], int reset) try {
    if (!ctx) return FBS_E_INVALID;
    if (int rc = profile_collect(ctx)) return rc;
    for (int k = 0; k < 3; k++) {
        if (ms) ms[k] = ctx->prof.ms[k];
        if (launches) launches[k] = ctx->prof.launches[k];
        if (reset) {
            ctx->prof.ms[k] = 0;
            ctx->prof.launches[k] = 0;
            ctx->prof.by_kernel[k].clear();
        }
    }
    return FBS_OK;
} FBS_API_CATCH(ctx)

int fbs_profile_kernels(fbs_ctx *ctx, char *buf, size_t cap, size_t *needed) try {
    if (!ctx) return FBS_E_INVALID;
    if (int rc = profile_collect(ctx)) return rc;
    std::string text;
    for (int k = 0; k < 3; k++)
        for (const auto &kv : ctx->prof.by_kernel[k]) {
            char line[64];
            snprintf(line, sizeof line, "\t%llu\t%.6f\n", (unsigned long long)kv.second.launches, kv.second.ms);
            text += std::to_string(k) + "\t" + kv.first + line;
        }
    if (needed) *needed = text.size() + 1;
    if (!buf || cap < text.size() + 1) return buf ? set_error(ctx, FBS_E_INVALID, "buffer too small") : FBS_OK;
    std::memcpy(buf, text.c_str(), text.size() + 1);
    return FBS_OK;
} FBS_API_CATCH(ctx)

const char *fbs_profile_kernel(const fbs_ctx *ctx, int which) try {
    if (!ctx || which < 0 || which > 2) return "";
    return ctx->prof.kernel[which].c_str();
} catch (...) { return ""; }

const char *fbs_kernel_catalog(void) try {
    static const std::string text = [] {
        std::vector<std::string> names;
        keyswitch_catalog(&names);
        blind_rotate_catalog(&names);
        std::string t;
        for (const std::string &n : names) t += n + "\n";
        return t;
    }();
    return text.c_str();
} catch (...) { return ""; }

int fbs_sync(fbs_ctx *ctx, void *stream) try {
    if (!ctx) return FBS_E_INVALID;
    FBS_HIP(ctx, hipSetDevice(ctx->device));
    FBS_HIP(ctx, hipStreamSynchronize(pick(ctx, stream)));
    return FBS_OK;
} FBS_API_CATCH(ctx)

int fbs_debug_polymul(fbs_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *c) try {
    int rc = check_ready(ctx, nullptr);
    if (rc != FBS_OK) return rc;
    const size_t bytes = (size_t)ctx->N * 8;
    uint64_t *d = nullptr;
    FBS_HIP(ctx, hipMalloc(&d, 3 * bytes));
    hipError_t e = hipMemcpy(d, a, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d + ctx->N, b, bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        rc = dev_polymul(ctx, d, d + ctx->N, d + 2 * (size_t)ctx->N, ctx->stream);
        if (rc == FBS_OK) e = hipStreamSynchronize(ctx->stream);
    }
    if (e == hipSuccess && rc == FBS_OK) e = hipMemcpy(c, d + 2 * (size_t)ctx->N, bytes, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (rc != FBS_OK) return rc;
    if (e != hipSuccess) return set_error(ctx, FBS_E_DEVICE, std::string("polymul: ") + hipGetErrorString(e));
    return FBS_OK;
} FBS_API_CATCH(ctx)

// test hook: raise inside an entry point what a host allocation or a library call could raise, to show the barrier holds
// (kind 0: std::bad_alloc, 1: std::length_error, 2: std::runtime_error, 3: a non-standard exception; anything else: no throw)
int fbs_debug_raise(fbs_ctx *ctx, int kind) try {
    if (kind == 0) throw std::bad_alloc();
    if (kind == 1) throw std::length_error("vector::_M_default_append");
    if (kind == 2) throw std::runtime_error("raised on request");
    if (kind == 3) throw 42;
    return FBS_OK;
} FBS_API_CATCH(ctx)

}  // extern "C"

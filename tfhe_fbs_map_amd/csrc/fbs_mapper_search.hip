// the reference mapper's coefficient search on the GPU.
//
// Reference: MapToFBSHeur._find_lincomb_coefs_search, fbs_mapper/map_to_fbs.py:363-392, on the validity rules of
// :70-113 and the candidate grid of :344-361.  Given the multi-value columns x, y of two cones over the R rows of their
// joint truth table (R <= 2^16: max_truth_table_size, map_circuit.py:106) and the merged output bit per row, find (a, b)
// such that v = a x + b y is a legal bootstrap input.  The reference walks ~(2 s1 + 1)(s2 + 1) candidates one by one,
// each with a handful of numpy passes and a Python set intersection over the rows -- its own bottleneck.
// Here every candidate is one workgroup: two passes over the rows (min / max of v; then which values occur with output
// 0 and with output 1, as two 128-bit masks relative to min v, and the sum of squares), and one thread applies the
// table rules to the masks.  The host enumerates the candidates in the reference's order and picks the winner by the
// reference's rule, so the result is the reference's, bit for bit.  Integer work, bound by L2 reads of x, y, tt
// (R x 9 bytes per candidate; the three arrays stay in L2).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <vector>

#include "fbs_internal.hpp"

namespace fbs {

typedef unsigned __int128 mask128;

struct SearchArgs {
    const int32_t *x, *y;
    const uint8_t *tt;
    const int32_t *ca, *cb;     // candidates
    uint8_t *valid;             // [n_cand]
    long long *norm2;           // [n_cand]
    uint32_t rows, fbs_size, max_fbs_size;
};

__device__ __forceinline__ mask128 low_bits(uint32_t n) { return n >= 128 ? ~(mask128)0 : (((mask128)1 << n) - 1); }

// the table over [min v, max v] with its don't-care slots filled by `tv` is evaluable at fbs_size (map_to_fbs.py:81-98)
__device__ bool table_ok(mask128 tv, uint32_t size, uint32_t fbs, uint32_t max_fbs) {
    if (size <= fbs) return true;
    if (size > max_fbs) return false;
    const uint32_t d = size - fbs;
    const mask128 m = low_bits(d), start = tv & m, end = (tv >> fbs) & m;
    const bool mode1 = (start ^ end) == m;                 // f(x) == -f(x + fbs_size)
    const bool mode2 = start == 0 && end == 0;             // both 0
    const bool mode3 = start == m && end == m;             // both 1
    return mode1 || mode2 || mode3;
}

__global__ __launch_bounds__(256) void k_lincomb_search(SearchArgs s) {
    __shared__ long long red_a[256], red_b[256];
    __shared__ unsigned long long m0lo[256], m0hi[256], m1lo[256], m1hi[256];
    const int t = threadIdx.x;
    const long long a = s.ca[blockIdx.x], b = s.cb[blockIdx.x];

    // pass 1: range of v = a x + b y
    long long lo = 0x7FFFFFFFFFFFFFFFll, hi = -0x7FFFFFFFFFFFFFFFll;
    for (uint32_t r = t; r < s.rows; r += 256) {
        const long long v = a * s.x[r] + b * s.y[r];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
    }
    red_a[t] = lo;
    red_b[t] = hi;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if (t < w) {
            red_a[t] = red_a[t + w] < red_a[t] ? red_a[t + w] : red_a[t];
            red_b[t] = red_b[t + w] > red_b[t] ? red_b[t + w] : red_b[t];
        }
        __syncthreads();
    }
    lo = red_a[0];
    hi = red_b[0];
    __syncthreads();
    const unsigned long long span = (unsigned long long)(hi - lo) + 1ull;
    if (span > 128 || span > s.max_fbs_size) {   // too long a table for any rule (fbs_size <= max_fbs_size <= 128)
        if (t == 0) {
            s.valid[blockIdx.x] = 0;
            s.norm2[blockIdx.x] = 0;
        }
        return;
    }
    // pass 2: values seen with output 0 / output 1 (bit v - lo), sum of squares
    mask128 seen0 = 0, seen1 = 0;
    long long sq = 0;
    for (uint32_t r = t; r < s.rows; r += 256) {
        const long long v = a * s.x[r] + b * s.y[r];
        const mask128 bit = (mask128)1 << (uint32_t)(v - lo);
        if (s.tt[r]) seen1 |= bit;
        else seen0 |= bit;
        sq += v * v;
    }
    m0lo[t] = (unsigned long long)seen0;
    m0hi[t] = (unsigned long long)(seen0 >> 64);
    m1lo[t] = (unsigned long long)seen1;
    m1hi[t] = (unsigned long long)(seen1 >> 64);
    red_a[t] = sq;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if (t < w) {
            m0lo[t] |= m0lo[t + w];
            m0hi[t] |= m0hi[t + w];
            m1lo[t] |= m1lo[t + w];
            m1hi[t] |= m1hi[t + w];
            red_a[t] += red_a[t + w];
        }
        __syncthreads();
    }
    if (t) return;
    seen0 = ((mask128)m0hi[0] << 64) | m0lo[0];
    seen1 = ((mask128)m1hi[0] << 64) | m1lo[0];
    const uint32_t size = (uint32_t)span;
    bool ok = (seen0 & seen1) == 0;                           // _is_mvt_valid, :78-79
    if (ok && size > s.fbs_size) {                            // _is_lut_valid, :100-113
        const mask128 holes = ~(seen0 | seen1) & low_bits(size);
        ok = table_ok(seen1, size, s.fbs_size, s.max_fbs_size) || table_ok(seen1 | holes, size, s.fbs_size, s.max_fbs_size);
    }
    s.valid[blockIdx.x] = ok ? 1 : 0;
    s.norm2[blockIdx.x] = red_a[0];
}

}  // namespace fbs

using namespace fbs;

struct fbs_searcher {
    int device = 0;
    hipStream_t stream = nullptr;
    mutable std::string err;
    void *d_rows = nullptr;     // x, y (int32 each) and tt (uint8), rows_cap rows
    size_t rows_cap = 0;
    void *d_cand = nullptr;     // ca, cb (int32), norm2 (int64), valid (uint8), cand_cap candidates
    size_t cand_cap = 0;
    double last_kernel_ms = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
};

static thread_local std::string g_search_error;
static int search_error(const fbs_searcher *s, int code, const std::string &msg) {
    if (s) s->err = msg;
    else g_search_error = msg;
    return code;
}
#define SEARCH_HIP(s, call)                                                                                  \
    do {                                                                                                     \
        hipError_t e__ = (call);                                                                             \
        if (e__ != hipSuccess) return search_error(s, FBS_E_DEVICE, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

// (the exception barrier of the ABI: fbs_internal.hpp, translate_exception)
#define SEARCH_API_CATCH(owner)                                                                                  \
    catch (...) {                                                                                                \
        const fbs_searcher *owner__ = (owner);                                                                   \
        return fbs::translate_exception([&](int code, const char *text) { search_error(owner__, code, text); }); \
    }

extern "C" {

int fbs_searcher_create(int device, fbs_searcher **out) try {
    if (!out) return FBS_E_INVALID;
    *out = nullptr;
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0)
        return search_error(nullptr, FBS_E_DEVICE, std::string("no HIP device: libfbsexec has no CPU path (") + hipGetErrorString(e) + ")");
    if (device < 0 || device >= n_dev) return search_error(nullptr, FBS_E_INVALID, "device ordinal out of range");
    SEARCH_HIP(nullptr, hipSetDevice(device));
    fbs_searcher *s = new fbs_searcher;
    s->device = device;
    if (hipStreamCreate(&s->stream) != hipSuccess || hipEventCreate(&s->e0) != hipSuccess || hipEventCreate(&s->e1) != hipSuccess) {
        delete s;
        return search_error(nullptr, FBS_E_DEVICE, "stream / event creation failed");
    }
    *out = s;
    return FBS_OK;
} SEARCH_API_CATCH(nullptr)

void fbs_searcher_destroy(fbs_searcher *s) try {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->d_rows) (void)hipFree(s->d_rows);
    if (s->d_cand) (void)hipFree(s->d_cand);
    if (s->e0) (void)hipEventDestroy(s->e0);
    if (s->e1) (void)hipEventDestroy(s->e1);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
} catch (...) {
}

const char *fbs_searcher_last_error(const fbs_searcher *s) { return s ? s->err.c_str() : g_search_error.c_str(); }
double fbs_searcher_last_kernel_ms(const fbs_searcher *s) { return s ? s->last_kernel_ms : 0.0; }

int fbs_search_lincomb_coefs(fbs_searcher *s, const int32_t *x, const int32_t *y, const uint8_t *tt, uint32_t rows,
                             uint32_t fbs_size, uint32_t max_fbs_size, int32_t ab[2], int64_t *mvt, int *found) try {
    if (!s || !x || !y || !tt || !ab || !found) return FBS_E_INVALID;
    *found = 0;
    if (rows == 0) return search_error(s, FBS_E_INVALID, "no rows");
    if (fbs_size < 1 || max_fbs_size < fbs_size || max_fbs_size > 128)
        return search_error(s, FBS_E_INVALID, "need 1 <= fbs_size <= max_fbs_size <= 128");
    SEARCH_HIP(s, hipSetDevice(s->device));
    // candidate grid in the reference's order (:344-361): keys ascending, pairs of one key descending
    int32_t x_lo = x[0], x_hi = x[0], y_lo = y[0], y_hi = y[0];
    for (uint32_t r = 1; r < rows; r++) {
        x_lo = std::min(x_lo, x[r]); x_hi = std::max(x_hi, x[r]);
        y_lo = std::min(y_lo, y[r]); y_hi = std::max(y_hi, y[r]);
    }
    const int64_t s1 = (int64_t)x_hi - x_lo + 1, s2 = (int64_t)y_hi - y_lo + 1;
    if (s1 > 4096 || s2 > 4096) return search_error(s, FBS_E_INVALID, "cone value range above 4096");
    struct Cand {
        int64_t key;
        int32_t a, b;
    };
    std::vector<Cand> cand;
    const int64_t a_lo = s1 < s2 ? 0 : -s2, a_hi = s2, b_lo = s1 < s2 ? -s1 : 0, b_hi = s1;
    for (int64_t a = a_lo; a <= a_hi; a++)
        for (int64_t b = b_lo; b <= b_hi; b++) {
            const int64_t key = std::llabs(a) * (s1 - 1) + std::llabs(b) * (s2 - 1);
            cand.push_back({key, (int32_t)a, (int32_t)b});
        }
    std::stable_sort(cand.begin(), cand.end(), [](const Cand &p, const Cand &q) {
        if (p.key != q.key) return p.key < q.key;
        if (p.a != q.a) return p.a > q.a;
        return p.b > q.b;
    });
    // a table longer than max_fbs_size is never legal, and a candidate's table is at most key + 1 long -- but it can be
    // shorter, so nothing is pruned by key; the kernel prunes by the actual range after its first pass
    const size_t n_cand = cand.size();
    if (rows > s->rows_cap) {
        if (s->d_rows) (void)hipFree(s->d_rows);
        s->d_rows = nullptr;
        s->rows_cap = 0;
        SEARCH_HIP(s, hipMalloc(&s->d_rows, (size_t)rows * 9 + 64));
        s->rows_cap = rows;
    }
    if (n_cand > s->cand_cap) {
        if (s->d_cand) (void)hipFree(s->d_cand);
        s->d_cand = nullptr;
        s->cand_cap = 0;
        SEARCH_HIP(s, hipMalloc(&s->d_cand, n_cand * 17 + 64));
        s->cand_cap = n_cand;
    }
    // layout: [x rows*4][y rows*4][tt rows] and [norm2 n*8][ca n*4][cb n*4][valid n]
    char *dr = (char *)s->d_rows, *dc = (char *)s->d_cand;
    int32_t *d_x = (int32_t *)dr, *d_y = (int32_t *)(dr + (size_t)rows * 4);
    uint8_t *d_tt = (uint8_t *)(dr + (size_t)rows * 8);
    long long *d_norm2 = (long long *)dc;
    int32_t *d_ca = (int32_t *)(dc + n_cand * 8), *d_cb = (int32_t *)(dc + n_cand * 12);
    uint8_t *d_valid = (uint8_t *)(dc + n_cand * 16);
    std::vector<int32_t> ca(n_cand), cb(n_cand);
    for (size_t i = 0; i < n_cand; i++) {
        ca[i] = cand[i].a;
        cb[i] = cand[i].b;
    }
    SEARCH_HIP(s, hipMemcpyAsync(d_x, x, (size_t)rows * 4, hipMemcpyHostToDevice, s->stream));
    SEARCH_HIP(s, hipMemcpyAsync(d_y, y, (size_t)rows * 4, hipMemcpyHostToDevice, s->stream));
    SEARCH_HIP(s, hipMemcpyAsync(d_tt, tt, rows, hipMemcpyHostToDevice, s->stream));
    SEARCH_HIP(s, hipMemcpyAsync(d_ca, ca.data(), n_cand * 4, hipMemcpyHostToDevice, s->stream));
    SEARCH_HIP(s, hipMemcpyAsync(d_cb, cb.data(), n_cand * 4, hipMemcpyHostToDevice, s->stream));
    SearchArgs args{d_x, d_y, d_tt, d_ca, d_cb, d_valid, d_norm2, rows, fbs_size, max_fbs_size};
    SEARCH_HIP(s, hipEventRecord(s->e0, s->stream));
    hipLaunchKernelGGL(k_lincomb_search, dim3((unsigned)n_cand), dim3(256), 0, s->stream, args);
    SEARCH_HIP(s, hipGetLastError());
    SEARCH_HIP(s, hipEventRecord(s->e1, s->stream));
    std::vector<uint8_t> valid(n_cand);
    std::vector<long long> norm2(n_cand);
    SEARCH_HIP(s, hipMemcpyAsync(valid.data(), d_valid, n_cand, hipMemcpyDeviceToHost, s->stream));
    SEARCH_HIP(s, hipMemcpyAsync(norm2.data(), d_norm2, n_cand * 8, hipMemcpyDeviceToHost, s->stream));
    SEARCH_HIP(s, hipStreamSynchronize(s->stream));
    float ms = 0;
    (void)hipEventElapsedTime(&ms, s->e0, s->e1);
    s->last_kernel_ms = ms;
    // the reference's choice (:371-390): the first key that has a legal candidate; within it the smallest sum of squares,
    // the first in walking order on ties
    size_t best = n_cand;
    for (size_t i = 0; i < n_cand; i++) {
        if (best != n_cand && cand[i].key != cand[best].key) break;
        if (valid[i] && (best == n_cand || norm2[i] < norm2[best])) best = i;
    }
    if (best == n_cand) return FBS_OK;
    *found = 1;
    ab[0] = cand[best].a;
    ab[1] = cand[best].b;
    if (mvt)
        for (uint32_t r = 0; r < rows; r++) mvt[r] = (int64_t)ab[0] * x[r] + (int64_t)ab[1] * y[r];
    return FBS_OK;
} SEARCH_API_CATCH(s)

}  // extern "C"

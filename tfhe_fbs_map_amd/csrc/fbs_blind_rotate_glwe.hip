// Blind rotation at ANY GLWE dimension k >= 2 the optimizer of the reference may return (experiments/concrete.patch:163 hands back
// k, N, n, br_l, br_b): N = 256, 512, 1024, any number of gadget levels, one or two key bits per step.  The one shape with kernels of
// its own -- k = 2, N = 1024, one level, two key bits per step: fbs_blind_rotate_k2.hip, what the selector picks for p <= 15 -- stays
// with them; this file is for every other (k, N, l): k = 3 at N = 512 (k N = 1536, between the two noise floors k = 1 offers),
// k = 2 with two levels, k = 2..4 at N = 256 / 512, one key bit per step at k = 2.  gfx950 only.
//
// One bootstrap = k + 1 waves, component c = wave c of the bootstrap (k mask polynomials and the body), every polynomial private to
// its wave (N / 64 = 4, 8 or 16 coefficients per lane; the transforms of the k = 1 kernels at these sizes: NttFor<LOGN, 6>, no
// workgroup barrier inside them); FPW bootstraps share a workgroup, its twiddle tables and -- walking the key in lock step -- the key
// lines in L1.  A step, per wave:
//   one key bit per step:  (X^r - 1) ACC_c through the wave's own LDS words (rotated read), rounded; two key bits: ACC_c itself
//   per gadget level:      balanced digit -> forward transform -> products with key row (c, level) for ALL k + 1 output components,
//                          summed over the levels in registers (two key bits: the key word is the bundle
//                          sum_jj (zeta^e_jj - 1) E_jj, zeta = the evaluation point the register holds)
//   hand-over:             the wave clears its own buffer; barrier; every wave ADDS its products for the k other components into
//                          their buffers (ds_add_f64: exact on integer-valued doubles below 2^53, so the order does not matter);
//                          barrier; own products + what landed -> inverse transform -> accumulate.
// Two workgroup barriers per step.  The columns of a key row are taken in ROTATED order (d = 0 .. k stands for component
// c + d mod k + 1), so that "which product goes where" is the same code in every wave and no register array is indexed by a
// run-time value.  Same rounding rules and, word for word, the same ciphertexts as the oracle (tests/test_gpu_glwe.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <type_traits>
#include <vector>

#include "fbs_blind_rotate.hpp"

namespace fbs {

// where the evaluation held in register m of lane t after W::forward sits in the output array of the textbook in-place
// Cooley-Tukey transform (position P holds the value at psi^(2 bitrev(P) + 1)); lane and register contribute disjoint bits
template <class W, int LOGN>
struct EvalPosition {
    __device__ static __forceinline__ uint32_t lane(uint32_t t) {
        if constexpr (W::HAS_EVAL_POSITION) return W::eval_position_lane(t);
        else return W::template index_of<W::GROUPS - 1>(t, 0);
    }
    __device__ static __forceinline__ uint32_t reg(int m) {
        if constexpr (W::HAS_EVAL_POSITION) return W::eval_position_reg(m);
        else return W::template index_of<W::GROUPS - 1>(0u, m);
    }
    __device__ static __forceinline__ uint32_t exponent(uint32_t position) { return 2u * (__builtin_bitreverse32(position) >> (32 - LOGN)); }
};

template <int LOGN, int K1, int GROUP, int FPW>
__global__ __launch_bounds__(64 * K1 * FPW) void k_blind_rotate_glwe(BrArgs a) {
    using W = typename NttFor<LOGN, 6>::type;
    using Pos = EvalPosition<W, LOGN>;
    constexpr int N = W::N, E = W::E, LANES = W::LANES;
    static_assert(LANES == 64 && (E == 4 || E == 8 || E == 16), "one wave per polynomial");
    // [wave][N] exchange buffers (wave = K1 * bootstrap + component) = landing words of the hand-over; forward and inverse per-lane
    // twiddle tables; psi^x, x < N (two key bits per step)
    __shared__ __attribute__((aligned(16))) double lds_all[FPW * K1 * N + 2 * N + (GROUP == 2 ? N : 0)];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t sub = wave / (uint32_t)K1, comp = wave - (uint32_t)K1 * sub;   // bootstrap of the workgroup, GLWE component
    const uint32_t t = threadIdx.x & 63u;
    double *mine = lds_all + wave * N;
    double *tables = lds_all + FPW * K1 * N;
    typename W::Xchg xc{mine, 0};
    xc.stride = 0;
    Twiddles twf(tables, a.tw_fwd), twi(tables + N, a.tw_inv);
    for (uint32_t x = threadIdx.x; x < (uint32_t)N; x += 64u * K1 * FPW) {
        tables[x] = a.tw_fwd[W::LANE_TABLE_OFFSET + x];
        tables[N + x] = a.tw_inv[W::LANE_TABLE_OFFSET + x];
        if constexpr (GROUP == 2) tables[2 * N + x] = a.psi_pow[x];
    }
    __syncthreads();

    // a workgroup past the end of a batch that is not a multiple of FPW repeats the last bootstrap (its waves must keep meeting
    // the others at the barriers) and writes nothing
    const size_t f_want = (size_t)blockIdx.x * FPW + sub;
    const bool live = f_want < a.count;
    const size_t f = live ? f_want : a.count - 1;
    size_t gate, ms_row;
    gate_of(a.gv, f, &gate, &ms_row);
    uint32_t table = a.gv.table_ids ? a.gv.table_ids[gate] : 0;
    if (table >= a.n_tables) table = 0;
    const uint32_t *ms = a.ms + ms_row * (a.n + 1);
    const uint64_t *tv = a.tvs + (size_t)table * N;

    double acc[E];   // ACC = (0, .., 0, X^{-b~} * TV), centred; register m of lane t = coefficient t + 64 m
    {
        const uint32_t r = (2u * N - ms[a.n]) & (2u * N - 1u);
#pragma unroll
        for (int m = 0; m < E; m++) {
            const uint32_t idx = (t + (uint32_t)LANES * m - r) & (2u * N - 1u);
            const uint64_t v = tv[idx & (N - 1)];
            acc[m] = comp == (uint32_t)(K1 - 1) ? fp_center(fp_from_u64((idx & N) ? fq_neg(v) : v)) : 0.0;
        }
    }
    // rounding / digit constants: as in k_blind_rotate (abar = round(d / 2^(46 - l beta)) mod B^l, balanced digits as two's
    // complement bit fields)
    const double round_scale = fp_exp2i(-(int)(FQ_BITS - a.l * a.beta));
    const uint32_t bhalf = 1u << (a.beta - 1);
    double round_offset = 0.5 + fp_exp2i((int)(a.l * a.beta));
    uint32_t sign_bits = 0;
    for (uint32_t j = 0; j < a.l; j++) {
        round_offset += (double)(bhalf << (j * a.beta));
        sign_bits |= bhalf << (j * a.beta);
    }
    // the components in rotated order: d stands for component comp + d (mod K1); d = 0 is this wave's own
    uint32_t col_bytes[K1];
    double *land[K1];
#pragma unroll
    for (int d = 0; d < K1; d++) {
        const uint32_t c = comp + (uint32_t)d >= (uint32_t)K1 ? comp + (uint32_t)d - (uint32_t)K1 : comp + (uint32_t)d;
        col_bytes[d] = c * (uint32_t)(N * 8);
        land[d] = lds_all + (sub * (uint32_t)K1 + c) * N;
    }
    const uint32_t rows = (uint32_t)K1 * a.l;                       // rows of a GGSW sample
    const uint32_t t16 = t * 16u;                                   // this thread's 16 bytes of a register pair's 64 lanes
    const uint32_t o_lane = Pos::exponent(Pos::lane(t)) + 1u;       // (two key bits per step: zeta = psi^(o_lane + c_m))
    const double *psi = tables + 2 * N;

    constexpr uint32_t STEP_BITS = GROUP;
    const uint32_t n_steps = a.n / STEP_BITS;
    uint32_t e0_next = ms[0], e1_next = GROUP == 2 ? ms[1] : 0u;
    for (uint32_t i = 0; i < n_steps; i++) {
        uint32_t e[3];
        e[0] = __builtin_amdgcn_readfirstlane(e0_next);
        e[1] = __builtin_amdgcn_readfirstlane(e1_next);
        // (ms has n + 1 entries: the last step reads the body word, or re-reads it, and ignores it)
        e0_next = ms[STEP_BITS * (i + 1) < a.n ? STEP_BITS * (i + 1) : a.n];
        if constexpr (GROUP == 2) e1_next = ms[2 * i + 3 < a.n ? 2 * i + 3 : a.n];
        if (e[0] == 0 && e[1] == 0) {   // nothing to add for this bootstrap: the others of the workgroup still meet their two barriers
            if constexpr (FPW > 1) {
                __syncthreads();
                __syncthreads();
            }
            continue;
        }
        e[2] = (e[0] + e[1]) & (2u * N - 1u);

        // ---- what is decomposed: (X^r - 1) ACC_c (one key bit per step) or ACC_c itself, rounded to the closest multiple of q / B^l ----
        uint32_t digits[E];
        if constexpr (GROUP == 1) {
            W::sync();
#pragma unroll
            for (int m = 0; m < E; m++) mine[t + (uint32_t)LANES * m] = acc[m];
            W::sync();
            const uint32_t from = (t - e[0]) & (2u * N - 1u);   // coefficient t of X^r * ACC is +-ACC[(t - r) mod 2N]
#pragma unroll
            for (int m = 0; m < E; m++) {
                const uint32_t idx = from + (uint32_t)LANES * m;   // < 3N: bit LOGN = sign, bits below = position
                const double w = mine[idx & (N - 1)];
                const double v = __hiloint2double(__double2hiint(w) ^ (int)((idx << (31 - LOGN)) & 0x80000000u), __double2loint(w));
                digits[m] = (uint32_t)__builtin_fma(v - acc[m], round_scale, round_offset) ^ sign_bits;
            }
            W::sync();   // (the transform's stores stay behind these reads)
        } else {
#pragma unroll
            for (int m = 0; m < E; m++) digits[m] = (uint32_t)__builtin_fma(acc[m], round_scale, round_offset) ^ sign_bits;
        }

        // ---- level by level: digit, forward transform, products for the K1 output components (lazy sums over the levels) ---------
        double prod[K1][E];
#pragma unroll
        for (int d = 0; d < K1; d++)
#pragma unroll
            for (int m = 0; m < E; m++) prod[d][m] = 0.0;
        for (int lv = (int)a.l - 1; lv >= 0; lv--) {
            const uint32_t shift = (a.l - 1u - (uint32_t)lv) * a.beta;
            double x[E];
#pragma unroll
            for (int m = 0; m < E; m++) x[m] = (double)(int)__builtin_amdgcn_sbfe(digits[m], shift, a.beta);   // balanced digit in [-B/2, B/2)
            W::template forward<0>(x, xc, t, twf, typename W::NoHook{});
            const uint32_t row_bytes = ((comp * a.l + (uint32_t)lv) * (uint32_t)K1) * (uint32_t)(N * 8);
            if constexpr (GROUP == 1) {
                // row (comp, lv) of the sample of step i: [row][column][N]
                const KeyRows keys(a.bsk_hat + (size_t)FBS_KEY_STEP(i) * rows * K1 * N);
#pragma unroll
                for (int j = 0; j < E / 2; j++) {
                    double2 kw[K1];
#pragma unroll
                    for (int d = 0; d < K1; d++) kw[d] = keys.load(t16 + (uint32_t)(j * LANES * 16), row_bytes + col_bytes[d]);
#pragma unroll
                    for (int d = 0; d < K1; d++) {
                        const double p0 = fp_mulmod(x[2 * j], kw[d].x), p1 = fp_mulmod(x[2 * j + 1], kw[d].y);
                        prod[d][2 * j] += p0;
                        prod[d][2 * j + 1] += p1;
                    }
                }
            } else {
                // the three samples of step i: [sample][row][column][N]
                const KeyRows keys(a.bsk_hat + (size_t)FBS_KEY_STEP(i) * 3 * rows * K1 * N);
                const uint32_t sample_bytes = rows * (uint32_t)K1 * (uint32_t)(N * 8);
#pragma unroll
                for (int j = 0; j < E / 2; j++) {
                    // zeta^e - 1 for the two registers of the pair and the three exponents: zeta = psi^(o_lane + c_m), psi^(x + N) = -psi^x
                    double mono[3][2];
#pragma unroll
                    for (int r = 0; r < 2; r++) {
                        const uint32_t o = o_lane + Pos::exponent(Pos::reg(2 * j + r));
#pragma unroll
                        for (int jj = 0; jj < 3; jj++) {
                            const uint32_t xx = (e[jj] * o) & (2u * N - 1u);
                            const double v = psi[xx & (N - 1)];
                            mono[jj][r] = __hiloint2double(__double2hiint(v) ^ (int)((xx << (31 - LOGN)) & 0x80000000u), __double2loint(v)) - 1.0;
                        }
                    }
#pragma unroll
                    for (int d = 0; d < K1; d++) {
                        double2 kw[3];
#pragma unroll
                        for (int jj = 0; jj < 3; jj++)
                            kw[jj] = keys.load(t16 + (uint32_t)(j * LANES * 16), (uint32_t)jj * sample_bytes + row_bytes + col_bytes[d]);
                        // bundle words: lazy sums of three exact products (< 2.4 q); |x| < 2^49.3, so the products below stay exact
                        double w0 = fp_mulmod(kw[0].x, mono[0][0]), w1 = fp_mulmod(kw[0].y, mono[0][1]);
#pragma unroll
                        for (int jj = 1; jj < 3; jj++) {
                            w0 += fp_mulmod(kw[jj].x, mono[jj][0]);
                            w1 += fp_mulmod(kw[jj].y, mono[jj][1]);
                        }
                        const double p0 = fp_mulmod(x[2 * j], w0), p1 = fp_mulmod(x[2 * j + 1], w1);
                        prod[d][2 * j] += p0;
                        prod[d][2 * j + 1] += p1;
                    }
                }
            }
        }

        // ---- hand the other components theirs: clear, barrier, add, barrier -----------------------------------------------------------
        W::sync();
#pragma unroll
        for (int m = 0; m < E; m++) mine[W::handoff_word(t, m)] = 0.0;
        __syncthreads();
#pragma unroll
        for (int d = 1; d < K1; d++)
#pragma unroll
            for (int m = 0; m < E; m++)
                __hip_atomic_fetch_add(&land[d][W::handoff_word(t, m)], prod[d][m], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __syncthreads();

        // ---- the total for this component ((k + 1) l products below 0.8 q each: centred by the transform first); accumulate ---------
        double own[E];
#pragma unroll
        for (int m = 0; m < E; m++) own[m] = prod[0][m] + mine[W::handoff_word(t, m)];
        W::sync();   // the inverse transform's stores stay behind these reads (same wave, same words)
        W::template inverse<false>(own, xc, t, twi, W::inverse_uniform(t, twi));
#pragma unroll
        for (int m = 0; m < E; m++) acc[m] = fp_center(acc[m] + own[m]);
    }

    // ---- sample extraction of coefficient 0 (k mask polynomials, the body), plus the table's constant -----------------
    if (!live) return;
    if (uint64_t *raw = gate_acc(a.gv, f, K1 * N)) {   // a rotation of TV_0 that several tables share: the whole accumulator
#pragma unroll
        for (int m = 0; m < E; m++) raw[comp * N + t + (uint32_t)LANES * m] = fp_to_u64(fp_canon(acc[m]));
        return;
    }
    uint64_t *out = gate_out(a.gv, f, a.ct_words);
    if (comp < (uint32_t)(K1 - 1)) {
#pragma unroll
        for (int m = 0; m < E; m++) {
            const uint32_t j = t + (uint32_t)LANES * m;
            const uint64_t v = fp_to_u64(fp_canon(acc[m]));
            if (j == 0) out[comp * N] = v;
            else out[comp * N + N - j] = fq_neg(v);
        }
    } else if (t == 0) {
        out[(K1 - 1) * N] = fq_add(fp_to_u64(fp_canon(acc[0])), a.post[table]);
    }
}

// bootstraps per workgroup: twelve waves where the registers allow three waves per SIMD (N <= 512), six to eight at N = 1024
template <int LOGN, int K1>
constexpr int glwe_fpw() {
    return LOGN >= 10 ? 2 : 12 / K1;
}

template <int LOGN, int K1, int GROUP>
static void launch_one(const BrArgs &a, hipStream_t stream, std::string *kernel) {
    constexpr int FPW = glwe_fpw<LOGN, K1>();
    *kernel = "k_blind_rotate_glwe<" + std::to_string(LOGN) + "," + std::to_string(K1) + "," + std::to_string(GROUP) + ">";
    hipLaunchKernelGGL((k_blind_rotate_glwe<LOGN, K1, GROUP, FPW>), dim3((unsigned)((a.count + FPW - 1) / FPW)), dim3(64 * K1 * FPW), 0, stream, a);
}

// the shapes built: k = 2, 3, 4 at N = 256 and 512, k = 2, 3 at N = 1024; one or two key bits per step
#define FBS_GLWE_SHAPES(X) X(8, 3) X(8, 4) X(8, 5) X(9, 3) X(9, 4) X(9, 5) X(10, 3) X(10, 4)

bool glwe_shape_built(uint32_t log_n, uint32_t k) {
#define X(L, K) \
    if (log_n == L && k + 1 == K) return true;
    FBS_GLWE_SHAPES(X)
#undef X
    return false;
}

bool launch_blind_rotate_glwe(fbs_ctx *ctx, const BrArgs &a, hipStream_t stream, std::string *kernel) {
    const fbs_params &p = ctx->p;
    if (p.k < 2 || !glwe_shape_built(p.log_n_poly, p.k)) return false;
#define X(L, K)                                                      \
    if (p.log_n_poly == L && p.k + 1 == K) {                         \
        if (ctx->group == 2) launch_one<L, K, 2>(a, stream, kernel); \
        else launch_one<L, K, 1>(a, stream, kernel);                 \
        return true;                                                 \
    }
    FBS_GLWE_SHAPES(X)
#undef X
    return false;
}

void blind_rotate_glwe_catalog(std::vector<std::string> *out) {
#define X(L, K)                                                                                                   \
    out->push_back("k_blind_rotate_glwe<" + std::to_string(L) + "," + std::to_string(K) + ",1>");                 \
    out->push_back("k_blind_rotate_glwe<" + std::to_string(L) + "," + std::to_string(K) + ",2>");
    FBS_GLWE_SHAPES(X)
#undef X
}

}  // namespace fbs

// Blind rotation with GLWE dimension k = 2 at N = 1024, two key bits per step, one gadget level: the shape that carries the
// 128-bit sets N = 2048 carries at k = 1 (same k N, the same noise floor) on THREE wave-private 1024-point transforms each way per
// step instead of two 2048-point ones spread over wave pairs -- a third fewer butterflies, and no workgroup barrier inside a
// transform (the structure of the benchmark kernel k_blind_rotate<10,6,3,4>, which issues 0.79-0.80 of the FP64 peak where the
// two-waves-per-polynomial kernels issue 0.69-0.71).  gfx950 only; the algebra is k_blind_rotate_pairs' (fbs_blind_rotate.hip):
//     ACC += [ (X^a0 - 1) E0 + (X^a1 - 1) E1 + (X^(a0+a1) - 1) E2 ]  (x)  ACC
// with the bundle built in the transform domain and ACC itself decomposed; same rounding rules and, word for word, the same
// ciphertexts as the k = 1 kernels' conventions give at k = 2 (tests/test_gpu_k2.py).
//
// One bootstrap = three waves (component c = wave c of the bootstrap: two mask polynomials and the body).  Three does not divide
// the eight 256-register waves a CU holds, so the kernel is written for THREE waves per SIMD: 168 registers, twelve waves = four
// bootstraps per CU in ONE workgroup (so that the twiddle tables and the psi table are shared: 12 x 8 KB of exchange buffers +
// 24 KB of tables = 120 KB of LDS).  What makes 168 registers enough: a component's products for the three output components
// (48 lazy sums) are never held.  When its forward transform is done a wave clears its exchange buffer; after a barrier every wave
// ADDS its products straight into the three components' buffers with ds_add_f64 -- exact on integer-valued doubles below 2^53,
// so the order in which the three additions land does not matter -- and after a second barrier each wave reads its total back
// and inverts it.  Two workgroup barriers per step, none inside the transforms.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "fbs_blind_rotate.hpp"

namespace fbs {

// FPW: bootstraps per workgroup.  4 = the throughput shape above (twelve waves, three per SIMD); 1 and 2 = launches that leave
// most of the chip empty (at most one / two bootstraps per CU): the same code with one or two bootstraps' three waves on a CU's
// four SIMDs, every wave (nearly) alone on its SIMD -- a step's latency is then one wave's instruction chain instead of three
// waves' sharing an issue port.
template <int LOGN, int FPW>
__global__ __launch_bounds__(192 * FPW) void k_blind_rotate_pairs_k2(BrArgs a) {
    using W = SplitNtt<LOGN, 6>;
    static_assert(W::HAS_EVAL_POSITION && W::E == 16, "one wave per polynomial, 16 coefficients per lane");
    constexpr int N = W::N, E = W::E, LANES = W::LANES, K1 = 3;
    constexpr int GLOG = W::EVAL_GROUP_LOG2, G = 1 << GLOG;
    // [wave][N] exchange buffers (wave = 3 * bootstrap + component), forward and inverse per-lane twiddle tables, psi^x (x < N,
    // transposed as in k_blind_rotate_pairs: word (x mod G) N/G + x / G)
    __shared__ __attribute__((aligned(16384))) double lds_all[FPW * K1 * N + 3 * N];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // 0 .. 11 (wave-uniform, and known as such)
    const uint32_t sub = wave / 3u, comp = wave - 3u * sub;                   // bootstrap of the workgroup, GLWE component
    const uint32_t t = threadIdx.x & 63u;
    double *mine = lds_all + wave * N;
    double *tables = lds_all + FPW * K1 * N;
    typename W::Xchg xc{mine, 0};
    xc.stride = 0;
    Twiddles twf(tables, a.tw_fwd), twi(tables + N, a.tw_inv);
    for (uint32_t x = threadIdx.x; x < (uint32_t)N; x += 192u * FPW) {
        tables[x] = a.tw_fwd[W::LANE_TABLE_OFFSET + x];
        tables[N + x] = a.tw_inv[W::LANE_TABLE_OFFSET + x];
        tables[2 * N + (x & (G - 1)) * (N / G) + (x >> GLOG)] = a.psi_pow[x];
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

    double acc[E];   // ACC = (0, 0, X^{-b~} * TV), centred; register m of lane t = coefficient t + 64 m
    {
        const uint32_t r = (2u * N - ms[a.n]) & (2u * N - 1u);
#pragma unroll
        for (int m = 0; m < E; m++) {
            const uint32_t idx = (t + (uint32_t)LANES * m - r) & (2u * N - 1u);
            const uint64_t v = tv[idx & (N - 1)];
            acc[m] = comp == 2u ? fp_center(fp_from_u64((idx & N) ? fq_neg(v) : v)) : 0.0;
        }
    }
    // rounding / digit constants: as in k_blind_rotate (one level: abar = round(acc / 2^(46 - beta)) mod B, balanced)
    const double round_scale = fp_exp2i(-(int)(FQ_BITS - a.beta));
    const uint32_t bhalf = 1u << (a.beta - 1);
    const double round_offset = 0.5 + fp_exp2i((int)a.beta) + (double)bhalf;
    // zeta^e for the evaluation point a register holds: as in k_blind_rotate_pairs
    const uint32_t o_lane = 2u * (__builtin_bitreverse32(W::eval_position_lane(t)) >> (32 - LOGN)) + 1u;
    const uint32_t k_lane8 = (o_lane >> GLOG) << 3;
    const uint32_t r_lane = __builtin_amdgcn_readfirstlane(o_lane & (G - 1));
    constexpr uint32_t MASK8 = (uint32_t)(N / G - 1) << 3;
    const uint32_t psi_base = (uint32_t)(uintptr_t)(tables + 2 * N);   // LDS byte address of the table (a multiple of its 8 KB)
    // where the products for component c' of this bootstrap are added: that wave's exchange buffer
    double *land[K1];
#pragma unroll
    for (int c = 0; c < K1; c++) land[c] = lds_all + (sub * 3u + (uint32_t)c) * N;

    const uint32_t t16 = t * 16u;   // this thread's 16 bytes of a register pair's 16 LANES
    const uint32_t n_pairs = a.n / 2;
    uint32_t e0_next = ms[0], e1_next = ms[1];
    for (uint32_t i = 0; i < n_pairs; i++) {
        uint32_t e[3];
        e[0] = __builtin_amdgcn_readfirstlane(e0_next);
        e[1] = __builtin_amdgcn_readfirstlane(e1_next);
        e0_next = ms[2 * i + 2 < a.n ? 2 * i + 2 : a.n];   // (the last pair re-reads the body word and ignores it)
        e1_next = ms[2 * i + 3 < a.n ? 2 * i + 3 : a.n];
        if (e[0] == 0 && e[1] == 0) {   // the bundle is zero for this bootstrap: the other three still meet their two barriers
            __syncthreads();
            __syncthreads();
            continue;
        }
        e[2] = (e[0] + e[1]) & (2u * N - 1u);
        uint32_t lane8[3];
#pragma unroll
        for (int jj = 0; jj < 3; jj++) lane8[jj] = __umul24(e[jj], k_lane8);

        // ---- ACC_c itself, rounded to the closest multiple of q / B; balanced digit; forward transform -------------
        double x[E];
#pragma unroll
        for (int m = 0; m < E; m++) {
            const uint32_t d = (uint32_t)__builtin_fma(acc[m], round_scale, round_offset) ^ bhalf;
            x[m] = (double)(int)__builtin_amdgcn_sbfe(d, 0, a.beta);
        }
        W::template forward<0>(x, xc, t, twf, typename W::NoHook{});
        // ---- clear the landing words (this wave's own buffer: its transform is done with it) ------------------------
        W::sync();
#pragma unroll
        for (int m = 0; m < E; m++) mine[W::handoff_word(t, m)] = 0.0;
        __syncthreads();

        // ---- bundle x digits, register pair by register pair, added into the three components' buffers ----------------
        // row `comp` of the three samples of step i: [sample][row][column][N]; column c' = the products for component c'
        const KeyRows keys(a.bsk_hat + ((size_t)i * 3 * K1 + comp) * K1 * N);
#pragma unroll
        for (int j = 0; j < E / 2; j++) {
            double2 kw[3][K1];
#pragma unroll
            for (int jj = 0; jj < 3; jj++)
#pragma unroll
                for (int c = 0; c < K1; c++)
                    kw[jj][c] = keys.load(t16 + (uint32_t)(j * LANES * 16), ((uint32_t)(jj * K1 * K1 + c)) * (uint32_t)(N * 8));
            // zeta^e - 1 for the two registers of the pair and the three exponents (one look-up per exponent: the registers'
            // evaluation points differ by psi^N = -1)
            static_assert(W::eval_position_reg(1) - W::eval_position_reg(0) == 1, "registers 2j, 2j+1 hold neighbouring array positions");
            double mono[3][2];
#pragma unroll
            for (int jj = 0; jj < 3; jj++) {
                const uint32_t c_m = 2u * (__builtin_bitreverse32(W::eval_position_reg(2 * j)) >> (32 - LOGN));
                const uint32_t rsum = r_lane + (c_m & (G - 1));                       // uniform from here ...
                const uint32_t ku = (c_m >> GLOG) + (rsum >> GLOG), ru = rsum & (G - 1);
                const uint32_t eru = e[jj] * ru;
                const uint32_t u8 = (e[jj] * ku + (eru >> GLOG)) << 3;
                const uint32_t sbase = psi_base + (eru & (G - 1)) * (uint32_t)(N / G * 8);
                const uint32_t odd = e[jj] << 31;                                      // ... to here
                const uint32_t t8 = lane8[jj] + u8;
                const double v = *reinterpret_cast<const __attribute__((address_space(3))) double *>((t8 & MASK8) | sbase);
                const int hi = __double2hiint(v) ^ (int)((t8 << (31 - 3 - (LOGN - GLOG))) & 0x80000000u);
                mono[jj][0] = __hiloint2double(hi, __double2loint(v)) - 1.0;
                mono[jj][1] = __hiloint2double(hi ^ (int)odd, __double2loint(v)) - 1.0;
            }
#pragma unroll
            for (int c = 0; c < K1; c++) {
                // bundle words: lazy sums of three exact products (< 2.4 q); |x| < 2^49.3, so the products below stay exact
                double w0 = fp_mulmod(kw[0][c].x, mono[0][0]), w1 = fp_mulmod(kw[0][c].y, mono[0][1]);
#pragma unroll
                for (int jj = 1; jj < 3; jj++) {
                    w0 += fp_mulmod(kw[jj][c].x, mono[jj][0]);
                    w1 += fp_mulmod(kw[jj][c].y, mono[jj][1]);
                }
                const double p0 = fp_mulmod(x[2 * j], w0), p1 = fp_mulmod(x[2 * j + 1], w1);
                __hip_atomic_fetch_add(&land[c][W::handoff_word(t, 2 * j)], p0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&land[c][W::handoff_word(t, 2 * j + 1)], p1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        __syncthreads();

        // ---- the total of the three components' products for this component; back to coefficients; accumulate ----------
        double own[E];
#pragma unroll
        for (int m = 0; m < E; m++) own[m] = mine[W::handoff_word(t, m)];
        W::sync();   // the inverse transform's stores stay behind these reads (same wave, same words)
        W::template inverse<true>(own, xc, t, twi, W::inverse_uniform(t, twi));   // three products below 0.8 q each: within 8 q
#pragma unroll
        for (int m = 0; m < E; m++) acc[m] = fp_center(acc[m] + own[m]);
    }

    // ---- sample extraction of coefficient 0 (two mask polynomials, the body), plus the table's constant -----------------
    if (!live) return;
    uint64_t *out = gate_out(a.gv, f, a.ct_words);
    if (comp < 2u) {
#pragma unroll
        for (int m = 0; m < E; m++) {
            const uint32_t j = t + (uint32_t)LANES * m;
            const uint64_t v = fp_to_u64(fp_canon(acc[m]));
            if (j == 0) out[comp * N] = v;
            else out[comp * N + N - j] = fq_neg(v);
        }
    } else if (t == 0) {
        out[2 * N] = fq_add(fp_to_u64(fp_canon(acc[0])), a.post[table]);
    }
}

// k = 2: N = 1024, two key bits per step, one gadget level (what dev_supported admits).  Returns false when the context is not
// of that shape.
bool launch_blind_rotate_k2(fbs_ctx *ctx, const BrArgs &a, hipStream_t stream, std::string *kernel) {
    const fbs_params &p = ctx->p;
    if (p.k != 2 || p.log_n_poly != 10 || ctx->group != 2 || p.l_bsk != 1) return false;
    // up to one bootstrap per CU: one per workgroup; up to two: two; beyond: four (measured per launch: tools/k2_check.py)
    const size_t cus = (size_t)ctx->cu_count;
    if (a.count <= cus && ctx->tune.br_cu_max_per_cu >= 1) {
        *kernel = "k_blind_rotate_pairs_k2<10,1>";
        hipLaunchKernelGGL((k_blind_rotate_pairs_k2<10, 1>), dim3((unsigned)a.count), dim3(192), 0, stream, a);
    } else if (a.count <= 2 * cus && ctx->tune.br_cu_max_per_cu >= 1) {
        *kernel = "k_blind_rotate_pairs_k2<10,2>";
        hipLaunchKernelGGL((k_blind_rotate_pairs_k2<10, 2>), dim3((unsigned)((a.count + 1) / 2)), dim3(384), 0, stream, a);
    } else {
        *kernel = "k_blind_rotate_pairs_k2<10,4>";
        hipLaunchKernelGGL((k_blind_rotate_pairs_k2<10, 4>), dim3((unsigned)((a.count + 3) / 4)), dim3(768), 0, stream, a);
    }
    return true;
}

void blind_rotate_k2_catalog(std::vector<std::string> *out) {
    for (const char *name : {"k_blind_rotate_pairs_k2<10,1>", "k_blind_rotate_pairs_k2<10,2>", "k_blind_rotate_pairs_k2<10,4>"}) out->push_back(name);
}

}  // namespace fbs

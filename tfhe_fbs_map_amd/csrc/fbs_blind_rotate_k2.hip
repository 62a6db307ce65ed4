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

#include <algorithm>
#include <type_traits>

#include "fbs_blind_rotate_cu.hpp"

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

// ---------------------------------------------------------------------------------------------
// The LATENCY shape: ONE k = 2 bootstrap on a whole workgroup of twelve (LOGW = 2) or six (LOGW = 1) waves -- what a launch that
// leaves most of the chip empty wants (a rank's slice of a level of a sharded circuit, a narrow level): a bootstrap is n / 2
// dependent steps, and in the three-waves-per-bootstrap kernel above each step is one wave's 2 965 instructions whatever the
// chip has idle (3.4-3.6 ms per launch at n = 734-760).  Here each of the three GLWE components is dealt over 2^LOGW waves as in
// k_blind_rotate_cu_pairs (fbs_blind_rotate_cu.hip; WavesNtt<10, LOGW>): LOGW cross stages in registers, ONE trip through LDS to
// re-deal, then wave-private part transforms (LaneNtt256 at 4 coefficients per lane / LaneNtt512 at 8: register <-> lane
// transpositions, one LDS exchange each).  Per step, component c = waves c 2^LOGW ..:
//   ACC_c itself -> balanced digit -> cross stages -> re-deal (barrier 1) -> private forward transform
//   -> bundle of row c (three samples x three columns: nine key polynomials' words, fetched register pair by register pair)
//      times the digits: products for all three output components; the own one stays in registers, the other two are HANDED
//      to the waves that hold the same evaluation points of those components -- plain stores into the receiver's two slots
//      (barrier 2), no atomics: a wave has exactly two senders
//   -> sum, private inverse transform, re-deal back (barrier 3), joining stages, accumulate.
// Three workgroup barriers per step, as the k = 1 shape.  Evaluation points: array position P = M w + j of a part holds the value
// at psi^(2 bitrev10(P) + 1), and after the part's forward transform the register index is the LOW bits of j, i.e. the HIGH bits
// of the bit reversal: 2 bitrev(P) + 1 = o_lane + (2N / E) k_m with psi^(2N/E) a primitive E-th root of unity (E = 4: i; E = 8: an
// eighth root) -- ONE table look-up per lane and exponent, then a wave-uniform choice among its products with that root's powers.
// LDS (doubles): [3][N] re-deal + private forward exchange; [3][2][N] hand-over slots -- slot 0 doubles as the private inverse
// exchange and the re-deal back (wave w receives in the words it alone touches until the re-deal, as in k_blind_rotate_cu_pairs);
// the inverse per-lane twiddles of 512-point parts: 72 KB (LOGW = 2), 89 KB (LOGW = 1).
template <int LOGW>
__global__ __launch_bounds__(192 << LOGW) void k_blind_rotate_cu_k2(BrArgs a) {
    constexpr int LOGN = 10, K1 = 3, PARTS = 1 << LOGW;
    using W = WavesNtt<LOGN, LOGW>;
    using Part = typename W::Half;
    constexpr int N = W::N, E = W::E, LANES = W::LANES, M = W::M, EP = W::EP, LOGE = W::LOGE;
    static_assert((LOGW == 2 && std::is_same<Part, LaneNtt256>::value && E == 4) || (LOGW == 1 && std::is_same<Part, LaneNtt512>::value && E == 8),
                  "four waves per polynomial at 4 coefficients per lane, or two at 8");
    using Tw = CuTwiddles<Part, false, PARTS>;
    __shared__ double lds_all[K1 * N + K1 * 2 * N + Tw::LDS_WORDS];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t comp = wave >> LOGW, w = wave & (PARTS - 1u);            // GLWE component, part of it (both wave-uniform)
    const uint32_t t = threadIdx.x & (LANES - 1), ln = t & 63u;             // thread of the component, lane
    double *xf = lds_all + comp * N;                                        // re-deal region of this component
    double *hand = lds_all + K1 * N;                                        // [component][slot][N]
    double *back = hand + comp * 2 * N;                                     // slot 0 of this component: [part][M]
    double *mine = back + w * M;                                            // ... this wave's part of it
    const double *mine2 = back + N + w * M;                                 // ... and of slot 1
    // where this wave's products for the two other components go: the same part of their slots.  Component c receives from
    // c + 2 in slot 0 and from c + 1 in slot 1 (mod 3).  c1 = comp + 1, c2 = comp + 2 (mod 3)
    const uint32_t c1 = comp == 2u ? 0u : comp + 1u, c2 = comp == 0u ? 2u : comp - 1u;
    double *to_c1 = hand + (c1 * 2u + 0u) * N + w * M;                      // c1 = comp + 1 receives from comp = c1 + 2: slot 0
    double *to_c2 = hand + (c2 * 2u + 1u) * N + w * M;                      // c2 = comp + 2 receives from comp = c2 + 1: slot 1
    const uniform_doubles big_f = (uniform_doubles)(uintptr_t)a.tw_fwd, big_i = (uniform_doubles)(uintptr_t)a.tw_inv;
    Tw tw;
    tw.init(big_f, big_i, a.tw_fwd + W::LANE_TABLE_OFFSET, a.tw_inv + W::LANE_TABLE_OFFSET, w, ln, lds_all + 3 * K1 * N);

    const bool live = (size_t)blockIdx.x < a.count;
    const size_t f = live ? (size_t)blockIdx.x : a.count - 1;
    size_t gate, ms_row;
    gate_of(a.gv, f, &gate, &ms_row);
    uint32_t table = a.gv.table_ids ? a.gv.table_ids[gate] : 0;
    if (table >= a.n_tables) table = 0;
    const uint32_t *ms = a.ms + ms_row * (a.n + 1);
    const uint64_t *tv = a.tvs + (size_t)table * N;

    double acc[E];   // ACC = (0, 0, X^{-b~} * TV), centred; register m of thread t = coefficient t + LANES m
    {
        const uint32_t r = (2u * N - ms[a.n]) & (2u * N - 1u);
#pragma unroll
        for (int m = 0; m < E; m++) {
            const uint32_t idx = (t + (uint32_t)LANES * m - r) & (2u * N - 1u);
            const uint64_t v = tv[idx & (N - 1)];
            acc[m] = comp == 2u ? fp_center(fp_from_u64((idx & N) ? fq_neg(v) : v)) : 0.0;
        }
    }
    // rounding / digit constants: as in k_blind_rotate_pairs_k2 (one level)
    const double round_scale = fp_exp2i(-(int)(FQ_BITS - a.beta));
    const uint32_t bhalf = 1u << (a.beta - 1);
    const double round_offset = 0.5 + fp_exp2i((int)a.beta) + (double)bhalf;
    // cross-stage twiddles: nodes 1 .. PARTS - 1 of the big tree (wave-uniform)
    double cw[PARTS - 1], iw[PARTS - 1];
#pragma unroll
    for (int i = 0; i < PARTS - 1; i++) cw[i] = big_f[1 + i], iw[i] = big_i[1 + i];
    // o_lane = 2 bitrev10(M w + the lane's part of j) + 1; the register part of j is its low log2(E) bits
    const uint32_t j_lane = LOGW == 2 ? (((ln & 15u) << 4) | ((ln >> 4) << 2)) : (((ln & 31u) << 4) | ((ln >> 5) << 3));
    const uint32_t o_lane = 2u * (__builtin_bitreverse32((uint32_t)M * w + j_lane) >> (32 - LOGN)) + 1u;
    const uniform_doubles psi_u = (uniform_doubles)(uintptr_t)a.psi_pow;
    constexpr uint32_t ROOT = 2u * N / E;                                   // psi^ROOT: a primitive E-th root of unity
    const double om1 = psi_u[ROOT], om2 = E == 8 ? psi_u[2 * ROOT] : 0.0, om3 = E == 8 ? psi_u[3 * ROOT] : 0.0;
    __syncthreads();   // (the inverse twiddle table of 512-point parts is in place)

    const uint32_t t16 = t * 16u;   // this thread's 16 bytes of a register pair's 16 LANES
    const uint32_t n_pairs = a.n / 2;
    uint32_t e0_next = ms[0], e1_next = ms[1];
    for (uint32_t i = 0; i < n_pairs; i++) {
        uint32_t e[3];
        e[0] = __builtin_amdgcn_readfirstlane(e0_next);
        e[1] = __builtin_amdgcn_readfirstlane(e1_next);
        e0_next = ms[2 * i + 2 < a.n ? 2 * i + 2 : a.n];   // (the last pair re-reads the body word and ignores it)
        e1_next = ms[2 * i + 3 < a.n ? 2 * i + 3 : a.n];
        if (e[0] == 0 && e[1] == 0) continue;               // the bundle is zero (uniform over the workgroup: one bootstrap)
        e[2] = (e[0] + e[1]) & (2u * N - 1u);

        // ---- what memory has to bring: psi^(e o_lane) for the three exponents, and the first register pair's key words --------
        double A[3];
#pragma unroll
        for (int jj = 0; jj < 3; jj++) {
            const uint32_t x = __umul24(e[jj], o_lane) & (2u * N - 1u);
            const double v = a.psi_pow[x & (N - 1)];
            A[jj] = __hiloint2double(__double2hiint(v) ^ (int)((x << (31 - LOGN)) & 0x80000000u), __double2loint(v));   // psi^(x + N) = -psi^x
        }
        // row `comp` of the three samples of step i: [sample][row][column][N]; column c' = the products for component c'.  The
        // columns are taken in ROTATED order -- d = 0, 1, 2 stands for component comp + d (mod 3): own, next, next but one -- which
        // costs nothing (the column is the load's scalar offset) and makes "which product goes where" the same code in every wave
        const KeyRows keys(a.bsk_hat + ((size_t)i * 3 * K1 + comp) * K1 * N);
        const uint32_t col_bytes[K1] = {comp * (uint32_t)(N * 8), c1 * (uint32_t)(N * 8), c2 * (uint32_t)(N * 8)};
        auto request = [&](auto jc, double2 (&k)[3][K1]) {
            constexpr int j = decltype(jc)::value;
#pragma unroll
            for (int jj = 0; jj < 3; jj++)
#pragma unroll
                for (int d = 0; d < K1; d++)
                    k[jj][d] = keys.load(t16 + (uint32_t)(j * LANES * 16), (uint32_t)(jj * K1 * K1) * (uint32_t)(N * 8) + col_bytes[d]);
        };
        double2 kw[3][K1];
        request(std::integral_constant<int, 0>{}, kw);

        // ---- ACC_c itself, rounded to the closest multiple of q / B; the cross stages; re-deal; private transform ---------------
        double x[1][E];
#pragma unroll
        for (int m = 0; m < E; m++) {
            const uint32_t d = (uint32_t)__builtin_fma(acc[m], round_scale, round_offset) ^ bhalf;
            x[0][m] = (double)(int)__builtin_amdgcn_sbfe(d, 0, a.beta);
        }
#pragma unroll
        for (int s = 0; s < LOGW; s++) {
            const int half = (E / 2) >> s;
#pragma unroll
            for (int m = 0; m < E; m++) {
                if (m & half) continue;
                const double wv = cw[(1 << s) - 1 + (m >> (LOGE - s))];
                if (s == 0) {
                    first_butterfly<0>(x[0][m], x[0][m + half], wv);
                } else {
                    const double u = x[0][m], v = fp_mulmod(x[0][m + half], wv);
                    x[0][m] = u + v;
                    x[0][m + half] = u - v;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < PARTS; q++)
#pragma unroll
            for (int r = 0; r < EP; r++) xf[q * M + t + (uint32_t)LANES * r] = x[0][q * EP + r];
        __syncthreads();
        double *bufs[1] = {xf + w * M};   // the words only this wave reads: its private exchange buffer from here on
#pragma unroll
        for (int m = 0; m < E; m++) x[0][m] = bufs[0][ln + 64u * m];
        tw.template forward<1>(x, bufs, ln);

        // ---- the monomial factors: psi^(e o_lane) times the powers of the E-th root the registers differ by -------------------
        struct Powers {
            double s0, s1, s2, s3;
        };
        auto powers = [&](double base) {
            if constexpr (E == 8) return Powers{base, fp_mulmod(base, om1), fp_mulmod(base, om2), fp_mulmod(base, om3)};
            else return Powers{base, fp_mulmod(base, om1), 0.0, 0.0};
        };
        const Powers V0 = powers(A[0]), V1 = powers(A[1]), V2 = powers(A[2]);
        // zeta^e - 1 for register m.  E = 8: m = (r2 r1 r0), k_m = r1 + 2 r0 + 4 r2, root^(t + 4) = -root^t;
        // E = 4: m = (j1 j0), k_m = j1 + 2 j0, root^(t + 2) = -root^t
        auto mono = [&](const Powers &V, uint32_t ej, int m) {
            if constexpr (E == 8) {
                const uint32_t km = (uint32_t)(((m >> 1) & 1) | ((m & 1) << 1) | (m & 4));
                const uint32_t tt = (ej * km) & 7u;                                  // wave-uniform
                const double v = (tt & 2u) ? ((tt & 1u) ? V.s3 : V.s2) : ((tt & 1u) ? V.s1 : V.s0);
                return ((tt & 4u) ? -v : v) - 1.0;
            } else {
                const uint32_t km = (uint32_t)((m >> 1) | ((m & 1) << 1));
                const uint32_t tt = (ej * km) & 3u;
                const double v = (tt & 1u) ? V.s1 : V.s0;
                return ((tt & 2u) ? -v : v) - 1.0;
            }
        };
        // ---- bundle x digits, register pair by register pair: products for the three output components (rotated order) ---------
        double prod[K1][E];
        auto consume = [&](auto jc, const double2 (&k)[3][K1]) {
            constexpr int j = decltype(jc)::value;
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int m = 2 * j + r;
                const double mo[3] = {mono(V0, e[0], m), mono(V1, e[1], m), mono(V2, e[2], m)};
#pragma unroll
                for (int c = 0; c < K1; c++) {
                    // bundle word: lazy sum of three exact products (< 2.4 q); |x| < 2^49.3, so the product below stays exact
                    double wsum = fp_mulmod(r ? k[0][c].y : k[0][c].x, mo[0]);
#pragma unroll
                    for (int jj = 1; jj < 3; jj++) wsum += fp_mulmod(r ? k[jj][c].y : k[jj][c].x, mo[jj]);
                    prod[c][m] = fp_mulmod(x[0][m], wsum);
                }
            }
        };
#define FBS_K2_PAIR_STEP(J)                                                                                      \
    if constexpr ((J) < E / 2) {                                                                                 \
        consume(std::integral_constant<int, (J) < E / 2 ? (J) : 0>{}, kw);                                       \
        __builtin_amdgcn_sched_barrier(0);   /* (or every request is hoisted to the top, and spilled) */         \
        if constexpr ((J) + 1 < E / 2) request(std::integral_constant<int, ((J) + 1) % (E / 2)>{}, kw);          \
    }
        FBS_K2_PAIR_STEP(0) FBS_K2_PAIR_STEP(1) FBS_K2_PAIR_STEP(2) FBS_K2_PAIR_STEP(3)
#undef FBS_K2_PAIR_STEP

        // ---- hand the other two components theirs; sum; private inverse; re-deal back; joining stages; accumulate ---------------
#pragma unroll
        for (int m = 0; m < E; m++) {
            to_c1[64u * m + ln] = prod[1][m];
            to_c2[64u * m + ln] = prod[2][m];
        }
        __syncthreads();
        double own[E];
#pragma unroll
        for (int m = 0; m < E; m++) own[m] = prod[0][m] + mine[64u * m + ln] + mine2[64u * m + ln];
        tw.inverse(own, mine, ln, typename Part::NoHook{});   // three products below 0.8 q each: centred first by the transform
        Part::sync();
#pragma unroll
        for (int m = 0; m < E; m++) mine[ln + 64u * m] = own[m];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PARTS; q++)
#pragma unroll
            for (int r = 0; r < EP; r++) own[q * EP + r] = back[q * M + t + (uint32_t)LANES * r];
#pragma unroll
        for (int s = LOGW - 1; s >= 0; s--) {
            const int half = (E / 2) >> s;
#pragma unroll
            for (int m = 0; m < E; m++) {
                if (m & half) continue;
                const double u = own[m], v = own[m + half];
                own[m] = u + v;
                own[m + half] = fp_mulmod(u - v, iw[(1 << s) - 1 + (m >> (LOGE - s))]);
            }
        }
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
    const size_t cus = (size_t)ctx->cu_count;
    // tune.br_k2_shape: 0 = by launch size (below); 3 = always the three-waves-per-bootstrap kernels; 12 / 6 = always the
    // twelve- / six-wave whole-workgroup shape (A/B measurements, dispatch tests)
    const int64_t shape = ctx->tune.br_k2_shape;
    const bool small = a.count <= cus * (size_t)ctx->tune.br_cu_max_per_cu && ctx->tune.br_cu_kernel;
    if ((shape == 12 || (shape == 0 && small)) && ctx->d_bsk_hat_small) {
        // launches that leave most of the chip empty: one bootstrap on twelve waves (round after round beyond one per CU)
        BrArgs b = a;
        b.bsk_hat = reinterpret_cast<const double *>(ctx->d_bsk_hat_small);
        *kernel = "k_blind_rotate_cu_k2<2>";
        hipLaunchKernelGGL((k_blind_rotate_cu_k2<2>), dim3((unsigned)a.count), dim3(768), 0, stream, b);
        return true;
    }
    if (shape == 6 && ctx->d_bsk_hat_mid) {
        BrArgs b = a;
        b.bsk_hat = reinterpret_cast<const double *>(ctx->d_bsk_hat_mid);
        *kernel = "k_blind_rotate_cu_k2<1>";
        hipLaunchKernelGGL((k_blind_rotate_cu_k2<1>), dim3((unsigned)a.count), dim3(384), 0, stream, b);
        return true;
    }
    // the three-waves-per-bootstrap kernel: up to one bootstrap per CU one per workgroup; up to two: two; beyond: four
    // (measured per launch: tools/k2_check.py)
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

// the key in the evaluation order of WavesNtt<10, 1> (two waves per polynomial, 512-point parts): what k_blind_rotate_cu_k2<1> reads
__global__ __launch_bounds__(128) void k_bsk_transform_two_waves(const uint64_t *__restrict__ src, double *__restrict__ dst,
                                                                 const double *__restrict__ tw_fwd, double n_inv, size_t polys) {
    using W = WavesNtt<10, 1>;
    __shared__ double lds[2 * W::N];
    const uint32_t t = threadIdx.x;
    typename W::Xchg xc{lds, 0};
    for (size_t p = blockIdx.x; p < polys; p += gridDim.x) {   // uniform trip count per workgroup
        double x[W::E];
#pragma unroll
        for (int m = 0; m < W::E; m++) x[m] = fp_from_u64(src[p * W::N + W::template index_of<0>(t, m)]);
        W::forward(x, xc, t, Twiddles(tw_fwd + W::LANE_TABLE_OFFSET, tw_fwd));
#pragma unroll
        for (int m = 0; m < W::E; m++) dst[p * W::N + W::key_word(t, m)] = fp_center(fp_mulmod(x[m], n_inv));
    }
}
void launch_bsk_transform_two_waves(const uint64_t *d_src, double *d_dst, const double *tw_fwd, double n_inv, size_t polys, hipStream_t stream) {
    hipLaunchKernelGGL(k_bsk_transform_two_waves, dim3((unsigned)std::min<size_t>(polys, 4096)), dim3(128), 0, stream, d_src, d_dst, tw_fwd, n_inv, polys);
}

void blind_rotate_k2_catalog(std::vector<std::string> *out) {
    for (const char *name : {"k_blind_rotate_pairs_k2<10,1>", "k_blind_rotate_pairs_k2<10,2>", "k_blind_rotate_pairs_k2<10,4>", "k_blind_rotate_cu_k2<2>",
                             "k_blind_rotate_cu_k2<1>"})
        out->push_back(name);
}

}  // namespace fbs

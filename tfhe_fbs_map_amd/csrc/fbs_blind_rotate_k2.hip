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
// (Round 4, measured and not adopted: TWO six-wave workgroups per CU instead of one of twelve waves (<10,2> at every size: 72 KB of LDS
// and 166 registers each would allow it) -- 9.5 against 7.2 ms per 1 024; the next step's first register pair of key words requested behind the read-back, as the
// twelve-wave shape below does -- 168 registers with 160 bytes spilled instead of 12: 7.60 against 7.21 ms per 1 024; and the
// monomial factors as psi^(e o_lane) (ONE gather per exponent) times the wave-uniform psi^(e c_m) read through the scalar cache, in
// the place of a gather from the table in LDS per register pair and exponent: 7.40 against 7.24 ms, same box, twice.
// And the three waves of ONE bootstrap meeting on a counter in LDS (release, ds_add, spin with s_sleep, acquire) in the place of the
// workgroup's s_barrier, so that a bootstrap's waits are filled by the other three: 9.47 against 7.25 ms per 1 024, 18.7 against 14.4
// per 2 048 -- the same loss as the two six-wave workgroups.  What both give up is the LOCK STEP of the four bootstraps of a CU: they
// stream the same 221 KB of key words per step, and only while they ask for them together does one of them pull a line out of L2 and
// the other three find it in the CU's 32 KB of L1.  profiles/r04/k2_boot_meet_ab.txt.
// Issue priority by turns among the three waves of a SIMD (s_setprio 2 / 1 / 0 or 2 / 0 / 0, rotated every step or every half
// step -- what gives the benchmark kernel's TWO waves per SIMD 10 %): 7.28-7.36 against 7.30 ms per 1 024, nothing either way
// (profiles/r04/k2_turns_ab.txt); and with every step reading the same two key rows (-DFBS_EXP_HOT_KEYS) 7.18 against 7.18: no
// key word is waited for from beyond L2 (profiles/r04/hot_keys_l2_ahead.txt).  And the products as a software pipeline over chunks
// (register pair, column) -- the key words of chunk c + 1 asked for before chunk c is multiplied, its products landing behind the next
// chunk's, the first chunk ahead of the clearing barrier: what took k_blind_rotate_glwe from 8.9 to 7.3 ms, where the compiler had left
// one load in flight -- 7.23-7.28 against 7.24-7.25 ms per 1 024 here: a pair's nine loads at the top of its iteration are flight enough.
// And the 48 ds_add_f64 of a step as plain stores (wrong results, a timing experiment: what any hand-over without atomics could
// save at most): 7.27 against 7.18 ms -- the landings are not what the step waits for either.)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
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
        const KeyRows keys(a.bsk_hat + ((size_t)FBS_KEY_STEP(i) * 3 * K1 + comp) * K1 * N);
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
    if (uint64_t *raw = gate_acc(a.gv, f, K1 * N)) {   // a rotation of TV_0 that several tables share: the whole accumulator
#pragma unroll
        for (int m = 0; m < E; m++) raw[comp * N + t + (uint32_t)LANES * m] = fp_to_u64(fp_canon(acc[m]));
        return;
    }
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

// -DFBS_CU_TRACE (experiments only, tools/trace_k2.sh): cycles per phase of a step, per wave of workgroup 0, summed over the rotation
#ifdef FBS_CU_TRACE
__device__ unsigned long long g_k2_trace[12 * 16];
#define K2_TRACE_INIT unsigned long long tr_t = __builtin_readcyclecounter(), tr_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define K2_TRACE(k)                                                  \
    {                                                                \
        const unsigned long long now = __builtin_readcyclecounter(); \
        tr_acc[k] += now - tr_t;                                     \
        tr_t = now;                                                  \
    }
#define K2_TRACE_FLUSH                                               \
    if (blockIdx.x == 0 && (threadIdx.x & 63u) == 0)                 \
        for (int k = 0; k < 10; k++) g_k2_trace[(threadIdx.x >> 6) * 16 + k] = tr_acc[k];
#else
#define K2_TRACE_INIT
#define K2_TRACE(k)
#define K2_TRACE_FLUSH
#endif
// ---------------------------------------------------------------------------------------------
// The LATENCY shape: ONE k = 2 bootstrap on the twelve waves of a workgroup -- what a launch that leaves most of the chip empty
// wants (a rank's slice of a level of a sharded circuit, a narrow level, the leftovers of a round): a bootstrap is n / 2 dependent
// steps, and in the three-waves-per-bootstrap kernel above each step is one wave's 2 965 instructions whatever the chip has idle
// (3.3-3.5 ms per launch at n = 734).  Here each of the three GLWE components is dealt over FOUR waves as in the k = 1 whole-CU
// kernels (fbs_blind_rotate_cu.hip; WavesNtt<10, 2>): two cross stages in registers, ONE trip through LDS to re-deal, then
// wave-private 256-point transforms at 4 coefficients per lane (LaneNtt256: register <-> lane transpositions, one LDS exchange
// each).  Per step, component c = waves 4 c .. 4 c + 3:
//   ACC_c itself -> balanced digit -> cross stages -> re-deal (barrier 1) -> private forward transform
//   -> bundle of row c (three samples x three columns: nine key polynomials' words) times the digits: products for all three
//      output components; the own one stays in registers, the other two are HANDED to the waves that hold the same evaluation
//      points of those components -- plain stores into the receiver's two slots (barrier 2); no atomics: a wave has exactly
//      two senders
//   -> sum, private inverse transform, re-deal back (barrier 3), joining stages, accumulate.
// Three workgroup barriers per step, as the k = 1 shape; 792 vector instructions per wave and step, 168 registers (eight dwords of
// loop-invariant twiddles are kept in scratch and re-read once per step).
// THE KEY ROWS OF A STEP DO NOT DEPEND ON ITS DATA.  One workgroup per CU pulls a step's whole row out of L2 by itself (27
// polynomials, 221 KB; a CU streams ~100 GB/s from L2: 2.2 us of a 6 us step), so when the words are asked for decides how much of
// that is waited for.  Measured at n = 734, per launch of 64 / 256 bootstraps (tools/k2_latency.py, one box):
//     first register pair at the top of the step, second after the first's products      2.30 / 2.51 ms
//     both at the top                                                                      2.49 / 2.66
//     first at the top, second after the re-deal barrier                                   2.43 / 2.62
//     first at the top, second ahead of the forward transform's last four stages           2.20 / 2.39
//     FIRST DURING THE STEP BEFORE (after its hand-over barrier: the words stream in behind the inverse transform),
//       second ahead of the forward transform's last four stages                           1.96 / 2.21   <- this kernel
//     ... second at the top of the step / one step ahead as well (after the re-deal back)  2.26 / 2.48,  2.17 / 2.38
// (more loads in flight than one register pair's is worse every time; so is asking for the three psi^(e o_lane) of the NEXT step
// behind the hand-over barrier as well: 2.06 / 2.36 -> 2.17 / 2.42 ms, same box, twice).  Two rounds of it serve 512 bootstraps in 4.2 ms (the
// three-waves-per-bootstrap kernel: 4.85).  The same shape on SIX waves (two per component, 512-point parts at 8 coefficients per
// lane, 223 registers, 89 KB of LDS: one workgroup per CU all the same) was built and measured: 2.91 ms at 64 and at 256 -- dropped.
// Evaluation points: array position P = 256 w + j of part w holds the value at psi^(2 bitrev10(P) + 1), and after the part's
// forward transform register m = (j1 j0) of lane ln holds j = (ln & 15) << 4 | (ln >> 4) << 2 | m: the register index is the two
// LOWEST bits of P, the two HIGHEST of its bit reversal, so 2 bitrev(P) + 1 = o_lane + 512 k_m with k_m = j1 + 2 j0, and
// psi^512 = R is a primitive FOURTH root of unity: zeta_m^e = psi^(e o_lane) R^(e k_m).  ONE table look-up per lane and exponent;
// R^e is wave-uniform (picked among 1, R, -1, -R by scalar instructions) and multiplied in once.
// WHERE A STEP'S TIME GOES (-DFBS_CU_TRACE, tools/trace_k2.sh, profiles/r04/k2_cu_phase_trace.txt; cycle stamps per wave of workgroup 0): of
// ~17 k cycles per step the forward transform takes 5.7 k, the products 4.5 k, the inverse 4.4 k for the wave that finishes last; the three
// waves of a SIMD (component 0, 1, 2: oldest first) do NOT advance together -- component 0 is issued first whenever it is ready and waits
// 2.1 M of a rotation's 6.2 M cycles at the barriers, component 2 never waits.  Handing the lead over in the middle of every stretch
// (s_setprio by component: the youngest leads the first half, the oldest the second -- what helps the k = 1 shape's TWO waves per SIMD)
// moves the waiting around and makes the rotation LONGER: 2.07 -> 2.39-2.54 ms per launch of 64.  The transforms are dependent chains
// (two butterflies per stage at 4 coefficients per lane) that no order of three such chains fills the pipe with; measured, not adopted.
// LDS (doubles): [3][N] re-deal + private forward exchange; [3][2][N] hand-over slots -- slot 0 doubles as the private inverse
// exchange and the re-deal back (wave w receives in the words it alone touches until the re-deal, as in k_blind_rotate_cu_pairs):
// 72 KB.
__global__ __launch_bounds__(768) void k_blind_rotate_cu_k2(BrArgs a) {
    constexpr int LOGN = 10, K1 = 3, PARTS = 4;
    using W = WavesNtt<LOGN, 2>;
    using Part = typename W::Half;
    constexpr int N = W::N, E = W::E, LANES = W::LANES, M = W::M;
    static_assert(std::is_same<Part, LaneNtt256>::value && E == 4 && W::EP == 1, "four waves per polynomial at 4 coefficients per lane");
    using Tw = CuTwiddles<Part, false, PARTS>;
    __shared__ double lds_all[K1 * N + K1 * 2 * N];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t comp = wave >> 2, w = wave & 3u;                          // GLWE component, part of it (both wave-uniform)
    const uint32_t t = threadIdx.x & (LANES - 1), ln = t & 63u;             // thread of the component, lane
    double *xf = lds_all + comp * N;                                        // re-deal region of this component
    double *hand = lds_all + K1 * N;                                        // [component][slot][N]
    double *back = hand + comp * 2 * N;                                     // slot 0 of this component: [part][M]
    double *mine = back + w * M;                                            // ... this wave's part of it
    const double *mine2 = back + N + w * M;                                 // ... and of slot 1
    // where this wave's products for the two other components go: the same part of their slots.  Component c receives from
    // c + 2 in slot 0 and from c + 1 in slot 1 (mod 3).  c1 = comp + 1, c2 = comp + 2 (mod 3)
    const uint32_t c1 = comp == 2u ? 0u : comp + 1u, c2 = comp == 0u ? 2u : comp - 1u;
    double *to_c1 = hand + (c1 * 2u + 0u) * N + w * M;
    double *to_c2 = hand + (c2 * 2u + 1u) * N + w * M;
    const uniform_doubles big_f = (uniform_doubles)(uintptr_t)a.tw_fwd, big_i = (uniform_doubles)(uintptr_t)a.tw_inv;
    Tw tw;
    tw.init(big_f, big_i, a.tw_fwd + W::LANE_TABLE_OFFSET, a.tw_inv + W::LANE_TABLE_OFFSET, w, ln, nullptr);

    const bool live = (size_t)blockIdx.x < a.count;
    const size_t f = live ? (size_t)blockIdx.x : a.count - 1;
    size_t gate, ms_row;
    gate_of(a.gv, f, &gate, &ms_row);
    uint32_t table = a.gv.table_ids ? a.gv.table_ids[gate] : 0;
    if (table >= a.n_tables) table = 0;
    const uint32_t *ms = a.ms + ms_row * (a.n + 1);
    const uint64_t *tv = a.tvs + (size_t)table * N;

    double acc[E];   // ACC = (0, 0, X^{-b~} * TV), centred; register m of thread t = coefficient t + 256 m
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
    // cross-stage twiddles: nodes 1, 2, 3 of the big tree (wave-uniform, scalar registers for the whole rotation)
    const double cw[3] = {big_f[1], big_f[2], big_f[3]}, iw[3] = {big_i[1], big_i[2], big_i[3]};
    const uint32_t o_lane = 2u * (__builtin_bitreverse32((uint32_t)M * w + (((ln & 15u) << 4) | ((ln >> 4) << 2))) >> (32 - LOGN)) + 1u;
    const double root = ((uniform_doubles)(uintptr_t)a.psi_pow)[N / 2];     // R = psi^512

    // row `comp` of the three samples of a step: [sample][row][column][N]; column c' = the products for component c'.  The
    // columns are taken in ROTATED order -- d = 0, 1, 2 stands for component comp + d (mod 3): own, next, next but one -- which
    // costs nothing (the column is the load's scalar offset) and makes "which product goes where" the same code in every wave
    const uint32_t t16 = t * 16u;   // this thread's 16 bytes of a register pair's 16 LANES
    const uint32_t col_bytes[K1] = {comp * (uint32_t)(N * 8), c1 * (uint32_t)(N * 8), c2 * (uint32_t)(N * 8)};
    auto request = [&](uint32_t step, auto jc, double2 (&k)[3][K1]) {
        constexpr int j = decltype(jc)::value;
        const KeyRows keys(a.bsk_hat + ((size_t)FBS_KEY_STEP(step) * 3 * K1 + comp) * K1 * N);
#pragma unroll
        for (int jj = 0; jj < 3; jj++)
#pragma unroll
            for (int d = 0; d < K1; d++)
                k[jj][d] = keys.load(t16 + (uint32_t)(j * LANES * 16), (uint32_t)(jj * K1 * K1) * (uint32_t)(N * 8) + col_bytes[d]);
    };
    using Pair0 = std::integral_constant<int, 0>;
    using Pair1 = std::integral_constant<int, 1>;
    auto lookup = [&](uint32_t ea, uint32_t eb, double (&out)[3]) {
        const uint32_t ee[3] = {ea, eb, (ea + eb) & (2u * N - 1u)};
#pragma unroll
        for (int jj = 0; jj < 3; jj++) {
            const uint32_t x = __umul24(ee[jj], o_lane) & (2u * N - 1u);
            const double v = a.psi_pow[x & (N - 1)];
            out[jj] = __hiloint2double(__double2hiint(v) ^ (int)((x << (31 - LOGN)) & 0x80000000u), __double2loint(v));
        }
    };
    const uint32_t n_pairs = a.n / 2;
    double2 kw0[3][K1], kw1[3][K1];
    request(0, Pair0{}, kw0);
    uint32_t e0_next = ms[0], e1_next = ms[1];
    K2_TRACE_INIT
    for (uint32_t i = 0; i < n_pairs; i++) {
        uint32_t e[3];
        e[0] = __builtin_amdgcn_readfirstlane(e0_next);
        e[1] = __builtin_amdgcn_readfirstlane(e1_next);
        e0_next = ms[2 * i + 2 < a.n ? 2 * i + 2 : a.n];   // (the last pair re-reads the body word and ignores it)
        e1_next = ms[2 * i + 3 < a.n ? 2 * i + 3 : a.n];
        const uint32_t i_next = i + 1 < n_pairs ? i + 1 : i;   // (the last step asks for its own row again: in bounds, unused)
        if (e[0] == 0 && e[1] == 0) {                       // the bundle is zero (uniform over the workgroup: one bootstrap)
            request(i_next, Pair0{}, kw0);
                continue;
        }
        e[2] = (e[0] + e[1]) & (2u * N - 1u);
        K2_TRACE(0)

        // ---- psi^(e o_lane) for the three exponents: one look-up each (psi^(x + N) = -psi^x) -------------------------------------
        double A[3];
        lookup(e[0], e[1], A);

        // ---- ACC_c itself, rounded to the closest multiple of q / B; the two cross stages; re-deal; private transform -----------
        double x[1][E];
#pragma unroll
        for (int m = 0; m < E; m++) {
            const uint32_t d = (uint32_t)__builtin_fma(acc[m], round_scale, round_offset) ^ bhalf;
            x[0][m] = (double)(int)__builtin_amdgcn_sbfe(d, 0, a.beta);
        }
        first_butterfly<0>(x[0][0], x[0][2], cw[0]);       // stage 0 pairs register m with m + 2 (node 1)
        first_butterfly<0>(x[0][1], x[0][3], cw[0]);
#pragma unroll
        for (int h = 0; h < 2; h++) {                       // stage 1 pairs m with m + 1 inside each half (nodes 2, 3)
            const double u = x[0][2 * h], v = fp_mulmod(x[0][2 * h + 1], cw[1 + h]);
            x[0][2 * h] = u + v;
            x[0][2 * h + 1] = u - v;
        }
#pragma unroll
        for (int q = 0; q < PARTS; q++) xf[q * M + t] = x[0][q];
        K2_TRACE(1)
        __syncthreads();
        K2_TRACE(2)
        double *bufs[1] = {xf + w * M};   // the words only this wave reads: its private exchange buffer from here on
#pragma unroll
        for (int m = 0; m < E; m++) x[0][m] = bufs[0][ln + 64u * m];
        // ---- the monomial factors: psi^(e o_lane) and its product with R^e (see above); the sign (-1)^e as a bit operation --------
        // (Written as selects among +-A, +-A R the compiler builds a table per exponent in SCRATCH memory and indexes it: six
        // stores and three dependent loads per step, each behind an s_waitcnt vmcnt(0) that also waits for the key words in flight.)
        double AR[3];
        auto root_powers = [&] {
#pragma unroll
            for (int jj = 0; jj < 3; jj++) {
                const uint32_t r = e[jj] & 3u;                                       // wave-uniform
                const double re = r == 0u ? 1.0 : r == 1u ? root : r == 2u ? -1.0 : -root;
                AR[jj] = fp_mulmod(A[jj], re);
            }
        };
        // zeta^e - 1 for register m = (j1 j0): k_m = j1 + 2 j0, zeta_m^e = A R^(e k_m): m = 0: A; 1: (-1)^e A; 2: A R^e; 3: (-1)^e A R^e
        auto mono = [&](int jj, int m) {
            const double v = (m & 2) ? AR[jj] : A[jj];
            if (!(m & 1)) return v - 1.0;
            return __hiloint2double(__double2hiint(v) ^ (int)(e[jj] << 31), __double2loint(v)) - 1.0;
        };
        // ---- the bundle words of a register pair (they do not depend on the digits), then their products with the digits' evaluations:
        // products for the three output components (rotated order)
        auto bundle = [&](auto jc, const double2 (&k)[3][K1], double (&w)[K1][2]) {
            constexpr int j = decltype(jc)::value;
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int m = 2 * j + r;
                const double mo[3] = {mono(0, m), mono(1, m), mono(2, m)};
#pragma unroll
                for (int d = 0; d < K1; d++) {
                    // lazy sum of three exact products (< 2.4 q); |x| < 2^49.3, so its product with a digit's evaluation stays exact
                    double wsum = fp_mulmod(r ? k[0][d].y : k[0][d].x, mo[0]);
#pragma unroll
                    for (int jj = 1; jj < 3; jj++) wsum += fp_mulmod(r ? k[jj][d].y : k[jj][d].x, mo[jj]);
                    w[d][r] = wsum;
                }
            }
        };
        double prod[K1][E], w0[K1][2], w1[K1][2];
        auto finish = [&](auto jc, const double (&w)[K1][2]) {
            constexpr int j = decltype(jc)::value;
#pragma unroll
            for (int r = 0; r < 2; r++)
#pragma unroll
                for (int d = 0; d < K1; d++) prod[d][2 * j + r] = fp_mulmod(x[0][2 * j + r], w[d][r]);
        };
        // (Measured and not adopted: the first pair's bundle -- 126 instructions that need nothing from the transform -- INSIDE the forward
        // transform, ahead of its last four stages, and both pairs' with the second pair's key words asked for a step ahead as well:
        // 2.06 / 2.35 -> 2.30 / 2.50 and 2.25 / 2.45 ms per launch of 64 / 256, same box, twice.  profiles/r04/k2_cu_bundle_in_transform_ab.txt)
        LaneNtt256::forward_multi<1, 0>(x, bufs, ln, tw.f, [&] { request(i, Pair1{}, kw1); });
        K2_TRACE(3)
        root_powers();
        bundle(Pair0{}, kw0, w0);
        finish(Pair0{}, w0);
        bundle(Pair1{}, kw1, w1);
        finish(Pair1{}, w1);

        // ---- hand the other two components theirs; sum; private inverse; re-deal back; joining stages; accumulate ---------------
#pragma unroll
        for (int m = 0; m < E; m++) {
            to_c1[64u * m + ln] = prod[1][m];
            to_c2[64u * m + ln] = prod[2][m];
        }
        K2_TRACE(4)
        __syncthreads();
        K2_TRACE(5)
        double own[E];
#pragma unroll
        for (int m = 0; m < E; m++) own[m] = prod[0][m] + mine[64u * m + ln] + mine2[64u * m + ln];
        request(i_next, Pair0{}, kw0);   // the NEXT step's first register pair: it streams in behind the inverse transform
        tw.inverse(own, mine, ln, LaneNtt256::NoHook{});   // three products below 0.8 q each: centred first by the transform
        Part::sync();
#pragma unroll
        for (int m = 0; m < E; m++) mine[ln + 64u * m] = own[m];
        K2_TRACE(6)
        __syncthreads();
        K2_TRACE(7)
#pragma unroll
        for (int q = 0; q < PARTS; q++) own[q] = back[q * M + t];
#pragma unroll
        for (int h = 0; h < 2; h++) {                       // the two joining stages (Gentleman-Sande: nodes 2, 3, then node 1)
            const double u = own[2 * h], v = own[2 * h + 1];
            own[2 * h] = u + v;
            own[2 * h + 1] = fp_mulmod(u - v, iw[1 + h]);
        }
#pragma unroll
        for (int m = 0; m < 2; m++) {
            const double u = own[m], v = own[m + 2];
            own[m] = u + v;
            own[m + 2] = fp_mulmod(u - v, iw[0]);
        }
#pragma unroll
        for (int m = 0; m < E; m++) acc[m] = fp_center(acc[m] + own[m]);
        K2_TRACE(8)
    }
    K2_TRACE_FLUSH

    // ---- sample extraction of coefficient 0 (two mask polynomials, the body), plus the table's constant -----------------
    if (!live) return;
    if (uint64_t *raw = gate_acc(a.gv, f, K1 * N)) {   // a rotation of TV_0 that several tables share: the whole accumulator
#pragma unroll
        for (int m = 0; m < E; m++) raw[comp * N + t + (uint32_t)LANES * m] = fp_to_u64(fp_canon(acc[m]));
        return;
    }
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
    // tune.br_k2_shape: 0 = by launch size (below); 3 = always the three-waves-per-bootstrap kernels; 12 = always the twelve-wave
    // whole-workgroup shape (A/B measurements, dispatch tests)
    const int64_t shape = ctx->tune.br_k2_shape;
    const bool small = a.count <= cus * K2_CU_ROUNDS && ctx->tune.br_cu_kernel && ctx->tune.br_cu_max_per_cu >= 1;
    if ((shape == 12 || (shape == 0 && small)) && ctx->d_bsk_hat_small) {
        // launches of up to three bootstraps per CU: one bootstrap on twelve waves, round after round (n = 734, one box: 2.06 ms at
        // 64, 2.36 at 256, 4.43 at 512, 6.37 at 768 bootstraps against 3.33 / 3.49 / 4.92 / 6.79 on three waves per bootstrap;
        // from 769 on a round of four-bootstrap workgroups is ahead: 6.8-7.2 ms up to 1 024)
        BrArgs b = a;
        b.bsk_hat = reinterpret_cast<const double *>(ctx->d_bsk_hat_small);
        *kernel = "k_blind_rotate_cu_k2";
        hipLaunchKernelGGL(k_blind_rotate_cu_k2, dim3((unsigned)a.count), dim3(768), 0, stream, b);
#ifdef FBS_CU_TRACE
        {
            unsigned long long h[12 * 16];
            if (hipDeviceSynchronize() == hipSuccess && hipMemcpyFromSymbol(h, HIP_SYMBOL(g_k2_trace), sizeof h) == hipSuccess) {
                fprintf(stderr, "trace k_blind_rotate_cu_k2 (cycles per phase, workgroup 0, whole rotation; wave = 4 component + part):\n");
                for (int w = 0; w < 12; w++) {
                    fprintf(stderr, "  wave %2d:", w);
                    for (int k = 0; k < 9; k++) fprintf(stderr, " %9llu", h[w * 16 + k]);
                    fprintf(stderr, "\n");
                }
            }
        }
#endif
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

void blind_rotate_k2_catalog(std::vector<std::string> *out) {
    for (const char *name : {"k_blind_rotate_pairs_k2<10,1>", "k_blind_rotate_pairs_k2<10,2>", "k_blind_rotate_pairs_k2<10,4>", "k_blind_rotate_cu_k2"})
        out->push_back(name);
}

}  // namespace fbs

// Blind rotation of ONE bootstrap on the eight waves of a CU: the shape of launches that leave most of the chip empty (a level
// of a circuit cut across GPUs, a narrow level, the leftovers of a round).  gfx950 only; same arithmetic, same rounding rules
// and the same ciphertexts as k_blind_rotate (fbs_blind_rotate.hip), word for word.
//
// A bootstrap's latency is n CMUX steps one after the other; inside a step the work is (k+1) l forward transforms, their
// products with the key row, and k+1 inverse transforms.  The two-waves-per-bootstrap kernel runs those one after the other in
// each wave (3 535 FP64-pipe instructions per wave and step at P1024: 4.8 ms per bootstrap however empty the chip is).  Here
// the N coefficients of a GLWE component are dealt over FOUR waves (N = 1024: 4 coefficients per lane, N = 2048: 8; component
// c = waves 4c .. 4c+3), as in WavesNtt<LOGN, 2>:
//   * the first two Cooley-Tukey stages pair registers of one thread and leave four independent N/4-point transforms hanging
//     from nodes 4 .. 7 of the twiddle tree; ONE trip through LDS re-deals them so that wave w owns part w;
//   * everything after that is private to a wave: LaneNtt256 / LaneNtt512 (fbs_ntt_lane.hpp) move index bits between registers
//     and lanes with v_permlane32/16_swap and go through LDS once per transform, and the l digit levels of a component go
//     through it TOGETHER (forward_multi): one re-deal and one exchange for all levels, 2 l (4 l) independent butterflies per
//     stage to cover the FP64 pipe's latency;
//   * products with the key row, hand-over of the partner component's half, private inverse transform, one re-deal back, the
//     two joining stages, accumulate.
// Four workgroup barriers per step (re-deal, hand-over, re-deal back, accumulator published for the next rotation) against
// twelve when the generic kernel is instantiated with the four-wave transform (4.15 ms) and twenty-eight with round 2's
// two-wave transform (4.8 ms).  The key copy is the one k_blind_rotate<LOGN, 8, ...> uses (d_bsk_hat_small: evaluation order of
// WavesNtt<LOGN, 2>).  Measured at P1024: 2.9 ms per bootstrap at up to 128, 3.15 ms at 256 bootstraps; the `lean` variant (two
// 128-register workgroups per CU) 5.4 ms for 512.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <type_traits>

#include "fbs_blind_rotate_cu.hpp"

namespace fbs {

// -DFBS_CU_TRACE (experiments only, tools/variant builds): cycles per phase of a step, per wave of workgroup 0, summed over the
// rotation and printed by the launcher -- where a step's time goes when the instruction count says it should be shorter
#ifdef FBS_CU_TRACE
__device__ unsigned long long g_cu_trace[8 * 16];
#define FBS_TRACE_INIT unsigned long long tr_t = __builtin_readcyclecounter(), tr_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define FBS_TRACE(k)                                              \
    {                                                             \
        const unsigned long long now = __builtin_readcyclecounter(); \
        tr_acc[k] += now - tr_t;                                  \
        tr_t = now;                                               \
    }
#define FBS_TRACE_FLUSH                                           \
    if (blockIdx.x == 0 && (threadIdx.x & 63u) == 0)              \
        for (int k = 0; k < 12; k++) g_cu_trace[(threadIdx.x >> 6) * 16 + k] = tr_acc[k];
#else
#define FBS_TRACE_INIT
#define FBS_TRACE(k)
#define FBS_TRACE_FLUSH
#endif
// Issue priority between the two waves of a SIMD (wave w of component 0 and wave w of component 1).  Left alone the SIMD issues
// the OLDER wave first whenever both are ready: component 0 runs every stretch between two barriers at full speed, waits at
// the barrier, and component 1 finishes it ALONE, with nobody to fill its stalls (traced with -DFBS_CU_TRACE at P1024: of a
// 7.0 M-cycle rotation component 0 spends 2.9 M waiting at barriers while component 1 issues at 54 % of the rate the pair
// reaches together).  favour() flips the lead in the middle of a stretch -- component 1 leads the first half, component 0
// the second -- so both arrive at the barrier together.  FBS_CU_PRIO: 0 off, 1 the long stretch only (forward transforms |
// products), 2 also the inverse transform (before | after its exchange).  Measured at P1024 (64 / 256 bootstraps): 3.02 / 3.10
// -> 2.95 / 3.05 -> 2.92 / 3.00 ms per step; flipping in the rotation-and-digits stretch as well: no further gain.  What remains
// is not imbalance: two waves per SIMD issue one FP64 instruction per 4.8-5.4 cycles at best (profiles/r01/fp64_issue_rate.txt).
#ifndef FBS_CU_PRIO
#define FBS_CU_PRIO 2
#endif
// (`me` must be WAVE-UNIFORM, a scalar register: on a value derived from threadIdx the compiler predicates both s_setprio
// with the exec mask, which scalar instructions ignore -- every wave then runs both and ends at priority 1.)
template <int LEVEL>
__device__ __forceinline__ void favour(bool me) {
    if constexpr (FBS_CU_PRIO >= LEVEL) {
        if (me) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    }
}
#ifndef FBS_CU_WAVES_PER_EU
#define FBS_CU_WAVES_PER_EU 2   // waves per SIMD the compiler must leave room for in the standard variant (experiments)
#endif
// NL: gadget levels (compile time: the levels' transforms are interleaved in registers); FIRST: what is known about the digits
// (first_butterfly, fbs_ntt.hpp): 2 = beta <= 7, 1 = beta <= 9, 0 = nothing.
// LEAN: the variant for launches of between one and two bootstraps per CU -- 128 registers per thread, so that two workgroups
// share a CU and fill each other's barrier and LDS stalls: the partner component's key words and the inverse twiddles are
// requested after the forward transforms instead of being held through them.
// FBS_CU_PREFETCH: the key words of step i + 1 requested during step i, behind its hand-over barrier (a step's key rows do not
// depend on its data): they stream in behind the inverse transform instead of the next step's rotation and digits.  Measured
// (tools/cu_latency.py, one box, per launch of 64 / 256 bootstraps at P1024): 2.86 / 3.12 -> 2.80 / 3.05 ms.
#ifndef FBS_CU_PREFETCH
#define FBS_CU_PREFETCH 1
#endif
template <int LOGN, int NL, int FIRST, bool LEAN = false>
__global__ __launch_bounds__(512, LEAN ? 4 : FBS_CU_WAVES_PER_EU) void k_blind_rotate_cu(BrArgs a) {
    using W = WavesNtt<LOGN, 2>;
    using Part = typename W::Half;
    constexpr int N = W::N, E = W::E, LANES = W::LANES, M = W::M, EP = W::EP, LOGE = W::LOGE;
    static_assert(LANES == 256 && (E == 4 || E == 8) && EP * 4 == E, "four waves per polynomial, 4 or 8 coefficients per lane");
    // LDS (doubles): [2][N] accumulator as the next rotation reads it (the hand-over borrows it between two rotations);
    // [2][NL][N] re-deal + private exchange buffers of the forward transforms (level 0's doubles for the inverse); the
    // inverse per-lane twiddles of 512-point parts.  64 KB at N = 1024, NL = 3; 130 KB at N = 2048, NL = 2.
    __shared__ double lds_all[2 * N + 2 * NL * N + CuTwiddles<Part, LEAN>::LDS_WORDS];
    // GLWE component owned by this thread: 0 = mask, 1 = body (wave-uniform, and known to the compiler as such)
    const uint32_t comp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
    const uint32_t t = threadIdx.x & (LANES - 1);    // thread of the component
    const uint32_t w = W::wave_of(t), ln = t & 63u;  // which part this wave owns; lane
    double *accbuf = lds_all + comp * N;
    double *accbuf_partner = lds_all + (comp ^ 1u) * N;
    double *xbuf = lds_all + 2 * N + comp * (NL * N);
    // twiddles: the per-lane ones from the part's own table, the wave-uniform ones from the big tree at the part's root 4 + w
    const uniform_doubles big_f = (uniform_doubles)(uintptr_t)a.tw_fwd, big_i = (uniform_doubles)(uintptr_t)a.tw_inv;
    CuTwiddles<Part, LEAN> tw;
    tw.init(big_f, big_i, a.tw_fwd + W::LANE_TABLE_OFFSET, a.tw_inv + W::LANE_TABLE_OFFSET, w, ln, lds_all + 2 * N + 2 * NL * N);

    const bool live = (size_t)blockIdx.x < a.count;
    const size_t f = live ? (size_t)blockIdx.x : a.count - 1;
    size_t gate, ms_row;
    gate_of(a.gv, f, &gate, &ms_row);
    uint32_t table = a.gv.table_ids ? a.gv.table_ids[gate] : 0;
    if (table >= a.n_tables) table = 0;
    const uint32_t *ms = a.ms + ms_row * (a.n + 1);
    const uint64_t *tv = a.tvs + (size_t)table * N;
    const uint32_t rows = 2 * NL;

    // ACC = (0, X^{-b~} * TV), centred; register m of thread t = coefficient t + 256 m
    double acc[E];
    {
        const uint32_t r = (2u * N - ms[a.n]) & (2u * N - 1u);
#pragma unroll
        for (int m = 0; m < E; m++) {
            const uint32_t idx = (t + (uint32_t)LANES * m - r) & (2u * N - 1u);
            const uint64_t v = tv[idx & (N - 1)];
            acc[m] = comp ? fp_center(fp_from_u64((idx & N) ? fq_neg(v) : v)) : 0.0;
        }
    }
    // rounding / digit constants: as in k_blind_rotate
    const double round_scale = fp_exp2i(-(int)(FQ_BITS - NL * a.beta));
    const uint32_t bhalf = 1u << (a.beta - 1);
    double round_offset = 0.5 + fp_exp2i((int)(NL * a.beta));
    uint32_t sign_bits = 0;
#pragma unroll
    for (uint32_t j = 0; j < NL; j++) {
        round_offset += (double)(bhalf << (j * a.beta));
        sign_bits |= bhalf << (j * a.beta);
    }
    // cross-stage twiddles: nodes 1, 2, 3 of the big tree (wave-uniform, scalar registers for the whole rotation)
    const double cw[3] = {big_f[1], big_f[2], big_f[3]}, iw[3] = {big_i[1], big_i[2], big_i[3]};

#pragma unroll
    for (int m = 0; m < E; m++) accbuf[t + (uint32_t)LANES * m] = acc[m];
    __syncthreads();

    const uint32_t t16 = t * 16u;   // this thread's 16 bytes of a register pair's 16 LANES
    uint32_t r_next = ms[0];
    const bool second = comp != 0u;   // this wave belongs to component 1 (scalar)
    constexpr bool PREFETCH = FBS_CU_PREFETCH != 0 && !LEAN;
    double2 ko[NL][E / 2], kt[NL][E / 2];
    auto request_keys = [&](uint32_t step) {
        const KeyRows rows_of(a.bsk_hat + ((size_t)FBS_KEY_STEP(step) * rows + comp * NL) * 2 * N);
#pragma unroll
        for (int lv = 0; lv < NL; lv++) {
            const uint32_t k_own = ((uint32_t)lv * 2u + comp) * (uint32_t)(N * 8), k_oth = ((uint32_t)lv * 2u + (comp ^ 1u)) * (uint32_t)(N * 8);
#pragma unroll
            for (int j = 0; j < E / 2; j++) {
                ko[lv][j] = rows_of.load(t16 + (uint32_t)(j * LANES * 16), k_own);
                if constexpr (!LEAN) kt[lv][j] = rows_of.load(t16 + (uint32_t)(j * LANES * 16), k_oth);
            }
        }
    };
    if constexpr (PREFETCH) request_keys(0);
    FBS_TRACE_INIT
    for (uint32_t i = 0; i < a.n; i++) {
        const uint32_t r = __builtin_amdgcn_readfirstlane(r_next);
        r_next = ms[i + 1];     // ms has n+1 entries; the last one (the body) is read here and ignored
        const uint32_t i_next = i + 1 < a.n ? i + 1 : i;
        if (r == 0) {           // X^0 * ACC - ACC = 0 (uniform over the workgroup: no barrier is skipped by part of it)
            if constexpr (PREFETCH) request_keys(i_next);
            continue;
        }
        FBS_TRACE(0)

        // key words of this step: this thread's four evaluations of the 2 NL polynomials of its component's rows.  Requested
        // now, used after the forward transforms: a step's 96 KB come out of L2 while the transforms run.
        // (FBS_CU_PREFETCH: requested during the step BEFORE instead, behind its hand-over barrier)
        const KeyRows keys(a.bsk_hat + ((size_t)FBS_KEY_STEP(i) * rows + comp * NL) * 2 * N);   // (buffer loads: fbs_blind_rotate.hpp)
        if constexpr (!PREFETCH) request_keys(i);

        // ---- (X^r - 1) * ACC_c, centred, rounded to the closest multiple of q / B^l; packed balanced digits -------------
        uint32_t digits[E];
        {
            const uint32_t from = (t - r) & (2u * N - 1u);
#pragma unroll
            for (int m = 0; m < E; m++) {
                const uint32_t idx = from + (uint32_t)LANES * m;   // < 3N: bit LOGN = sign, bits below = position
                const double wv = accbuf[idx & (N - 1)];
                const double v = __hiloint2double(__double2hiint(wv) ^ (int)((idx << (31 - LOGN)) & 0x80000000u), __double2loint(wv));
                const double d = v - acc[m];
                digits[m] = (uint32_t)__builtin_fma(d, round_scale, round_offset) ^ sign_bits;
            }
        }
        FBS_TRACE(1)

        // ---- all levels: digits -> the two cross stages -> re-deal -> private transforms, together ------------------------
        double x[NL][E];
#pragma unroll
        for (int lv = 0; lv < NL; lv++) {
            const uint32_t shift = ((uint32_t)NL - 1u - (uint32_t)lv) * a.beta;
#pragma unroll
            for (int m = 0; m < E; m++) x[lv][m] = (double)(int)__builtin_amdgcn_sbfe(digits[m], shift, a.beta);
            // stage 0 pairs register m with m + E/2 (node 1), stage 1 m with m + E/4 inside each half (nodes 2, 3)
#pragma unroll
            for (int m = 0; m < E / 2; m++) first_butterfly<FIRST>(x[lv][m], x[lv][m + E / 2], cw[0]);
#pragma unroll
            for (int m = 0; m < E; m++) {
                if (m & (E / 4)) continue;
                const double u = x[lv][m], v = fp_mulmod(x[lv][m + E / 4], cw[1 + (m >> (LOGE - 1))]);
                x[lv][m] = u + v;
                x[lv][m + E / 4] = u - v;
            }
            double *region = xbuf + lv * N;   // register q EP + r = element t + 256 r of part q
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int r = 0; r < EP; r++) region[q * M + t + (uint32_t)LANES * r] = x[lv][q * EP + r];
        }
        FBS_TRACE(2)
        __syncthreads();
        FBS_TRACE(3)
        if constexpr (!LEAN) favour<1>(second);
        double *bufs[NL];
#pragma unroll
        for (int lv = 0; lv < NL; lv++) {
            bufs[lv] = xbuf + lv * N + w * M;   // the words only this wave reads: its private exchange buffer from here on
#pragma unroll
            for (int m = 0; m < E; m++) x[lv][m] = bufs[lv][ln + 64u * m];
        }
        tw.template forward<NL>(x, bufs, ln);
        FBS_TRACE(4)
        if constexpr (!LEAN) favour<1>(!second);
        if constexpr (LEAN) {   // what the forward transforms had no registers for
#pragma unroll
            for (int lv = 0; lv < NL; lv++) {
                const uint32_t k_oth = ((uint32_t)lv * 2u + (comp ^ 1u)) * (uint32_t)(N * 8);
#pragma unroll
                for (int j = 0; j < E / 2; j++) kt[lv][j] = keys.load(t16 + (uint32_t)(j * LANES * 16), k_oth);
            }
            tw.prefetch_inverse();
        }

        // ---- products with the key row: contributions to this component and to the partner's (lazy sums) ----------------
        double own[E], other[E];
#pragma unroll
        for (int lv = 0; lv < NL; lv++)
#pragma unroll
            for (int j = 0; j < E / 2; j++) {
                const double p0 = fp_mulmod(x[lv][2 * j], ko[lv][j].x), p1 = fp_mulmod(x[lv][2 * j + 1], ko[lv][j].y);
                own[2 * j] = lv ? own[2 * j] + p0 : p0;
                own[2 * j + 1] = lv ? own[2 * j + 1] + p1 : p1;
            }
#pragma unroll
        for (int lv = 0; lv < NL; lv++)
#pragma unroll
            for (int j = 0; j < E / 2; j++) {
                const double q0 = fp_mulmod(x[lv][2 * j], kt[lv][j].x), q1 = fp_mulmod(x[lv][2 * j + 1], kt[lv][j].y);
                other[2 * j] = lv ? other[2 * j] + q0 : q0;
                other[2 * j + 1] = lv ? other[2 * j + 1] + q1 : q1;
            }

        {
            // ---- hand the partner its half (through the accumulator words: every rotation has read them by now) ---------
#pragma unroll
            for (int m = 0; m < E; m++) accbuf_partner[(uint32_t)LANES * m + t] = other[m];
            FBS_TRACE(5)
            __syncthreads();
            FBS_TRACE(6)
            if constexpr (!LEAN) favour<2>(second);
#pragma unroll
            for (int m = 0; m < E; m++) own[m] += accbuf[(uint32_t)LANES * m + t];
            if constexpr (PREFETCH) request_keys(i_next);   // the NEXT step's key words stream in behind the inverse transform
            // ---- private inverse, re-deal back -----------------------------------------------------------------------
            tw.inverse(own, bufs[0], ln, [&] {
                if constexpr (!LEAN) favour<2>(!second);
            });
            Part::sync();
#pragma unroll
            for (int m = 0; m < E; m++) bufs[0][ln + 64u * m] = own[m];
            FBS_TRACE(7)
            __syncthreads();
            FBS_TRACE(8)
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int r = 0; r < EP; r++) own[q * EP + r] = xbuf[q * M + t + (uint32_t)LANES * r];
        }
        // ---- the two joining stages (Gentleman-Sande: nodes 2, 3, then node 1), accumulate ---------------------------------
#pragma unroll
        for (int m = 0; m < E; m++) {
            if (m & (E / 4)) continue;
            const double u = own[m], v = own[m + E / 4];
            own[m] = u + v;
            own[m + E / 4] = fp_mulmod(u - v, iw[1 + (m >> (LOGE - 1))]);
        }
#pragma unroll
        for (int m = 0; m < E / 2; m++) {
            const double u = own[m], v = own[m + E / 2];
            own[m] = u + v;
            own[m + E / 2] = fp_mulmod(u - v, iw[0]);
        }
#pragma unroll
        for (int m = 0; m < E; m++) {
            acc[m] = fp_center(acc[m] + own[m]);
            accbuf[t + (uint32_t)LANES * m] = acc[m];
        }
        FBS_TRACE(9)
        __syncthreads();
        FBS_TRACE(10)
    }
    FBS_TRACE_FLUSH

    // ---- sample extraction of coefficient 0, plus the table's constant -----------------------------
    if (!live) return;
    if (uint64_t *raw = gate_acc(a.gv, f, 2 * N)) {   // a rotation of TV_0 that several tables share: the whole accumulator
#pragma unroll
        for (int m = 0; m < E; m++) raw[comp * N + t + (uint32_t)LANES * m] = fp_to_u64(fp_canon(acc[m]));
        return;
    }
    uint64_t *out = gate_out(a.gv, f, a.ct_words);
    if (comp == 0) {
#pragma unroll
        for (int m = 0; m < E; m++) {
            const uint32_t j = t + (uint32_t)LANES * m;
            const uint64_t v = fp_to_u64(fp_canon(acc[m]));
            if (j == 0) out[0] = v;
            else out[N - j] = fq_neg(v);
        }
    } else if (t == 0) {
        out[N] = fq_add(fp_to_u64(fp_canon(acc[0])), a.post[table]);
    }
}

// ---------------------------------------------------------------------------------------------
// The same shape for TWO KEY BITS PER STEP (bsk_group = 2; k_blind_rotate_pairs in fbs_blind_rotate.hip has the algebra): what
// the 128-bit parameter sets for p = 15 take (N = 2048, one gadget level, n/2 steps).  Per pair of key bits with rotation
// amounts (a0, a1) and e = (a0, a1, a0 + a1):
//     ACC += [ (X^e0 - 1) E0 + (X^e1 - 1) E1 + (X^e2 - 1) E2 ]  (x)  ACC,
// the bundle built in the transform domain, where X^e - 1 is the pointwise factor zeta^e - 1 with zeta the evaluation point a
// register holds.  ACC ITSELF is decomposed, so there is no rotated read and no barrier for one: three per step.
//
// Evaluation points in the LaneNtt512 layout.  Array position P = 512 w + j holds the value at psi^(2 bitrev11(P) + 1); after
// forward_multi register m = (r2 r1 r0) of lane ln of wave w holds j = (ln & 31) << 4 | (ln >> 5) << 3 | r1 << 2 | r0 << 1 | r2.
// The three register bits are the three LOWEST bits of P, i.e. the three HIGHEST of its bit reversal:
//     2 bitrev(P) + 1 = o_lane + 512 k_m,   k_m = r1 + 2 r0 + 4 r2,   o_lane < 512 (lane and wave bits only),
// and psi^512 = omega is a primitive EIGHTH root of unity.  So zeta^e = psi^(e o_lane) omega^(e k_m mod 8): ONE table look-up
// per lane and exponent (psi^x, x < 2N, from the table in global memory: three gathers per step where the two-wave kernel does
// 48 from LDS, with their bank conflicts), three products by the constants omega, omega^2, omega^3, and per register a
// wave-uniform choice among +-(A, A omega, A omega^2, A omega^3).
// NL gadget levels (1: the p = 15 sets; 2: what p = 31 takes with two key bits per step): the levels' forward transforms run
// together (forward_multi), and the key words -- 6 NL polynomials per component and step -- are fetched register pair by
// register pair while the bundle of the pair before is built, instead of being held through the transforms.
// FBS_CU_PAIRS_PREFETCH: the first register pair's key words of step i + 1 requested during step i, behind its hand-over barrier
// (k_blind_rotate_cu_k2, fbs_blind_rotate_k2.hip, has the measurements of this order at k = 2, where it is worth 15 %).  Here, per
// launch of 64 / 256 bootstraps (tools/cu_latency.py, one box): one level (128-bit p = 15 set, n = 714) 2.63 / 2.90 -> 2.56 / 2.82 ms;
// two levels (p = 31, n = 766) 4.10 / 4.40 -> 4.15 / 4.46 ms and 16.53 -> 16.73 per 1 024 -- so it is on for one level only.
// FBS_CU_PAIRS_L2_AHEAD: steps ahead at which the workgroups of an XCD touch the key row (below); 0 = nobody does
#ifndef FBS_CU_PAIRS_L2_AHEAD
#define FBS_CU_PAIRS_L2_AHEAD 2
#endif
#ifndef FBS_CU_PAIRS_PREFETCH
#define FBS_CU_PAIRS_PREFETCH 1
#endif
template <int LOGN, int NL>
__global__ __launch_bounds__(512, 2) void k_blind_rotate_cu_pairs(BrArgs a) {
    using W = WavesNtt<LOGN, 2>;
    using Part = typename W::Half;
    constexpr int N = W::N, E = W::E, LANES = W::LANES, M = W::M, EP = W::EP, LOGE = W::LOGE;
    static_assert(LOGN == 11 && E == 8 && std::is_same<Part, LaneNtt512>::value, "written for 512-point parts at 8 coefficients per lane");
    static_assert(NL == 1 || NL == 2, "one or two gadget levels");
    // LDS (doubles): [2][N] hand-over, then the inverse transform's private exchange and the re-deal back -- wave w of a component
    // receives its hand-over words in [w M, (w + 1) M), the words it alone touches until the re-deal, so the three uses need no
    // barrier beyond the two they have (the next step's hand-over lies behind that step's re-deal barrier); [2][NL][N] re-deal +
    // private exchange of the forward transforms; the inverse per-lane twiddles.  100 KB (NL = 1), 133 KB (NL = 2).
    __shared__ double lds_all[2 * N + 2 * NL * N + CuTwiddles<Part, false>::LDS_WORDS];
    const uint32_t comp = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8), t = threadIdx.x & (LANES - 1);
    const uint32_t w = W::wave_of(t), ln = t & 63u;
    double *back = lds_all + comp * N;
    double *hand_mine = back + w * M, *hand_partner = lds_all + (comp ^ 1u) * N + w * M;
    double *xf = lds_all + 2 * N + comp * (NL * N);
    const uniform_doubles big_f = (uniform_doubles)(uintptr_t)a.tw_fwd, big_i = (uniform_doubles)(uintptr_t)a.tw_inv;
    CuTwiddles<Part, false> tw;
    tw.init(big_f, big_i, a.tw_fwd + W::LANE_TABLE_OFFSET, a.tw_inv + W::LANE_TABLE_OFFSET, w, ln, lds_all + 2 * N + 2 * NL * N);

    const bool live = (size_t)blockIdx.x < a.count;
    const size_t f = live ? (size_t)blockIdx.x : a.count - 1;
    size_t gate, ms_row;
    gate_of(a.gv, f, &gate, &ms_row);
    uint32_t table = a.gv.table_ids ? a.gv.table_ids[gate] : 0;
    if (table >= a.n_tables) table = 0;
    const uint32_t *ms = a.ms + ms_row * (a.n + 1);
    const uint64_t *tv = a.tvs + (size_t)table * N;
    constexpr uint32_t rows = 2 * NL;

    double acc[E];   // ACC = (0, X^{-b~} * TV), centred; register m of thread t = coefficient t + 256 m
    {
        const uint32_t r = (2u * N - ms[a.n]) & (2u * N - 1u);
#pragma unroll
        for (int m = 0; m < E; m++) {
            const uint32_t idx = (t + (uint32_t)LANES * m - r) & (2u * N - 1u);
            const uint64_t v = tv[idx & (N - 1)];
            acc[m] = comp ? fp_center(fp_from_u64((idx & N) ? fq_neg(v) : v)) : 0.0;
        }
    }
    // rounding / digit constants: as in k_blind_rotate (abar = round(acc / 2^(46 - NL beta)) mod B^NL, balanced digits packed in
    // two's complement fields)
    const double round_scale = fp_exp2i(-(int)(FQ_BITS - NL * a.beta));
    const uint32_t bhalf = 1u << (a.beta - 1);
    double round_offset = 0.5 + fp_exp2i((int)(NL * a.beta));
    uint32_t sign_bits = 0;
#pragma unroll
    for (uint32_t j = 0; j < NL; j++) {
        round_offset += (double)(bhalf << (j * a.beta));
        sign_bits |= bhalf << (j * a.beta);
    }
    const double cw[3] = {big_f[1], big_f[2], big_f[3]}, iw[3] = {big_i[1], big_i[2], big_i[3]};
    // o_lane = 2 bitrev11(512 w + (ln & 31) << 4 | (ln >> 5) << 3) + 1; omega^s = psi^(512 s), s = 1, 2, 3 (wave-uniform)
    const uint32_t o_lane = 2u * (__builtin_bitreverse32(512u * w + ((ln & 31u) << 4) + ((ln >> 5) << 3)) >> (32 - LOGN)) + 1u;
    const uniform_doubles psi_u = (uniform_doubles)(uintptr_t)a.psi_pow;
    const double om1 = psi_u[512], om2 = psi_u[1024], om3 = psi_u[1536];
    __syncthreads();   // (the inverse twiddle table is in place)

    const uint32_t t16 = t * 16u;   // this thread's 16 bytes of a register pair's 16 LANES
    const uint32_t n_pairs = a.n / 2;
    // (the key words of one register pair (2j, 2j + 1) of a step: three samples x NL rows, of the own or of the partner's column)
    auto request_at = [&](uint32_t step, auto jc, auto partner, double2 (&k)[3][NL]) {
        constexpr int j = decltype(jc)::value;
        const KeyRows keys(a.bsk_hat + (size_t)FBS_KEY_STEP(step) * (3u * rows * 2u) * N + (size_t)comp * (NL * 2u) * N);   // (buffer loads: fbs_blind_rotate.hpp)
        const uint32_t col = decltype(partner)::value ? comp ^ 1u : comp;
#pragma unroll
        for (int jj = 0; jj < 3; jj++)
#pragma unroll
            for (int lv = 0; lv < NL; lv++)
                k[jj][lv] = keys.load(t16 + (uint32_t)(j * LANES * 16), ((uint32_t)(jj * (int)rows + lv) * 2u + col) * (uint32_t)(N * 8));
    };
    using Own = std::false_type;
    using Oth = std::true_type;
    using Pair0 = std::integral_constant<int, 0>;
    constexpr bool PAIRS_PREFETCH = FBS_CU_PAIRS_PREFETCH != 0 && NL == 1;
    uint32_t e0_next = ms[0], e1_next = ms[1];
    double2 ko[3][NL], kt[3][NL];
    if constexpr (PAIRS_PREFETCH) {
        request_at(0, Pair0{}, Own{}, ko);
        request_at(0, Pair0{}, Oth{}, kt);
    }
#if FBS_CU_PAIRS_L2_AHEAD
    // A step's key row comes out of L2 -- if somebody has brought it there: the key (150 MB at the p = 31 set) lives in the Infinity
    // Cache / HBM, and the first CU of an XCD to ask for a line waits for the fabric, with every wave of the CU behind the same words.
    // With every step reading the rows of steps 0 and 1 (-DFBS_EXP_HOT_KEYS, wrong results, a timing experiment) a launch of 1 024 at
    // that set takes 15.92 ms against 16.58, of 64 3.94 against 4.10; the throughput shapes (three or two waves per SIMD and four
    // bootstraps per CU asking together) lose nothing there (profiles/r04/hot_keys_l2_ahead.txt).  So the workgroups of an XCD
    // (blockIdx mod 8), which walk the key in step, each touch ONE line in every `peers` of the row FBS_CU_PAIRS_L2_AHEAD steps
    // ahead, behind the hand-over barrier when the step's own key words are all in: one load per thread of the first waves, its
    // value never looked at.  Measured, same box, twice: p = 31 16.65-16.70 -> 16.36-16.45 ms per 1 024, 4.08-4.10 -> 3.93-3.95 per
    // 64; p = 15 at k = 1 2.48-2.49 -> 2.41-2.42 per 64 (one step ahead: the same within 0.3 %).
    // (The load is written without a branch -- a thread without a line asks beyond the end of the resource, which touches no memory:
    // with `if (mine) load` the compiler's schedule of the whole step changed, 196 -> 256 registers and 250 bytes spilled.)
    // Measured with it and NOT adopted (NL = 2): pairs 0 AND 1 of the key words asked for at the top of the step, two sets of
    // registers (246, nothing spilled), so that half the row streams in behind the forward transform: 16.88-16.90 against
    // 16.65-16.70 ms per 1 024, 4.16-4.17 against 4.08-4.10 per 64.  The phase trace agrees that the products are no longer what
    // waits: 2.4 M cycles per rotation for 2.0 M cycles of instructions, alone on the SIMD or not (profiles/r04/cu_pairs_p31_phase_trace.txt).
    constexpr uint32_t ROW_LINES = 3u * rows * 2u * N * 8u / 128u;
    const uint32_t peers = (uint32_t)std::min<size_t>(32, (a.count + 7) / 8);
    const uint32_t slice_lines = std::min<uint32_t>(512u, (ROW_LINES + peers - 1) / peers);
    const uint32_t ahead_line = ((blockIdx.x >> 3) % peers) * slice_lines + threadIdx.x;
    const uint32_t ahead_off = threadIdx.x < slice_lines && ahead_line < ROW_LINES ? ahead_line * 128u : 0x7FFFFFF0u;
    uint32_t ahead_word = 0;
#endif
    FBS_TRACE_INIT
    for (uint32_t i = 0; i < n_pairs; i++) {
        uint32_t e[3];
        e[0] = __builtin_amdgcn_readfirstlane(e0_next);
        e[1] = __builtin_amdgcn_readfirstlane(e1_next);
        e0_next = ms[2 * i + 2 < a.n ? 2 * i + 2 : a.n];   // (the last pair re-reads the body word and ignores it)
        e1_next = ms[2 * i + 3 < a.n ? 2 * i + 3 : a.n];
        const uint32_t i_next = i + 1 < n_pairs ? i + 1 : i;   // (the last step asks for its own row again: in bounds, unused)
        if (e[0] == 0 && e[1] == 0) {                       // the bundle is zero (uniform over the workgroup)
            if constexpr (PAIRS_PREFETCH) {     // (the words asked for ahead were this step's)
                request_at(i_next, Pair0{}, Own{}, ko);
                request_at(i_next, Pair0{}, Oth{}, kt);
            }
            continue;
        }
        e[2] = (e[0] + e[1]) & (2u * N - 1u);
        FBS_TRACE(0)

        // ---- what memory has to bring: psi^(e o_lane) for the three exponents, and the key words -----------------------------
        double A[3];
#pragma unroll
        for (int jj = 0; jj < 3; jj++) {
            const uint32_t x = __umul24(e[jj], o_lane) & (2u * N - 1u);
            const double v = a.psi_pow[x & (N - 1)];
            A[jj] = __hiloint2double(__double2hiint(v) ^ (int)((x << (31 - LOGN)) & 0x80000000u), __double2loint(v));   // psi^(x + N) = -psi^x
        }
        // row (jj, comp NL + lv) of step i: its own column (this component's products) and the partner's
        auto request = [&](auto jc, auto partner, double2 (&k)[3][NL]) { request_at(i, jc, partner, k); };
        // Pair 0's words are requested here, ahead of the transforms; pair j + 1's when pair j has been used up.  Measured against
        // holding more through the transforms (NL = 1: all own words, 3.09 ms per 256 bootstraps against 2.89; NL = 2: two pairs,
        // 17.7 against 17.1 ms per 1024): what counts is that nothing is spilled.
        // (With the buffer loads' spare registers -- 196 at NL = 2 -- a second set of key words fits: pair j + 1 requested BEFORE pair j is
        // consumed.  Measured, same box, twice each: NL = 1 2.76-2.79 against 2.79-2.85 ms per 256 bootstraps, NL = 2 16.72-16.74
        // against 16.55 ms per 1024 -- within the noise one way, 1 % the other: not adopted.)
        if constexpr (!PAIRS_PREFETCH) {
            request(Pair0{}, Own{}, ko);
            request(Pair0{}, Oth{}, kt);
        }

        // ---- ACC_c itself, rounded to the closest multiple of q / B^NL; the two cross stages; re-deal; private transforms ---------
        double x[NL][E];
        {
            uint32_t digits[E];
#pragma unroll
            for (int m = 0; m < E; m++) digits[m] = (uint32_t)__builtin_fma(acc[m], round_scale, round_offset) ^ sign_bits;
#pragma unroll
            for (int lv = 0; lv < NL; lv++) {
                const uint32_t shift = ((uint32_t)NL - 1u - (uint32_t)lv) * a.beta;
#pragma unroll
                for (int m = 0; m < E; m++) x[lv][m] = (double)(int)__builtin_amdgcn_sbfe(digits[m], shift, a.beta);
#pragma unroll
                for (int m = 0; m < E / 2; m++) first_butterfly<0>(x[lv][m], x[lv][m + E / 2], cw[0]);
#pragma unroll
                for (int m = 0; m < E; m++) {
                    if (m & (E / 4)) continue;
                    const double u = x[lv][m], v = fp_mulmod(x[lv][m + E / 4], cw[1 + (m >> (LOGE - 1))]);
                    x[lv][m] = u + v;
                    x[lv][m + E / 4] = u - v;
                }
                double *region = xf + lv * N;
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int r = 0; r < EP; r++) region[q * M + t + (uint32_t)LANES * r] = x[lv][q * EP + r];
            }
        }
        FBS_TRACE(1)
        __syncthreads();
        FBS_TRACE(2)
        double *bufs[NL];
#pragma unroll
        for (int lv = 0; lv < NL; lv++) {
            bufs[lv] = xf + lv * N + w * M;
#pragma unroll
            for (int m = 0; m < E; m++) x[lv][m] = bufs[lv][ln + 64u * m];
        }
        tw.template forward<NL>(x, bufs, ln);
        FBS_TRACE(3)

        // ---- the monomial factors ----------------------------------------------------------------------------------------
        // zeta_m^e = psi^(e o_lane) omega^(e k_m), k_m = r1 + 2 r0 + 4 r2 for register m = (r2 r1 r0); omega^(t + 4) = -omega^t.
        // NL = 1: omega^e is WAVE-UNIFORM -- it and its second and third power are picked by SCALAR instructions among 1, omega,
        // omega^2, omega^3 and their negatives and multiplied into psi^(e o_lane) once each (three exact products per exponent);
        // what is left per register is a compile-time choice among the four and, for k_m >= 4, the sign omega^(4 e) = (-1)^e as
        // one bit operation.  (Round 4; before, the products were by omega, omega^2, omega^3 and the choice per register was 35
        // v_cndmask behind scalar bit tests: 1 548 -> 1 510 instructions per wave and step, 235 -> 159 registers; with the key words
        // asked for a step ahead, 2.62 / 2.84 -> 2.48 / 2.67 ms per launch of 64 / 256 bootstraps on one box.)
        // NL = 2 keeps the form it had: the wave-uniform factor omega^(e k_m) is picked by scalar instructions PER REGISTER and
        // multiplied in (24 exact products per step where the form above has 9).  The form above was measured there too, same box:
        // 2 424 -> 2 332 instructions, 196 -> 212 registers, and 16.51 -> 16.81 ms per 1 024 bootstraps, 4.09 -> 4.17 per 64 -- that
        // kernel waits for its key rows (below), not for the issue port.
        // (The two forms are written out separately on purpose: how the selections are spelled decides whether the compiler keeps
        // them in registers -- a nested-conditional spelling of the NL = 1 form came back with a table in LDS and 32 bytes of scratch.)
        struct Powers {
            double s[4];
        };
        auto powers = [&](double base, uint32_t ej) {
            Powers V;
            V.s[0] = base;
            if constexpr (NL == 1) {
#pragma unroll
                for (int q = 1; q < 4; q++) {
                    const uint32_t idx = ej * (uint32_t)q, r = idx & 3u;                 // wave-uniform
                    const double v = r == 3u ? om3 : r == 2u ? om2 : r == 1u ? om1 : 1.0;
                    V.s[q] = fp_mulmod(base, (idx & 4u) ? -v : v);
                }
            }
            return V;
        };
        const Powers V0 = powers(A[0], e[0]), V1 = powers(A[1], e[1]), V2 = powers(A[2], e[2]);
        auto mono = [&](const Powers &V, uint32_t ej, int m) {
            if constexpr (NL == 1) {
                const int km = ((m >> 1) & 1) | ((m & 1) << 1) | (m & 4);
                const double v = V.s[km & 3];
                if (!(km & 4)) return v - 1.0;
                return __hiloint2double(__double2hiint(v) ^ (int)(ej << 31), __double2loint(v)) - 1.0;   // times (-1)^e
            } else {
                const uint32_t km = (uint32_t)(((m >> 1) & 1) | ((m & 1) << 1) | (m & 4));
                const uint32_t tt = (ej * km) & 7u;                                      // wave-uniform
                const double om = (tt & 2u) ? ((tt & 1u) ? om3 : om2) : ((tt & 1u) ? om1 : 1.0);
                return fp_mulmod(V.s[0], (tt & 4u) ? -om : om) - 1.0;    // |.| < 0.75 q + 1: the products below stay exact
            }
        };
        // bundle words of a register (lazy sums of three exact products, < 2.4 q) times the digits' evaluations: |x| < 2^49.3,
        // within fp_mulmod's range (as in k_blind_rotate_pairs)
        double own[E], other[E];
        auto consume = [&](auto jc, const double2 (&k_own)[3][NL], const double2 (&k_oth)[3][NL]) {
            constexpr int j = decltype(jc)::value;
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int m = 2 * j + r;
                const double mo[3] = {mono(V0, e[0], m), mono(V1, e[1], m), mono(V2, e[2], m)};
#pragma unroll
                for (int lv = 0; lv < NL; lv++) {
                    // (the first product initialises the sums: 0.0 + x is an instruction the compiler may not drop)
                    double wo = fp_mulmod(r ? k_own[0][lv].y : k_own[0][lv].x, mo[0]), wt = fp_mulmod(r ? k_oth[0][lv].y : k_oth[0][lv].x, mo[0]);
#pragma unroll
                    for (int jj = 1; jj < 3; jj++) {
                        const double2 a_own = k_own[jj][lv], a_oth = k_oth[jj][lv];
                        wo += fp_mulmod(r ? a_own.y : a_own.x, mo[jj]);
                        wt += fp_mulmod(r ? a_oth.y : a_oth.x, mo[jj]);
                    }
                    const double p = fp_mulmod(x[lv][m], wo), q = fp_mulmod(x[lv][m], wt);
                    own[m] = lv ? own[m] + p : p;
                    other[m] = lv ? other[m] + q : q;
                }
            }
        };
#define FBS_CU_PAIR_STEP(J)                                                                                      \
    {                                                                                                            \
        consume(std::integral_constant<int, (J)>{}, ko, kt);                                                     \
        __builtin_amdgcn_sched_barrier(0);   /* (or every request is hoisted to the top, and spilled) */         \
        if constexpr ((J) + 1 < E / 2) {                                                                         \
            request(std::integral_constant<int, ((J) + 1) % (E / 2)>{}, Own{}, ko);                              \
            request(std::integral_constant<int, ((J) + 1) % (E / 2)>{}, Oth{}, kt);                              \
        }                                                                                                        \
    }
        FBS_CU_PAIR_STEP(0) FBS_CU_PAIR_STEP(1) FBS_CU_PAIR_STEP(2) FBS_CU_PAIR_STEP(3)
#undef FBS_CU_PAIR_STEP

        // ---- hand the partner its half, private inverse, re-deal back, the two joining stages, accumulate ------------------------
#pragma unroll
        for (int m = 0; m < E; m++) hand_partner[64u * m + ln] = other[m];
        FBS_TRACE(4)
        __syncthreads();
        FBS_TRACE(5)
#if FBS_CU_PAIRS_L2_AHEAD
        {   // (no branch: a thread without a line asks beyond the resource's end, which touches no memory)
            const uint32_t far_step = i + FBS_CU_PAIRS_L2_AHEAD < n_pairs ? i + FBS_CU_PAIRS_L2_AHEAD : n_pairs - 1;
            const __amdgpu_buffer_rsrc_t far = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<double *>(a.bsk_hat + (size_t)far_step * (3u * rows * 2u) * N), 0, ROW_LINES * 128u, 0x00020000);
            ahead_word = __builtin_amdgcn_raw_buffer_load_b32(far, ahead_off, 0, 0);
        }
#endif
#pragma unroll
        for (int m = 0; m < E; m++) own[m] += hand_mine[64u * m + ln];
        if constexpr (PAIRS_PREFETCH) {
            // the NEXT step's first register pair: the key rows of a step do not depend on its data, so they stream in behind the
            // inverse transform (k_blind_rotate_cu_k2, fbs_blind_rotate_k2.hip, has the measurements of this order)
            request_at(i_next, Pair0{}, Own{}, ko);
            request_at(i_next, Pair0{}, Oth{}, kt);
        }
        // (no favour() in this kernel.  Measured with it: NL = 1 2.92 -> 2.96-3.05 ms per 256 bootstraps, NL = 2 4.5 -> 4.5 / 7.1 ms.
        // The trace shows the products phase far from issue-bound -- the wave that leads it takes 3.0 M cycles for 1.6 M cycles of
        // instructions (NL = 2), the one that follows 5.4 M, whichever way the lead is given.  Its arithmetic alone runs at the
        // FP64 issue ceiling (tools/products_bench.hip: 5.1-5.3 cycles per instruction at two waves per SIMD); what it waits for
        // is the KEY ROW: every CU pulls the step's whole row out of L2 by itself (393 KB at NL = 2, 196 KB at NL = 1), a CU
        // streams 90-120 GB/s from L2 at best (tools/l2_stream_bench.hip), and the loads come in this phase's third of the step.
        // With half the distinct bytes (timing experiment) 4.62 -> 3.80 ms and 2.98 -> 2.63 ms per 256 bootstraps.  Requesting
        // ALL key words at the top of the step is slower (NL = 1: 3.17 against 2.95 ms: the queue fills and blocks the wave at
        // issue), non-temporal loads too, and starting the workgroups of an XCD a quarter of a step apart changes nothing
        // (profiles/r03/microbench_products_l2.txt).)
        tw.inverse(own, hand_mine, ln, LaneNtt512::NoHook{});
        Part::sync();
#pragma unroll
        for (int m = 0; m < E; m++) hand_mine[ln + 64u * m] = own[m];
        FBS_TRACE(6)
        __syncthreads();
        FBS_TRACE(7)
#pragma unroll
        for (int q = 0; q < 4; q++)
#pragma unroll
            for (int r = 0; r < EP; r++) own[q * EP + r] = back[q * M + t + (uint32_t)LANES * r];
#pragma unroll
        for (int m = 0; m < E; m++) {
            if (m & (E / 4)) continue;
            const double u = own[m], v = own[m + E / 4];
            own[m] = u + v;
            own[m + E / 4] = fp_mulmod(u - v, iw[1 + (m >> (LOGE - 1))]);
        }
#pragma unroll
        for (int m = 0; m < E / 2; m++) {
            const double u = own[m], v = own[m + E / 2];
            own[m] = u + v;
            own[m + E / 2] = fp_mulmod(u - v, iw[0]);
        }
#pragma unroll
        for (int m = 0; m < E; m++) acc[m] = fp_center(acc[m] + own[m]);
        FBS_TRACE(8)
    }
    FBS_TRACE_FLUSH
#if FBS_CU_PAIRS_L2_AHEAD
    asm volatile("" ::"v"(ahead_word));   // (never looked at; this keeps the loads)
#endif

    if (!live) return;
    if (uint64_t *raw = gate_acc(a.gv, f, 2 * N)) {
#pragma unroll
        for (int m = 0; m < E; m++) raw[comp * N + t + (uint32_t)LANES * m] = fp_to_u64(fp_canon(acc[m]));
        return;
    }
    uint64_t *out = gate_out(a.gv, f, a.ct_words);
    if (comp == 0) {
#pragma unroll
        for (int m = 0; m < E; m++) {
            const uint32_t j = t + (uint32_t)LANES * m;
            const uint64_t v = fp_to_u64(fp_canon(acc[m]));
            if (j == 0) out[0] = v;
            else out[N - j] = fq_neg(v);
        }
    } else if (t == 0) {
        out[N] = fq_add(fp_to_u64(fp_canon(acc[0])), a.post[table]);
    }
}

#ifdef FBS_CU_TRACE
void cu_trace_dump(const char *kernel);
#define FBS_TRACE_DUMP(k) cu_trace_dump(k)
#else
#define FBS_TRACE_DUMP(k)
#endif

bool launch_blind_rotate_cu_pairs(fbs_ctx *ctx, const BrArgs &a, hipStream_t stream, std::string *kernel) {
    const fbs_params &p = ctx->p;
    if (ctx->group != 2 || !ctx->d_bsk_hat_small || p.log_n_poly != 11 || p.l_bsk > 2 || !ctx->tune.br_cu_kernel) return false;
    BrArgs b = a;
    b.bsk_hat = reinterpret_cast<const double *>(ctx->d_bsk_hat_small);
    if (p.l_bsk == 1) {
        *kernel = "k_blind_rotate_cu_pairs<11,1>";
        hipLaunchKernelGGL((k_blind_rotate_cu_pairs<11, 1>), dim3((unsigned)a.count), dim3(512), 0, stream, b);
        FBS_TRACE_DUMP(kernel->c_str());
    } else {
        *kernel = "k_blind_rotate_cu_pairs<11,2>";
        hipLaunchKernelGGL((k_blind_rotate_cu_pairs<11, 2>), dim3((unsigned)a.count), dim3(512), 0, stream, b);
        FBS_TRACE_DUMP(kernel->c_str());
    }
    return true;
}

bool launch_blind_rotate_cu(fbs_ctx *ctx, const BrArgs &a, hipStream_t stream, std::string *kernel) {
    const fbs_params &p = ctx->p;
    // N = 1024 with up to four gadget levels, N = 2048 with up to two (LDS: 2 N + 2 l N words + the inverse twiddles)
    if (ctx->group != 1 || !ctx->d_bsk_hat_small) return false;
    if (!((p.log_n_poly == 10 && p.l_bsk <= 4) || (p.log_n_poly == 11 && p.l_bsk <= 2))) return false;
    if (!ctx->tune.br_cu_kernel) return false;   // (A/B switch: the generic kernel on the four-wave transform)
    const int first = p.beta_bsk <= 7 ? 2 : p.beta_bsk <= 9 ? 1 : 0;
    BrArgs b = a;
    b.bsk_hat = reinterpret_cast<const double *>(ctx->d_bsk_hat_small);
    const dim3 grid((unsigned)a.count), block(512);
    // more than one bootstrap per CU: the two-workgroups-per-CU variant where there is one (N = 1024, up to three levels)
    // (512 bootstraps at P1024: 6.07 ms as two rounds of the 162-register kernel, 5.40 ms with two workgroups per CU; 1 536 =
    // 1 024 + 512: 99.7 -> 105.3 k FBS/s)
    const bool lean = p.log_n_poly == 10 && p.l_bsk <= 3 &&
                      (ctx->tune.br_cu_lean == 2 || (ctx->tune.br_cu_lean == 1 && a.count > (size_t)ctx->cu_count));
#define CU_CASE(L, NL, FIRST)                                                                    \
    if (p.log_n_poly == L && p.l_bsk == NL && first == FIRST) {                                  \
        if constexpr (L == 10 && NL <= 3) {                                                      \
            if (lean) {                                                                          \
                *kernel = "k_blind_rotate_cu<" #L "," #NL "," #FIRST ",lean>";                   \
                hipLaunchKernelGGL((k_blind_rotate_cu<L, NL, FIRST, true>), grid, block, 0, stream, b); \
                FBS_TRACE_DUMP(kernel->c_str());                                                 \
                return true;                                                                     \
            }                                                                                    \
        }                                                                                        \
        *kernel = "k_blind_rotate_cu<" #L "," #NL "," #FIRST ">";                                \
        hipLaunchKernelGGL((k_blind_rotate_cu<L, NL, FIRST>), grid, block, 0, stream, b);        \
        FBS_TRACE_DUMP(kernel->c_str());                                                         \
        return true;                                                                             \
    }
    CU_CASE(10, 1, 0) CU_CASE(10, 1, 1) CU_CASE(10, 1, 2)
    CU_CASE(10, 2, 0) CU_CASE(10, 2, 1) CU_CASE(10, 2, 2)
    CU_CASE(10, 3, 0) CU_CASE(10, 3, 1) CU_CASE(10, 3, 2)
    CU_CASE(10, 4, 2)   // (l * beta <= 30: four levels have at most 7 bits each)
    CU_CASE(11, 1, 0) CU_CASE(11, 1, 1) CU_CASE(11, 1, 2)
    CU_CASE(11, 2, 0) CU_CASE(11, 2, 1) CU_CASE(11, 2, 2)
#undef CU_CASE
    return false;
}

#ifdef FBS_CU_TRACE
void cu_trace_dump(const char *kernel) {
    unsigned long long h[8 * 16];
    if (hipDeviceSynchronize() != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_cu_trace), sizeof h) != hipSuccess) return;
    fprintf(stderr, "trace %s (cycles per phase, workgroup 0, whole rotation):\n", kernel);
    for (int w = 0; w < 8; w++) {
        fprintf(stderr, "  wave %d:", w);
        for (int k = 0; k < 12; k++) fprintf(stderr, " %9llu", h[w * 16 + k]);
        fprintf(stderr, "\n");
    }
}
#endif

void blind_rotate_cu_catalog(std::vector<std::string> *out) {
    for (int nl = 1; nl <= 4; nl++)
        for (int first = nl == 4 ? 2 : 0; first < 3; first++) {
            out->push_back("k_blind_rotate_cu<10," + std::to_string(nl) + "," + std::to_string(first) + ">");
            if (nl <= 3) out->push_back("k_blind_rotate_cu<10," + std::to_string(nl) + "," + std::to_string(first) + ",lean>");
        }
    for (int nl = 1; nl <= 2; nl++)
        for (int first = 0; first < 3; first++)
            out->push_back("k_blind_rotate_cu<11," + std::to_string(nl) + "," + std::to_string(first) + ">");
    out->push_back("k_blind_rotate_cu_pairs<11,1>");
    out->push_back("k_blind_rotate_cu_pairs<11,2>");
}

}  // namespace fbs

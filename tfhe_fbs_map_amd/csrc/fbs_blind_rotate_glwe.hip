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
//   per gadget level:      balanced digit -> forward transform -> products with key row (c, level) for ALL k + 1 output components
//                          (two key bits: the key word is the bundle sum_jj (zeta^e_jj - 1) E_jj, zeta = the evaluation point the
//                          register holds)
//   hand-over:             the products for the k other components are ADDED into those components' landing words as they are made
//                          (ds_add_f64: exact on integer-valued doubles below 2^53, so the order does not matter); barrier; own
//                          products + what landed -> landing words cleared -> inverse transform -> accumulate.
// ONE workgroup barrier per step where LDS holds two sets of landing words taken in turns (N <= 512), two where it holds one.  The columns of a key row are taken in ROTATED order (d = 0 .. k stands for component
// c + d mod k + 1), so that "which product goes where" is the same code in every wave and no register array is indexed by a
// run-time value.  Same rounding rules and, word for word, the same ciphertexts as the oracle (tests/test_gpu_glwe.py).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <type_traits>
#include <vector>

#include "fbs_blind_rotate.hpp"

// chunks of key words in flight beyond the one being multiplied (measured at k = 3, N = 512, n = 614, ms per 1 536 bootstraps:
// 1: 7.28, 2: 7.54 (64-148 bytes spilled at the 168 registers of three waves per SIMD), 3: 9.62; k = 3 at N = 1024: 16.8 / 17.7 / 18.9)
#ifndef FBS_GLWE_CHUNKS_AHEAD
#define FBS_GLWE_CHUNKS_AHEAD 1
#endif
// ... with one bootstrap per workgroup (every wave alone on its SIMD, 512 registers to itself): k = 3, N = 512, n = 614, ms per launch of
// 64 / 256: 1: 1.56 / 1.78, 2: 1.50 / 1.70, 3: 1.50 / 1.73, 5: 1.53 / 1.77
#ifndef FBS_GLWE_CHUNKS_AHEAD_ALONE
#define FBS_GLWE_CHUNKS_AHEAD_ALONE 2
#endif
// -DFBS_EXP_GLWE_FLAT_PSI=1 (experiments only: WRONG results): the psi^x look-ups at conflict-free addresses
#ifndef FBS_EXP_GLWE_FLAT_PSI
#define FBS_EXP_GLWE_FLAT_PSI 0
#endif
#ifndef FBS_GLWE_L2_AHEAD
#define FBS_GLWE_L2_AHEAD 2
#endif

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
    constexpr int N = W::N, E = W::E, LANES = W::LANES, WAVES = FPW * K1, NS = GROUP == 2 ? 3 : 1;   // NS: GGSW samples per step
    static_assert(LANES == 64 && (E == 4 || E == 8 || E == 16), "one wave per polynomial");
    // LANDING words of the hand-over, one set per wave beside its exchange buffer: TWO sets taken in turns where LDS has the room
    // (then a set is cleared by its owner a whole step before anybody adds into it again, and ONE barrier per step is enough)
    constexpr int SETS = (size_t)(3 * WAVES + 3) * N * 8 <= 160 * 1024 ? 2 : 1;
    // [wave][N] exchange buffers (wave = K1 * bootstrap + component); [set][wave][N] landing words; forward and inverse per-lane
    // twiddle tables; psi^x, x < N (two key bits per step)
    __shared__ __attribute__((aligned(16))) double lds_all[(1 + SETS) * WAVES * N + 2 * N + (GROUP == 2 ? N : 0)];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // bootstrap of the workgroup, GLWE component.  (Measured and dropped: with four components the waves that hold the SAME component of
    // different bootstraps -- which ask for the same key words -- are four apart and would share a SIMD if the waves of a workgroup are
    // dealt round the SIMDs in order; bootstrap s taking its components rotated by s changed nothing: 5.55 / 11.10 against 5.55 / 11.11 ms.)
    const uint32_t sub = wave / (uint32_t)K1, comp = wave - (uint32_t)K1 * sub;
    const uint32_t t = threadIdx.x & 63u;
    double *mine = lds_all + wave * N;
    double *landing = lds_all + WAVES * N;
    double *tables = lds_all + (1 + SETS) * WAVES * N;
    typename W::Xchg xc{mine, 0};
    xc.stride = 0;
    Twiddles twf(tables, a.tw_fwd), twi(tables + N, a.tw_inv);
    for (uint32_t x = threadIdx.x; x < (uint32_t)N; x += 64u * WAVES) {
        tables[x] = a.tw_fwd[W::LANE_TABLE_OFFSET + x];
        tables[N + x] = a.tw_inv[W::LANE_TABLE_OFFSET + x];
        if constexpr (GROUP == 2) tables[2 * N + x] = a.psi_pow[x];
    }
    for (uint32_t x = threadIdx.x; x < (uint32_t)(SETS * WAVES * N); x += 64u * WAVES) landing[x] = 0.0;
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
    uint32_t col_bytes[K1], land_word[K1];
#pragma unroll
    for (int d = 0; d < K1; d++) {
        const uint32_t c = comp + (uint32_t)d >= (uint32_t)K1 ? comp + (uint32_t)d - (uint32_t)K1 : comp + (uint32_t)d;
        col_bytes[d] = c * (uint32_t)(N * 8);
        land_word[d] = (sub * (uint32_t)K1 + c) * (uint32_t)N;
    }
    const uint32_t rows = (uint32_t)K1 * a.l;                       // rows of a GGSW sample
    const uint32_t t16 = t * 16u;                                   // this thread's 16 bytes of a register pair's 64 lanes
    const uint32_t o_lane = Pos::exponent(Pos::lane(t)) + 1u;       // (two key bits per step: zeta = psi^(o_lane + c_m))
    const double *psi = tables + 2 * N;

    constexpr uint32_t STEP_BITS = GROUP;
    const uint32_t n_steps = a.n / STEP_BITS;
    // The key rows of a step come out of L2 if somebody has brought them there (the key lives in the Infinity Cache / HBM; with every
    // step reading the rows of steps 0 and 1 -- -DFBS_EXP_HOT_KEYS, a timing experiment -- k = 3 at N = 512 takes 4.62 ms per 768
    // against 5.55, at N = 1024 15.3 against 25.1).  As in k_blind_rotate_cu_pairs: the workgroups of an XCD (blockIdx mod 8), which
    // walk the key in step, each touch one line in every `peers` of the rows FBS_GLWE_L2_AHEAD steps ahead, behind the step's barrier
    // -- one load per thread, branch-free (a thread without a line asks beyond the end of the resource), its value never looked at.
    const uint32_t row_lines = (GROUP == 2 ? 3u : 1u) * rows * (uint32_t)K1 * (uint32_t)(N * 8 / 128);
    const uint32_t n_groups = (uint32_t)((a.count + FPW - 1) / FPW);
    const uint32_t peers = std::min<uint32_t>(32u, (n_groups + 7u) / 8u);
    const uint32_t slice_lines = std::min<uint32_t>(64u * WAVES, (row_lines + peers - 1u) / peers);
    const uint32_t ahead_line = ((blockIdx.x >> 3) % peers) * slice_lines + threadIdx.x;
    const uint32_t ahead_off = threadIdx.x < slice_lines && ahead_line < row_lines ? ahead_line * 128u : 0x7FFFFFF0u;
    uint32_t ahead_word = 0;
    uint32_t e0_next = ms[0], e1_next = GROUP == 2 ? ms[1] : 0u;
    uint32_t turn = 0;   // steps this bootstrap has executed (wave-uniform, the same in every wave of the bootstrap)
    for (uint32_t i = 0; i < n_steps; i++) {
        uint32_t e[3];
        e[0] = __builtin_amdgcn_readfirstlane(e0_next);
        e[1] = __builtin_amdgcn_readfirstlane(e1_next);
        // (ms has n + 1 entries: the last step reads the body word, or re-reads it, and ignores it)
        e0_next = ms[STEP_BITS * (i + 1) < a.n ? STEP_BITS * (i + 1) : a.n];
        if constexpr (GROUP == 2) e1_next = ms[2 * i + 3 < a.n ? 2 * i + 3 : a.n];
        if (e[0] == 0 && e[1] == 0) {   // nothing to add for this bootstrap: the others of the workgroup still meet their barriers
            if constexpr (FPW > 1) {
                if constexpr (SETS == 1) __syncthreads();
                __syncthreads();
            }
            continue;
        }
        // this step's set of landing words.  The sets take turns by the steps this BOOTSTRAP executes (not by i: a skipped step has no
        // barrier when the bootstrap is alone in its workgroup), so between two uses of a set there is always an executed step, whose
        // barrier every wave of the bootstrap passes after the owner has read and cleared the set
        double *land = landing + (SETS == 2 ? (turn & 1u) * (uint32_t)(WAVES * N) : 0u);
        turn++;
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
        // one set of landing words: everybody has read and cleared its own since the last step's additions
        if constexpr (SETS == 1) __syncthreads();

        // ---- level by level: digit, forward transform, products for the K1 output components: the own one summed in registers, the
        // others ADDED into those components' landing words as they are made (ds_add_f64: exact on integer-valued doubles below
        // 2^53 -- (k + 1) l products below 0.8 q each -- so the order in which they land does not matter) ---------------------------
        double own[E];
#pragma unroll
        for (int m = 0; m < E; m++) own[m] = 0.0;
        for (int lv = (int)a.l - 1; lv >= 0; lv--) {
            const uint32_t shift = (a.l - 1u - (uint32_t)lv) * a.beta;
            double x[E];
#pragma unroll
            for (int m = 0; m < E; m++) x[m] = (double)(int)__builtin_amdgcn_sbfe(digits[m], shift, a.beta);   // balanced digit in [-B/2, B/2)
            const uint32_t row_bytes = ((comp * a.l + (uint32_t)lv) * (uint32_t)K1) * (uint32_t)(N * 8);
            // row (comp, lv) of the sample(s) of step i: [sample][row][column][N] (one sample with one key bit per step, three with two)
            const KeyRows keys(a.bsk_hat + (size_t)FBS_KEY_STEP(i) * NS * rows * K1 * N);
            const uint32_t sample_bytes = rows * (uint32_t)K1 * (uint32_t)(N * 8);
            // The work of a level after its transform comes in CHUNKS (register pair j, column d): the key words of the chunk -- one per
            // sample -- times the pair's evaluations, for component comp + d.  The compiler keeps a buffer load behind every LDS atomic
            // written before it (it cannot tell the two apart), so written chunk by chunk -- load, multiply, add -- every load would wait
            // for its own round trip with nothing else in flight (the first form of this kernel: s_waitcnt vmcnt(0) behind each of the
            // 48 loads of a step; k = 3 at N = 512, n = 614: 8.9 ms per 1 536 bootstraps against 7.3 now).  Here the words of chunk c + AHEAD are asked for
            // BEFORE chunk c is multiplied, and the products of chunk c land while chunk c + 1 is multiplied: a chunk's words have AHEAD
            // chunks of arithmetic to arrive in.  The first chunks of a level are asked for ahead of its transform.
            constexpr int CHUNKS = (E / 2) * K1, WANT = FPW == 1 ? FBS_GLWE_CHUNKS_AHEAD_ALONE : FBS_GLWE_CHUNKS_AHEAD;
            constexpr int AHEAD = WANT < CHUNKS ? WANT : CHUNKS - 1, BUFS = AHEAD + 1;
            double2 kbuf[BUFS][NS];
            auto request = [&](int c, double2 (&k)[NS]) {
                const int j = c / K1, d = c % K1;
#pragma unroll
                for (int jj = 0; jj < NS; jj++)
                    k[jj] = keys.load(t16 + (uint32_t)(j * LANES * 16), (uint32_t)jj * sample_bytes + row_bytes + col_bytes[d]);
            };
#pragma unroll
            for (int c = 0; c < AHEAD; c++) request(c, kbuf[c % BUFS]);
            W::template forward<0>(x, xc, t, twf, typename W::NoHook{});
            double mono[3][2];
            double late0 = 0.0, late1 = 0.0;   // the products of the chunk before, not landed yet
#pragma unroll
            for (int c = 0; c < CHUNKS; c++) {
                const int j = c / K1, d = c % K1;
                if (c + AHEAD < CHUNKS) request(c + AHEAD, kbuf[(c + AHEAD) % BUFS]);
                if constexpr (GROUP == 2) {
                    if (d == 0) {
                        // zeta^e - 1 for the two registers of the pair and the three exponents: zeta = psi^(o_lane + c_m), psi^(x + N) = -psi^x;
                        // where the pair's evaluation points differ by psi^N = -1, one look-up per exponent serves both
                        const uint32_t c0 = Pos::exponent(Pos::reg(2 * j)), c1 = Pos::exponent(Pos::reg(2 * j + 1));   // (constants once unrolled)
                        const bool twins = c1 - c0 == (uint32_t)N;
#pragma unroll
                        for (int jj = 0; jj < 3; jj++) {
                            const uint32_t xx = (e[jj] * (o_lane + c0)) & (2u * N - 1u);
                            const double v = psi[FBS_EXP_GLWE_FLAT_PSI ? ((t + (uint32_t)jj) & (N - 1)) : (xx & (N - 1))];
                            const int hi = __double2hiint(v) ^ (int)((xx << (31 - LOGN)) & 0x80000000u);
                            mono[jj][0] = __hiloint2double(hi, __double2loint(v)) - 1.0;
                            if (twins) {
                                mono[jj][1] = __hiloint2double(hi ^ (int)(e[jj] << 31), __double2loint(v)) - 1.0;
                            } else {
                                const uint32_t yy = (e[jj] * (o_lane + c1)) & (2u * N - 1u);
                                const double u = psi[yy & (N - 1)];
                                mono[jj][1] = __hiloint2double(__double2hiint(u) ^ (int)((yy << (31 - LOGN)) & 0x80000000u), __double2loint(u)) - 1.0;
                            }
                        }
                    }
                }
                const double2(&kw)[NS] = kbuf[c % BUFS];
                double w0, w1;
                if constexpr (GROUP == 1) {
                    w0 = kw[0].x;
                    w1 = kw[0].y;
                } else {
                    // bundle words: lazy sums of three exact products (< 2.4 q); |x| < 2^49.3, so the products below stay exact
                    w0 = fp_mulmod(kw[0].x, mono[0][0]);
                    w1 = fp_mulmod(kw[0].y, mono[0][1]);
#pragma unroll
                    for (int jj = 1; jj < 3; jj++) {
                        w0 += fp_mulmod(kw[jj].x, mono[jj][0]);
                        w1 += fp_mulmod(kw[jj].y, mono[jj][1]);
                    }
                }
                const double p0 = fp_mulmod(x[2 * j], w0), p1 = fp_mulmod(x[2 * j + 1], w1);
                // the chunk before lands now (its column: d - 1, or the last one of the pair before)
                if (c > 0) {
                    const int jb = (c - 1) / K1, db = (c - 1) % K1;
                    if (db != 0) {
                        __hip_atomic_fetch_add(&land[land_word[db] + W::handoff_word(t, 2 * jb)], late0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(&land[land_word[db] + W::handoff_word(t, 2 * jb + 1)], late1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                if (d == 0) {
                    own[2 * j] += p0;
                    own[2 * j + 1] += p1;
                } else {
                    late0 = p0;
                    late1 = p1;
                }
            }
            {   // the last chunk (column K1 - 1 of the last pair: never the own one)
                constexpr int jb = (CHUNKS - 1) / K1, db = (CHUNKS - 1) % K1;
                __hip_atomic_fetch_add(&land[land_word[db] + W::handoff_word(t, 2 * jb)], late0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(&land[land_word[db] + W::handoff_word(t, 2 * jb + 1)], late1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        __syncthreads();
        if constexpr (FBS_GLWE_L2_AHEAD != 0) {
            const uint32_t far_step = i + FBS_GLWE_L2_AHEAD < n_steps ? i + FBS_GLWE_L2_AHEAD : n_steps - 1;
            const __amdgpu_buffer_rsrc_t far = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<double *>(a.bsk_hat + (size_t)far_step * (GROUP == 2 ? 3 : 1) * rows * K1 * N), 0, row_lines * 128u, 0x00020000);
            ahead_word = __builtin_amdgcn_raw_buffer_load_b32(far, ahead_off, 0, 0);
        }

        // ---- the total for this component; its landing words cleared for their next turn; back to coefficients; accumulate -----------
#pragma unroll
        for (int m = 0; m < E; m++) {
            double *word = &land[land_word[0] + W::handoff_word(t, m)];
            own[m] += *word;
            *word = 0.0;
        }
        W::template inverse<false>(own, xc, t, twi, W::inverse_uniform(t, twi));   // (centred by the transform first)
#pragma unroll
        for (int m = 0; m < E; m++) acc[m] = fp_center(acc[m] + own[m]);
    }

    if constexpr (FBS_GLWE_L2_AHEAD != 0) asm volatile("" ::"v"(ahead_word));   // (never looked at; this keeps the loads)

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

// Bootstraps per workgroup.  The THROUGHPUT shape fills a CU: twelve waves where the registers allow three waves per SIMD (N <= 512:
// 4, 3, 2 bootstraps at k = 2, 3, 4), six to eight at N = 1024.  Launches that leave most of the chip empty take ONE bootstrap per
// workgroup up to one per CU (every wave alone on its SIMD: a step is one wave's instruction chain, not three waves' sharing an issue
// port) and two up to two per CU.  Measured at k = 3, N = 512, n = 614 (the 128-bit set for p <= 4), ms per launch: 64 / 256 bootstraps
// 1.55 / 1.81 with one per workgroup against 3.32 / 3.37 with three; 512: 2.90 with two against 3.61; 768: 4.06 with three.  (FOUR per workgroup
// there -- sixteen waves at 128 registers with 196 bytes spilled, one set of landing words -- 6.99 against 5.43 ms per 1 024, 20.6 against 14.5 per 3 072.)
template <int LOGN, int K1>
constexpr int glwe_fpw() {
#ifdef FBS_EXP_GLWE_FPW       // (experiments: every shape)
    return FBS_EXP_GLWE_FPW;
#elif defined(FBS_EXP_GLWE_FPW_K3N512)   // (experiments: k = 3 at N = 512 only)
    return LOGN == 9 && K1 == 4 ? FBS_EXP_GLWE_FPW_K3N512 : LOGN >= 10 ? 2 : 12 / K1;
#else
    return LOGN >= 10 ? 2 : 12 / K1;
#endif
}

template <int LOGN, int K1, int GROUP, int FPW>
static void launch_fpw(const BrArgs &a, hipStream_t stream, std::string *kernel) {
    *kernel = "k_blind_rotate_glwe<" + std::to_string(LOGN) + "," + std::to_string(K1) + "," + std::to_string(GROUP) + "," + std::to_string(FPW) + ">";
    hipLaunchKernelGGL((k_blind_rotate_glwe<LOGN, K1, GROUP, FPW>), dim3((unsigned)((a.count + FPW - 1) / FPW)), dim3(64 * K1 * FPW), 0, stream, a);
}

template <int LOGN, int K1, int GROUP>
static void launch_one(int fpw, const BrArgs &a, hipStream_t stream, std::string *kernel) {
    constexpr int FULL = glwe_fpw<LOGN, K1>();
    if (fpw == 1) launch_fpw<LOGN, K1, GROUP, 1>(a, stream, kernel);
    else if (fpw == 2) launch_fpw<LOGN, K1, GROUP, (FULL < 2 ? FULL : 2)>(a, stream, kernel);
    else launch_fpw<LOGN, K1, GROUP, FULL>(a, stream, kernel);
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

// bootstraps per workgroup of the throughput shape (what a full round is made of: glwe_full_fpw x CUs)
int glwe_full_fpw(uint32_t log_n, uint32_t k) {
#define X(L, K) \
    if (log_n == L && k + 1 == K) return glwe_fpw<L, K>();
    FBS_GLWE_SHAPES(X)
#undef X
    return 0;
}

// fpw: bootstraps per workgroup (1, 2, or anything else for the throughput shape)
bool launch_blind_rotate_glwe(fbs_ctx *ctx, const BrArgs &a, int fpw, hipStream_t stream, std::string *kernel) {
    const fbs_params &p = ctx->p;
    if (p.k < 2 || !glwe_shape_built(p.log_n_poly, p.k)) return false;
#define X(L, K)                                                           \
    if (p.log_n_poly == L && p.k + 1 == K) {                              \
        if (ctx->group == 2) launch_one<L, K, 2>(fpw, a, stream, kernel); \
        else launch_one<L, K, 1>(fpw, a, stream, kernel);                 \
        return true;                                                      \
    }
    FBS_GLWE_SHAPES(X)
#undef X
    return false;
}

void blind_rotate_glwe_catalog(std::vector<std::string> *out) {
    std::vector<std::string> names;
#define X(L, K)                                  \
    for (int g = 1; g <= 2; g++)                 \
        for (int fpw : {1, 2, glwe_fpw<L, K>()}) \
            names.push_back("k_blind_rotate_glwe<" + std::to_string(L) + "," + std::to_string(K) + "," + std::to_string(g) + "," + std::to_string(fpw) + ">");
    FBS_GLWE_SHAPES(X)
#undef X
    // (a shape whose throughput form holds two bootstraps per workgroup lists "2" once)
    std::sort(names.begin(), names.end());
    names.erase(std::unique(names.begin(), names.end()), names.end());
    out->insert(out->end(), names.begin(), names.end());
}

}  // namespace fbs

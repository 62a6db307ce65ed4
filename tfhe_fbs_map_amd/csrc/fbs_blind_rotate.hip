// Blind rotation + sample extraction (the hot kernel), the one-time key transform and a debug product, all on
// the NTT of fbs_ntt.hpp / fbs_ntt_split.hpp.  gfx950 only.  Exact modular arithmetic on integer-valued doubles (fbs_field.hpp): the
// FP64 FMA is the machine's widest exact multiplier; no MFMA, no tensor contraction -- every product is an
// element-wise residue product.
//
// One functional bootstrap = two waves (or 2 x LANES/64) of a workgroup; GLWE component c (k = 1: mask, body) is owned by LANES lanes, each
// holding E coefficients of the accumulator in VGPRs.  Per CMUX step: accumulator -> LDS, gather the rotated copy
// (X^a), subtract, round to l*beta bits; per digit level: balanced digit -> forward NTT -> multiply-accumulate with
// the two key polynomials of that row (lazy sums); hand the partner component its half through LDS; inverse NTT;
// accumulate and canonicalise.  The bootstrapping-key row of a step (96 KB at P1024) is read once per workgroup
// with 16-byte coalesced loads; all workgroups walk the key in step, so after the first touch it is served from
// L2/MALL, not HBM.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "fbs_blind_rotate.hpp"

namespace fbs {

// ---------------------------------------------------------------------------------------------
template <int LOGN, int LL>
__global__ __launch_bounds__(1 << LL) void k_bsk_transform(const uint64_t *__restrict__ src, double *__restrict__ dst,
                                                           const double *__restrict__ tw_fwd, double n_inv, size_t polys) {
    using W = typename NttFor<LOGN, LL>::type;
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

template <int LOGN, int LL>
__global__ __launch_bounds__(1 << LL) void k_polymul(const uint64_t *a, const uint64_t *b, uint64_t *c, const double *tw_fwd,
                                                     const double *tw_inv, double n_inv) {
    using W = typename NttFor<LOGN, LL>::type;
    __shared__ double lds[2 * W::N];
    const uint32_t t = threadIdx.x;
    typename W::Xchg xc{lds, 0};
    double x[W::E], y[W::E];
#pragma unroll
    for (int m = 0; m < W::E; m++) {
        x[m] = fp_from_u64(a[W::template index_of<0>(t, m)]);
        y[m] = fp_from_u64(b[W::template index_of<0>(t, m)]);
    }
    W::forward(x, xc, t, Twiddles(tw_fwd + W::LANE_TABLE_OFFSET, tw_fwd));
    W::forward(y, xc, t, Twiddles(tw_fwd + W::LANE_TABLE_OFFSET, tw_fwd));
#pragma unroll
    for (int m = 0; m < W::E; m++) x[m] = fp_mulmod(fp_mulmod(x[m], fp_center(y[m])), n_inv);
    W::inverse(x, xc, t, Twiddles(tw_inv + W::LANE_TABLE_OFFSET, tw_inv));
#pragma unroll
    for (int m = 0; m < W::E; m++) c[W::template index_of<0>(t, m)] = fp_to_u64(fp_canon(x[m]));
}

// ---------------------------------------------------------------------------------------------
// DIG: what the launcher knows about the gadget --
//   0  nothing;
//   1  l <= 5: the 2l lazy products (each below 0.8 q) that enter the inverse transform stay below 8 q (its first centring
//      pass is spared);
//   2  also beta <= 9: a balanced digit (|d| <= 2^8) times a twiddle (|w| <= 2^45) is exact in a double;
//   3  also beta <= 7: the first butterfly stage of the forward transforms is two exact FMAs (first_butterfly, fbs_ntt.hpp)
//   4  l = 1 (one wide digit, any beta: the shape the 128-bit parameter sets take at N = 2048): as 1, without the loop
//      over further levels -- and without the registers the compiler keeps alive for it
//   5, 6, 7  l = 2 (the 128-bit sets for p <= 4 at N = 1024 and for p = 31, 63): as 1, 2, 3 with the two levels written out
//      (no spills at N = 1024 where the loop form spills 21 registers, 13 instead of 27 at N = 2048)
// FPW: bootstraps per workgroup.  The hardware deals the waves of a workgroup round the four SIMDs of a CU but starts
// every workgroup at the same SIMD often enough that two-wave workgroups pile up on two SIMDs while the other two
// idle whenever a CU holds fewer than four of them (measured: 512 bootstraps took 9.7 ms, 256 took 5.7 ms); four-wave
// workgroups (two bootstraps) always cover all four SIMDs.
// TURNS: the two waves of a SIMD hand the issue priority back and forth (lead_if).  It pays while a launch is a round or two
// long; small workgroups of a long launch are refilled as they finish and do better left alone -- and the mere presence of
// s_setprio costs the compiler's schedule 7 % there (N = 1024, l = 2, 8192 bootstraps: 118 k FBS/s with, 134 k without;
// 1024 bootstraps: 129 against 125), so the launcher picks an instantiation, not a flag.
template <int LOGN, int LL, int DIG, int FPW, bool TURNS = true>
__global__ __launch_bounds__((2 << LL) * FPW) __attribute__((amdgpu_waves_per_eu(2))) void k_blind_rotate(BrArgs a) {
    using W = typename NttFor<LOGN, LL>::type;
    constexpr int FIRST = (DIG == 3 || DIG == 7) ? 2 : (DIG == 2 || DIG == 6) ? 1 : 0;
    constexpr bool BOUNDED = DIG >= 1;
    constexpr bool ONE_LEVEL = DIG == 4, TWO_LEVELS = DIG >= 5;
#ifndef FBS_PEEL_MAX_LL
#define FBS_PEEL_MAX_LL 6   // measured: peeling costs the two-waves-per-polynomial shapes more in spills than it saves
#endif
    // first level peeled off the loop (it assigns the sums instead of adding to zeros): pays where registers allow
    constexpr bool PEEL = ONE_LEVEL || TWO_LEVELS || LL <= FBS_PEEL_MAX_LL;
    constexpr int N = W::N, E = W::E, LANES = W::LANES;
    __shared__ double lds_all[FPW * 2 * 2 * N];   // [bootstrap][component][ping-pong][N]
    const uint32_t sub = threadIdx.x >> (LL + 1);          // which bootstrap of the workgroup
    const uint32_t comp = __builtin_amdgcn_readfirstlane((threadIdx.x >> LL) & 1u);        // GLWE component owned by this thread: 0 = mask, 1 = body
    const uint32_t t = threadIdx.x & (LANES - 1);
    double *lds = lds_all + sub * (2 * 2 * N);
    double *mine = lds + comp * 2 * N;
    double *theirs = lds + (comp ^ 1u) * 2 * N;
    typename W::Xchg xc{mine, 0};
    Twiddles twf(a.tw_fwd + W::LANE_TABLE_OFFSET, a.tw_fwd), twi(a.tw_inv + W::LANE_TABLE_OFFSET, a.tw_inv);
    if constexpr (LL <= FBS_ONE_BUFFER_MAX_LL) {
        // One exchange buffer per polynomial (fbs_ntt.hpp); the other half of each component's region
        // holds a twiddle table instead (forward in component 0's, inverse in component 1's), so the per-lane
        // twiddle gathers are LDS reads, not 64-address global loads.
        xc.stride = 0;
        double *table = mine + N;
        const double *src = (comp ? a.tw_inv : a.tw_fwd) + W::LANE_TABLE_OFFSET;
#pragma unroll
        for (int m = 0; m < E; m++) table[t + (uint32_t)LANES * m] = src[t + (uint32_t)LANES * m];
        __syncthreads();
        twf.lane = lds + N;
        twi.lane = lds + 3 * N;
    }

    // a workgroup past the end of an odd batch repeats the last bootstrap (its waves must keep meeting the others at
    // the barriers) and writes nothing
    const size_t f_want = (size_t)blockIdx.x * FPW + sub;
    const bool live = f_want < a.count;
    const size_t f = live ? f_want : a.count - 1;
    size_t gate, ms_row;
    gate_of(a.gv, f, &gate, &ms_row);
    // ids that arrive in device memory cannot be validated by the host: an id past the set reads table 0, never past
    // the end of the buffer
    uint32_t table = a.gv.table_ids ? a.gv.table_ids[gate] : 0;
    if (table >= a.n_tables) table = 0;
    const uint32_t *ms = a.ms + ms_row * (a.n + 1);
    const uint64_t *tv = a.tvs + (size_t)table * N;
    const uint32_t rows = 2 * a.l;

    // ACC = (0, X^{-b~} * TV), kept CENTRED (|.| <= (q-1)/2) for the whole rotation; register m of lane t is
    // coefficient t + LANES*m
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

    // Rounding runs on doubles too (floor(x * 2^-s + c) is exact on integers): abar = (d + 2^(s-1)) >> s with
    // s = 46 - l*beta, d = the difference of two centred residues (an integer in (-q, q), not reduced again), abar
    // kept mod B^l -- B^l is added so that the value converted to an unsigned word is positive (l*beta <= 30).  Adding B/2 at every digit position turns the
    // balanced digits (each in [-B/2, B/2), carries included) into plain bit fields:
    //     digit_j = ((abar + (B/2)(1 + B + .. + B^(l-1))) >> j*beta) mod B - B/2.
    const double round_scale = fp_exp2i(-(int)(FQ_BITS - a.l * a.beta));
    // Flipping the top bit of every field then leaves digit_j in two's complement, ready for a signed bit-field extract.
    const uint32_t bhalf = 1u << (a.beta - 1);
    double round_offset = 0.5 + fp_exp2i((int)(a.l * a.beta));
    uint32_t sign_bits = 0;
    for (uint32_t j = 0; j < a.l; j++) {
        round_offset += (double)(bhalf << (j * a.beta));
        sign_bits |= bhalf << (j * a.beta);
    }

    const uint32_t t16 = t * 16u;   // this thread's 16 bytes of a register pair's 16 LANES
    uint32_t r_next = ms[0];   // the rotation amount of a step is fetched one step ahead: its latency is never exposed
    constexpr int PRIO = !TURNS ? 0 : LL <= 6 ? FBS_PRIO_ONE_WAVE : FBS_PRIO_MULTI_WAVE;
    const uint32_t slot = PRIO ? wave_slot_parity() : 0u;
    for (uint32_t i = 0; i < a.n; i++) {
        lead_if<PRIO>(PRIO == 7 ? 2u * i : i, slot);
        const uint32_t r = __builtin_amdgcn_readfirstlane(r_next);
        r_next = ms[i + 1];     // ms has n+1 entries; the last one (the body) is read here and ignored
        if (r == 0) {           // X^0 * ACC - ACC = 0: nothing to add (uniform over the two waves of a bootstrap)
            if constexpr (FPW > 1) {   // the other bootstrap of the workgroup still meets its two barriers of this step
                __syncthreads();
                __syncthreads();
            }
            continue;
        }

        // ---- (X^r - 1) * ACC_c, centred, rounded to the closest multiple of q / B^l ---------------
        uint32_t digits[E];
        {
            double *buf = xc.next();
            if constexpr (LL <= FBS_ONE_BUFFER_MAX_LL) W::sync();
            // both the store and the rotated read walk consecutive words across the lanes: no swizzle needed here
#pragma unroll
            for (int m = 0; m < E; m++) buf[t + (uint32_t)LANES * m] = acc[m];
            W::sync();
            const uint32_t from = (t - r) & (2u * N - 1u);   // coefficient t of X^r * ACC is +-ACC[(t - r) mod 2N]
#pragma unroll
            for (int m = 0; m < E; m++) {
                const uint32_t idx = from + (uint32_t)LANES * m;   // < 3N: bit LOGN = sign, bits below = position
                const double w = buf[idx & (N - 1)];
                const double v = __hiloint2double(__double2hiint(w) ^ (int)((idx << (31 - LOGN)) & 0x80000000u), __double2loint(w));
                // Signed representatives, not canonical ones: the rounding below treats q as 2^46, an error proportional
                // to the value -- of one sign on [0, q) (it then adds up coherently through the key bits), symmetric
                // here.  v and acc are centred, so the difference needs no reduction of its own.
                const double d = v - acc[m];                     // in (-q, q)
                digits[m] = (uint32_t)__builtin_fma(d, round_scale, round_offset) ^ sign_bits;   // truncation = floor, < 3 * 2^(l*beta)
            }
        }

        // ---- digits, least significant level first; NTT; multiply-accumulate with the key row ---
        double own[E], other[E];   // contributions to this component and to the partner's (lazy sums)
        // one gadget level; ASSIGN: the first one initialises the sums instead of adding to zeros
        auto level = [&](int lv, auto assign) {
            constexpr bool ASSIGN = decltype(assign)::value;
            const uint32_t shift = ((uint32_t)a.l - 1u - (uint32_t)lv) * a.beta;
            double x[E];
#pragma unroll
            for (int m = 0; m < E; m++)
                x[m] = (double)(int)__builtin_amdgcn_sbfe(digits[m], shift, a.beta);   // balanced digit in [-B/2, B/2)
            // (buffer loads, KeyRows: the step's rows behind one resource, the polynomial picked by a scalar byte offset)
            const KeyRows keys(a.bsk_hat + (size_t)FBS_KEY_STEP(i) * rows * 2 * N);
            const uint32_t krow = ((comp * a.l + (uint32_t)lv) * 2u) * (uint32_t)(N * 8);
            const uint32_t k_own = krow + comp * (uint32_t)(N * 8), k_oth = krow + (comp ^ 1u) * (uint32_t)(N * 8);
            // The first half of the "own" key polynomial is requested before the last butterfly group of the transform
            // (one group ~ one L2 round trip), the rest after it.  Measured on MI355X: 13.5 -> 12.8 ms per 1024-batch;
            // asking for more ahead of time (all of it, or the partner's polynomial too) spills and loses again.
            constexpr int EARLY = E / 4;
            double2 ko[E / 2];
            W::template forward<FIRST>(x, xc, t, twf, [&] {
#pragma unroll
                for (int m = 0; m < EARLY; m++) ko[m] = keys.load(t16 + (uint32_t)(m * LANES * 16), k_own);
            });
#pragma unroll
            for (int m = EARLY; m < E / 2; m++) ko[m] = keys.load(t16 + (uint32_t)(m * LANES * 16), k_own);
            // own products first; each finished pair frees the registers its key words sat in, and the partner's
            // key words are requested into them while the remaining own products run
            double2 kt[E / 2];
#pragma unroll
            for (int j = 0; j < E / 2; j++) {
                const double p0 = fp_mulmod(x[2 * j], ko[j].x), p1 = fp_mulmod(x[2 * j + 1], ko[j].y);
                own[2 * j] = ASSIGN ? p0 : own[2 * j] + p0;
                own[2 * j + 1] = ASSIGN ? p1 : own[2 * j + 1] + p1;
                kt[j] = keys.load(t16 + (uint32_t)(j * LANES * 16), k_oth);
            }
#pragma unroll
            for (int j = 0; j < E / 2; j++) {
                const double p0 = fp_mulmod(x[2 * j], kt[j].x), p1 = fp_mulmod(x[2 * j + 1], kt[j].y);
                other[2 * j] = ASSIGN ? p0 : other[2 * j] + p0;
                other[2 * j + 1] = ASSIGN ? p1 : other[2 * j + 1] + p1;
            }
        };
        if constexpr (TWO_LEVELS) {
            level(1, std::true_type{});
            level(0, std::false_type{});
        } else if constexpr (PEEL) {
            level((int)a.l - 1, std::true_type{});
            if constexpr (!ONE_LEVEL)
                for (int lv = (int)a.l - 2; lv >= 0; lv--) level(lv, std::false_type{});
        } else {
#pragma unroll
            for (int m = 0; m < E; m++) own[m] = other[m] = 0.0;
            for (int lv = (int)a.l - 1; lv >= 0; lv--) level(lv, std::false_type{});
        }

        // ---- hand the partner its half of the external product (same ping-pong slot in both regions) ----
        // (the wave-uniform twiddles of the inverse transform are requested first: scalar loads, in flight across the
        // two barriers instead of in front of the first butterflies)
        const typename W::InvUniform inv_uni = W::inverse_uniform(t, twi);
        {
            const uint32_t slot = xc.pp ? xc.stride : 0;
            xc.pp ^= 1u;
            // one-buffer exchanges: the partner may still be reading its buffer, which is where this hand-off lands
            if constexpr (LL <= FBS_ONE_BUFFER_MAX_LL) __syncthreads();
#pragma unroll
            for (int m = 0; m < E; m++) theirs[slot + W::handoff_word(t, m)] = other[m];
            __syncthreads();
#pragma unroll
            for (int m = 0; m < E; m++) own[m] += mine[slot + W::handoff_word(t, m)];
        }

        if constexpr (PRIO == 7) lead_if<PRIO>(2u * i + 1u, slot);
        // ---- back to coefficients (the 1/N is folded into the key) and accumulate ---------------
        W::template inverse<BOUNDED>(own, xc, t, twi, inv_uni);
#pragma unroll
        for (int m = 0; m < E; m++) acc[m] = fp_center(acc[m] + own[m]);   // |.| <= 17 q -> centred
    }

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
// Two key bits per step ("multi-bit" blind rotation, bsk_group = 2).  With E0, E1, E2 the GGSW samples of s0(1-s1),
// (1-s0)s1, s0 s1 for a pair (s0, s1) of key bits and (a0, a1) the pair's rotation amounts,
//     ACC += [ (X^a0 - 1) E0 + (X^a1 - 1) E1 + (X^(a0+a1) - 1) E2 ]  (x)  ACC.
// The bundle in brackets is built in the NTT domain, where X^e - 1 is the pointwise factor zeta^e - 1 (zeta = the
// evaluation point a register holds = psi^(2 bitrev(P) + 1) at array position P; looked up in a table of psi^x - 1), and
// ACC itself is decomposed: no rotate-and-subtract step, no LDS round trip for it, and HALF the transforms per key bit.
// Costs: three key polynomials read and three exact products per key word built (per pair of bits), 1.5x the key, and a
// step's key-noise term triples (params.variances).  Same shapes, layouts and transforms as k_blind_rotate; DIG as there
// (0, 3 and 4 are built).
template <int LOGN, int LL, int DIG>
__global__ __launch_bounds__(2 << LL) __attribute__((amdgpu_waves_per_eu(2))) void k_blind_rotate_pairs(BrArgs a) {
    using W = typename NttFor<LOGN, LL>::type;
    static_assert(W::HAS_EVAL_POSITION, "the bundle needs to know which evaluation point a register holds");
    constexpr int N = W::N, E = W::E, LANES = W::LANES;
    constexpr int FIRST = DIG == 3 ? 2 : 0;
    constexpr bool ONE_LEVEL = DIG == 4;
    // [component][exchange buffer | twiddle table][N], then psi^x for x < N (psi^(x+N) = -psi^x): 5 N words = 40 KB at
    // N = 1024, 80 KB at N = 2048 -- exactly what lets 8 waves share a CU's 160 KB.  The psi table is stored TRANSPOSED,
    // word (x mod G) * N/G + x / G with G = 2^EVAL_GROUP_LOG2: the exponents a wave gathers for one register are G * (e *
    // lane permutation) + uniform, an arithmetic progression of stride G e -- in natural order all 64 lanes of an odd e would
    // fall on 4 of the 32 bank pairs (measured: 2.75e9 conflict cycles per launch, the LDS pipe 66 % busy); transposed they
    // walk the banks with stride e (two-way for odd e, which a 64-lane 8-byte read is anyway; at worst eight-way).
    __shared__ __attribute__((aligned(16384))) double lds_all[2 * 2 * N + N];
    constexpr int GLOG = W::EVAL_GROUP_LOG2, G = 1 << GLOG;
    const uint32_t comp = __builtin_amdgcn_readfirstlane((threadIdx.x >> LL) & 1u);
    const uint32_t t = threadIdx.x & (LANES - 1);
    double *lds = lds_all;
    double *mine = lds + comp * 2 * N;
    double *theirs = lds + (comp ^ 1u) * 2 * N;
    typename W::Xchg xc{mine, 0};
    Twiddles twf(a.tw_fwd + W::LANE_TABLE_OFFSET, a.tw_fwd), twi(a.tw_inv + W::LANE_TABLE_OFFSET, a.tw_inv);
    static_assert(LL <= FBS_ONE_BUFFER_MAX_LL, "one exchange buffer per polynomial, twiddle tables beside it");
    {
        xc.stride = 0;
        double *table = mine + N;
        const double *src = (comp ? a.tw_inv : a.tw_fwd) + W::LANE_TABLE_OFFSET;
#pragma unroll
        for (int m = 0; m < E; m++) table[t + (uint32_t)LANES * m] = src[t + (uint32_t)LANES * m];
#pragma unroll
        for (int m = 0; m < E / 2; m++) {   // the two components' threads copy half of the psi table each
            const uint32_t x = comp * (N / 2) + t + (uint32_t)LANES * m;
            lds[4 * N + (x & (G - 1)) * (N / G) + (x >> GLOG)] = a.psi_pow[x];
        }
        __syncthreads();
        twf.lane = lds + N;
        twi.lane = lds + 3 * N;
    }

    const bool live = (size_t)blockIdx.x < a.count;
    const size_t f = live ? (size_t)blockIdx.x : a.count - 1;
    size_t gate, ms_row;
    gate_of(a.gv, f, &gate, &ms_row);
    uint32_t table = a.gv.table_ids ? a.gv.table_ids[gate] : 0;
    if (table >= a.n_tables) table = 0;
    const uint32_t *ms = a.ms + ms_row * (a.n + 1);
    const uint64_t *tv = a.tvs + (size_t)table * N;
    const uint32_t rows = 2 * a.l;

    double acc[E];   // ACC = (0, X^{-b~} * TV), centred; register m of lane t = coefficient t + LANES*m
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
    const double round_scale = fp_exp2i(-(int)(FQ_BITS - a.l * a.beta));
    const uint32_t bhalf = 1u << (a.beta - 1);
    double round_offset = 0.5 + fp_exp2i((int)(a.l * a.beta));
    uint32_t sign_bits = 0;
    for (uint32_t j = 0; j < a.l; j++) {
        round_offset += (double)(bhalf << (j * a.beta));
        sign_bits |= bhalf << (j * a.beta);
    }
    // zeta^e for the evaluation point zeta = psi^o a register holds: exponent x = e * o mod 2N, o = o_lane + c_m with
    // o_lane = 2 bitrev(lane part of the array position) + 1 = G k_lane + r_lane (r_lane wave-uniform) and c_m = 2 bitrev(
    // register part), a compile-time constant.  x = G a + b with b wave-uniform; table word b N/G + (a mod N/G), sign = bit
    // log2(N/G) of a.  Everything but e * k_lane is scalar arithmetic.
    const uint32_t o_lane = 2u * (__builtin_bitreverse32(W::eval_position_lane(t)) >> (32 - LOGN)) + 1u;
    const uint32_t k_lane8 = (o_lane >> GLOG) << 3;
    const uint32_t r_lane = __builtin_amdgcn_readfirstlane(o_lane & (G - 1));
    constexpr uint32_t MASK8 = (uint32_t)(N / G - 1) << 3;
    const uint32_t psi_base = (uint32_t)(uintptr_t)(lds + 4 * N);   // LDS byte address of the table (16 KB aligned)

    const uint32_t t16 = t * 16u;   // this thread's 16 bytes of a register pair's 16 LANES
    const uint32_t n_pairs = a.n / 2;
    uint32_t e0_next = ms[0], e1_next = ms[1];
    constexpr int PRIO = LL <= 6 ? FBS_PRIO_ONE_WAVE : FBS_PRIO_MULTI_WAVE;
    const uint32_t slot = PRIO ? wave_slot_parity() : 0u;
    for (uint32_t i = 0; i < n_pairs; i++) {
        lead_if<PRIO>(PRIO == 7 ? 2u * i : i, slot);
        uint32_t e[3];
        e[0] = __builtin_amdgcn_readfirstlane(e0_next);
        e[1] = __builtin_amdgcn_readfirstlane(e1_next);
        e0_next = ms[2 * i + 2 < a.n ? 2 * i + 2 : a.n];   // (the last pair re-reads the body word and ignores it)
        e1_next = ms[2 * i + 3 < a.n ? 2 * i + 3 : a.n];
        if (e[0] == 0 && e[1] == 0) continue;               // the bundle is zero (uniform over the bootstrap's waves)
        e[2] = (e[0] + e[1]) & (2u * N - 1u);
        uint32_t lane8[3];
#pragma unroll
        for (int jj = 0; jj < 3; jj++) lane8[jj] = __umul24(e[jj], k_lane8);

        // ---- ACC_c itself, rounded to the closest multiple of q / B^l; packed balanced digits -------------
        uint32_t digits[E];
#pragma unroll
        for (int m = 0; m < E; m++) digits[m] = (uint32_t)__builtin_fma(acc[m], round_scale, round_offset) ^ sign_bits;

        double own[E], other[E];
        auto level = [&](int lv, auto assign) {
            constexpr bool ASSIGN = decltype(assign)::value;
            const uint32_t shift = ((uint32_t)a.l - 1u - (uint32_t)lv) * a.beta;
            double x[E];
#pragma unroll
            for (int m = 0; m < E; m++) x[m] = (double)(int)__builtin_amdgcn_sbfe(digits[m], shift, a.beta);
            // (the exponents are "redefined" here so that the 48 uniform products e * c_m below are computed where they are
            // used, level by level, instead of being hoisted out of the level loop into more SGPRs than there are)
            uint32_t e_lv[3] = {e[0], e[1], e[2]};
            if constexpr (!ONE_LEVEL) asm volatile("" : "+s"(e_lv[0]), "+s"(e_lv[1]), "+s"(e_lv[2]));
            // the step's key rows (three samples of `rows` rows of two polynomials) behind one buffer resource; own / partner's
            // polynomial of this component's row of sample jj: scalar byte offsets
            const KeyRows keys(a.bsk_hat + (size_t)FBS_KEY_STEP(i) * 3 * rows * 2 * N);
            uint32_t k_own[3], k_oth[3];
#pragma unroll
            for (int jj = 0; jj < 3; jj++) {
                const uint32_t krow = (((uint32_t)jj * rows + comp * a.l + (uint32_t)lv) * 2u) * (uint32_t)(N * 8);
                k_own[jj] = krow + comp * (uint32_t)(N * 8);
                k_oth[jj] = krow + (comp ^ 1u) * (uint32_t)(N * 8);
            }
            // what one register pair (2j, 2j+1) needs from memory: its words of the six key polynomials ...
            struct PairKeys {
                double2 ko[3], kt[3];
            };
            auto request = [&](auto jc, PairKeys &in) {
                constexpr int j = decltype(jc)::value;
#pragma unroll
                for (int jj = 0; jj < 3; jj++) {
                    in.ko[jj] = keys.load(t16 + (uint32_t)(j * LANES * 16), k_own[jj]);
                    in.kt[jj] = keys.load(t16 + (uint32_t)(j * LANES * 16), k_oth[jj]);
                }
            };
            auto consume = [&](auto jc, const PairKeys &in) {
                constexpr int j = decltype(jc)::value;
                // ... and zeta^e - 1 for its two registers and the three exponents, from the table of psi^x in LDS
                // One look-up per exponent serves BOTH registers of the pair: 2j and 2j + 1 differ in bit 0 of their array position,
                // i.e. in the top bit of its reversal, so their evaluation points differ by psi^N = -1 and zeta_(2j+1)^e = (-1)^e
                // zeta_(2j)^e -- a wave-uniform sign (round 3: half the gathers from the table, and half their bank conflicts).
                static_assert(W::eval_position_reg(1) - W::eval_position_reg(0) == 1, "registers 2j, 2j+1 hold neighbouring array positions");
                double mono[3][2];
#pragma unroll
                for (int jj = 0; jj < 3; jj++) {
                    const uint32_t c_m = 2u * (__builtin_bitreverse32(W::eval_position_reg(2 * j)) >> (32 - LOGN));
                    const uint32_t rsum = r_lane + (c_m & (G - 1));                       // uniform from here ...
                    const uint32_t ku = (c_m >> GLOG) + (rsum >> GLOG), ru = rsum & (G - 1);
                    const uint32_t eru = e_lv[jj] * ru;
                    const uint32_t u8 = (e_lv[jj] * ku + (eru >> GLOG)) << 3;
                    const uint32_t sbase = psi_base + (eru & (G - 1)) * (uint32_t)(N / G * 8);
                    const uint32_t odd = e_lv[jj] << 31;                                   // ... to here
                    const uint32_t t8 = lane8[jj] + u8;
                    const double v = *reinterpret_cast<const __attribute__((address_space(3))) double *>((t8 & MASK8) | sbase);
                    // (+-) by the bit above the table index, then - 1
                    const int hi = __double2hiint(v) ^ (int)((t8 << (31 - 3 - (LOGN - GLOG))) & 0x80000000u);
                    mono[jj][0] = __hiloint2double(hi, __double2loint(v)) - 1.0;
                    mono[jj][1] = __hiloint2double(hi ^ (int)odd, __double2loint(v)) - 1.0;
                }
                // key words of the bundle: lazy sums of three exact products (< 2.3 q).  With |x| < 2^49.3 (general first
                // stage) the product below stays within 0.9 q as it is; the two-FMA first stage leaves |x| near 2^51 and
                // wants the word centred first.
                // (the first product initialises the sums: 0.0 + x is an instruction the compiler may not drop, -0.0 being a double)
                double wo0 = fp_mulmod(in.ko[0].x, mono[0][0]), wo1 = fp_mulmod(in.ko[0].y, mono[0][1]);
                double wt0 = fp_mulmod(in.kt[0].x, mono[0][0]), wt1 = fp_mulmod(in.kt[0].y, mono[0][1]);
#pragma unroll
                for (int jj = 1; jj < 3; jj++) {
                    wo0 += fp_mulmod(in.ko[jj].x, mono[jj][0]);
                    wo1 += fp_mulmod(in.ko[jj].y, mono[jj][1]);
                    wt0 += fp_mulmod(in.kt[jj].x, mono[jj][0]);
                    wt1 += fp_mulmod(in.kt[jj].y, mono[jj][1]);
                }
                if constexpr (FIRST == 2) {
                    wo0 = fp_center(wo0);
                    wo1 = fp_center(wo1);
                    wt0 = fp_center(wt0);
                    wt1 = fp_center(wt1);
                }
                const double p0 = fp_mulmod(x[2 * j], wo0), p1 = fp_mulmod(x[2 * j + 1], wo1);
                const double r0 = fp_mulmod(x[2 * j], wt0), r1 = fp_mulmod(x[2 * j + 1], wt1);
                own[2 * j] = ASSIGN ? p0 : own[2 * j] + p0;
                own[2 * j + 1] = ASSIGN ? p1 : own[2 * j + 1] + p1;
                other[2 * j] = ASSIGN ? r0 : other[2 * j] + r0;
                other[2 * j + 1] = ASSIGN ? r1 : other[2 * j + 1] + r1;
            };
            // The key words of pair 0 are requested before the last butterfly group of the transform, those of pair j right
            // before it is consumed (keeping pair j+1 in flight as well was measured: 35 registers spilled, no gain).
            PairKeys in;
            W::template forward<FIRST>(x, xc, t, twf, [&] { request(std::integral_constant<int, 0>{}, in); });
            static_assert(E == 16, "eight register pairs, written out");
            // (round 3, with the buffer loads' forty spare registers: pair j + 1 requested before pair j is consumed -- 9.21 against 9.23 ms
            // per 1024 bootstraps at the 128-bit p = 15 set, i.e. nothing: left as it was)
#define FBS_PAIR_STEP(J)                                                                                    \
    if constexpr ((J) > 0) request(std::integral_constant<int, (J)>{}, in);                                 \
    consume(std::integral_constant<int, (J)>{}, in);                                                        \
    if constexpr (!ONE_LEVEL) __builtin_amdgcn_sched_barrier(0);
            FBS_PAIR_STEP(0) FBS_PAIR_STEP(1) FBS_PAIR_STEP(2) FBS_PAIR_STEP(3)
            FBS_PAIR_STEP(4) FBS_PAIR_STEP(5) FBS_PAIR_STEP(6) FBS_PAIR_STEP(7)
#undef FBS_PAIR_STEP
        };
        level((int)a.l - 1, std::true_type{});
        if constexpr (!ONE_LEVEL)
            for (int lv = (int)a.l - 2; lv >= 0; lv--) level(lv, std::false_type{});

        // ---- hand the partner its half, inverse transform, accumulate: as in k_blind_rotate ---------------
        const typename W::InvUniform inv_uni = W::inverse_uniform(t, twi);
        {
            __syncthreads();
#pragma unroll
            for (int m = 0; m < E; m++) theirs[W::handoff_word(t, m)] = other[m];
            __syncthreads();
#pragma unroll
            for (int m = 0; m < E; m++) own[m] += mine[W::handoff_word(t, m)];
        }
        if constexpr (PRIO == 7) lead_if<PRIO>(2u * i + 1u, slot);
        W::template inverse<true>(own, xc, t, twi, inv_uni);   // 2l products below 0.8 q each: l <= 5 (the launcher checks)
#pragma unroll
        for (int m = 0; m < E; m++) acc[m] = fp_center(acc[m] + own[m]);
    }

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
// launchers
// ---------------------------------------------------------------------------------------------
#define FBS_FOR_EACH_SHAPE(X) X(8) X(9) X(10) X(11) X(12)

template <int LOGN>
static int upload_keys_t(fbs_ctx *ctx) {
    constexpr int LL = lanes_log2_for(LOGN);
    const fbs_params &p = ctx->p;
    const uint32_t N = ctx->N;
    const size_t polys = ctx->n_ggsw * ctx->rows * (p.k + 1);
    uint64_t *d_src = nullptr;
    FBS_HIP(ctx, hipMalloc(&d_src, polys * N * 8));
    hipError_t e = hipMemcpyAsync(d_src, ctx->bsk.data(), polys * N * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        const double n_inv = fq_centered(fq_inv(N));
        unsigned grid = (unsigned)std::min<size_t>(polys, 4096);
        hipLaunchKernelGGL((k_bsk_transform<LOGN, LL>), dim3(grid), dim3(1 << LL), 0, ctx->stream, d_src,
                           reinterpret_cast<double *>(ctx->d_bsk_hat), reinterpret_cast<const double *>(ctx->d_tw_fwd), n_inv,
                           polys);
        e = hipGetLastError();
        constexpr int LLS = lanes_log2_for_small_launch(LOGN);
        if constexpr (LLS != LL) {
            // (two key bits per step: a second copy of the 1.5 times larger key only where a kernel reads it -- N = 2048, l <= 2)
            // (... and N = 1024 at GLWE dimension 2: the whole-workgroup latency shape of fbs_blind_rotate_k2.hip)
            if (ctx->group == 1 || (LOGN == 11 && ctx->p.l_bsk <= 2) || (LOGN == 10 && ctx->p.k == 2)) {
                if (e == hipSuccess && !ctx->d_bsk_hat_small) e = hipMalloc(&ctx->d_bsk_hat_small, polys * N * 8);
                if (e == hipSuccess) {
                    hipLaunchKernelGGL((k_bsk_transform<LOGN, LLS>), dim3(grid), dim3(1 << LLS), 0, ctx->stream, d_src,
                                       reinterpret_cast<double *>(ctx->d_bsk_hat_small), reinterpret_cast<const double *>(ctx->d_tw_fwd),
                                       n_inv, polys);
                    e = hipGetLastError();
                }
            }
        }
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_src);
    if (e != hipSuccess) return set_error(ctx, FBS_E_DEVICE, std::string("bootstrapping-key transform: ") + hipGetErrorString(e));
    return FBS_OK;
}

int dev_upload_keys(fbs_ctx *ctx) {
    const fbs_params &p = ctx->p;
    const uint32_t N = ctx->N;
    std::vector<uint64_t> fwd, inv;
    host_twiddles(p.log_n_poly, fwd, inv);
    std::vector<double> fwd_c(3 * (size_t)N), inv_c(3 * (size_t)N);   // the table, its two half-size and four quarter-size subtrees
    for (uint32_t i = 0; i < 3 * N; i++) {
        fwd_c[i] = fq_centered(fwd[i]);
        inv_c[i] = fq_centered(inv[i]);
    }
    const size_t bsk_words = ctx->n_ggsw * ctx->rows * (p.k + 1) * N;
    const size_t ksk_rows = (size_t)ctx->D * p.t_ksk;
    if (!ctx->d_tw_fwd) {
        FBS_HIP(ctx, hipMalloc(&ctx->d_tw_fwd, 3 * (size_t)N * 8));
        FBS_HIP(ctx, hipMalloc(&ctx->d_tw_inv, 3 * (size_t)N * 8));
        FBS_HIP(ctx, hipMalloc(&ctx->d_bsk_hat, bsk_words * 8));
        FBS_HIP(ctx, hipMalloc(&ctx->d_ksk, ksk_rows * ctx->ksk_stride * 8));
        FBS_HIP(ctx, hipMalloc(&ctx->d_ksk_f, ksk_rows * ctx->ksk_stride * 8));
        FBS_HIP(ctx, hipMalloc(&ctx->d_ks_corr, (size_t)ctx->ksk_stride * 8));
    }
    // (B/2) * sum of all key rows: what turns the unsigned bit fields of the key-switch kernels into balanced digits
    std::vector<uint64_t> corr(ctx->ksk_stride, 0);
    {
        std::vector<unsigned __int128> sum(p.n + 1, 0);
        for (size_t r = 0; r < ksk_rows; r++)
            for (uint32_t i = 0; i <= p.n; i++) sum[i] += ctx->ksk[r * (p.n + 1) + i];
        for (uint32_t i = 0; i <= p.n; i++) corr[i] = fq_mul((uint64_t)(sum[i] % FQ), 1ull << (p.gamma_ksk - 1));
    }
    FBS_HIP(ctx, hipMemcpyAsync(ctx->d_ks_corr, corr.data(), corr.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    FBS_HIP(ctx, hipMemcpyAsync(ctx->d_tw_fwd, fwd_c.data(), fwd_c.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    FBS_HIP(ctx, hipMemcpyAsync(ctx->d_tw_inv, inv_c.data(), inv_c.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    FBS_HIP(ctx, hipMemsetAsync(ctx->d_ksk, 0, ksk_rows * ctx->ksk_stride * 8, ctx->stream));
    FBS_HIP(ctx, hipMemcpy2DAsync(ctx->d_ksk, (size_t)ctx->ksk_stride * 8, ctx->ksk.data(), (size_t)(p.n + 1) * 8,
                                  (size_t)(p.n + 1) * 8, ksk_rows, hipMemcpyHostToDevice, ctx->stream));
    if (ctx->group == 2) {   // psi^x, x < N: what the pointwise factors zeta^e - 1 of the monomials X^e - 1 are made from
        std::vector<double> tbl(N);
        const uint64_t psi = fq_pow(FQ_GENERATOR, (FQ - 1) / (2ull * N));
        uint64_t pw = 1;
        for (uint32_t x = 0; x < N; x++) {
            tbl[x] = fq_centered(pw);
            pw = fq_mul(pw, psi);
        }
        if (!ctx->d_psi_pow) FBS_HIP(ctx, hipMalloc(&ctx->d_psi_pow, tbl.size() * 8));
        FBS_HIP(ctx, hipMemcpy(ctx->d_psi_pow, tbl.data(), tbl.size() * 8, hipMemcpyHostToDevice));
    }
    {   // centred doubles for the FP64 key-switch kernel, padded like the integer copy
        std::vector<double> kf(ksk_rows * (size_t)ctx->ksk_stride, 0.0);
        for (size_t r = 0; r < ksk_rows; r++)
            for (uint32_t i = 0; i <= p.n; i++) kf[r * ctx->ksk_stride + i] = fq_centered(ctx->ksk[r * (p.n + 1) + i]);
        FBS_HIP(ctx, hipMemcpy(ctx->d_ksk_f, kf.data(), kf.size() * 8, hipMemcpyHostToDevice));
    }
    FBS_HIP(ctx, hipStreamSynchronize(ctx->stream));   // fwd_c / inv_c are about to go out of scope
    if (int rc = dev_keyswitch_gemm_setup(ctx)) return rc;
    switch (p.log_n_poly) {
#define X(L) case L: return upload_keys_t<L>(ctx);
        FBS_FOR_EACH_SHAPE(X)
#undef X
    }
    return set_error(ctx, FBS_E_INVALID, "unsupported N");
}

// One workgroup per CU -- ALL the waves a CU holds in one barrier domain.  The two waves that share a SIMD then advance
// in lockstep; as separate workgroups the SIMD's oldest-first arbitration lets one of them run ahead (measured per
// workgroup with the wall clock, P1024: 6.0 ms for the favoured ones, 10.7 ms for the others, every XCD alike), and once
// the favoured half has left, the rest runs with one wave per SIMD and nothing to cover its stalls: 10.7 ms per 1024-batch
// against 10.0 ms in lockstep.  It pays when the launch fills whole rounds (a round = what the chip holds at once: 600
// bootstraps take 9.1 ms as small workgroups, 10.05 ms as whole-CU ones); beyond a few rounds the hardware refills freed
// slots anyway.  Measured and NOT adopted for the other shapes: two-level sets at N = 1024 (slower with the priority hand-over below:
// 152.5 against 155.5 k FBS/s at p = 2, 124 against 131 k at p = 4), N = 2048 with
// two bootstraps per workgroup (pairs 10.13 against 9.94 ms, l = 2 23.1 against 21.7 ms; with the priority hand-over as
// well: 10.34 against 9.87 ms at 1024 bootstraps, 82.4 against 73.7 ms at 8192 -- the transforms' own barriers then span
// eight waves).
// How many bootstraps of a launch of `count` go to whole-CU workgroups: all of them when the last round is (nearly) full,
// else the whole rounds only -- the rest follows as a launch of its own in the shape that suits its size (a partly
// filled round is faster as small workgroups: 768 bootstraps 8.2 ms against 9.3).
static size_t whole_cu_share(const fbs_ctx *ctx, size_t count, size_t per_round) {
    if (!ctx->tune.br_whole_cu) return 0;   // (A/B switch)
    const size_t r = count % per_round;
    return (r == 0 || 8 * r >= 7 * per_round) ? count : count - r;
}

int dev_blind_rotate(fbs_ctx *ctx, const fbs_tvset *tv, const GateView &gv, const uint32_t *d_ms, hipStream_t stream) {
    const fbs_params &p = ctx->p;
    BrArgs a{};
    a.gv = gv;
    a.ms = d_ms;
    a.bsk_hat = reinterpret_cast<const double *>(ctx->d_bsk_hat);
    a.tw_fwd = reinterpret_cast<const double *>(ctx->d_tw_fwd);
    a.tw_inv = reinterpret_cast<const double *>(ctx->d_tw_inv);
    a.tvs = tv->d_tvs;
    a.post = tv->d_post;
    a.n = p.n;
    a.l = p.l_bsk;
    a.beta = p.beta_bsk;
    a.ct_words = ctx->D + 1;
    // (entry n_tables of the set is TV_0: selectable only by the rotations of a fused program, whose ids the host wrote)
    a.n_tables = (gv.acc_rows || gv.row_words) ? tv->n_tables + 1 : std::max(1u, tv->n_tables);
    const size_t count = gv.count;
    if (count == 0) return FBS_OK;
    if (count > 0x7FFFFFFFull) return set_error(ctx, FBS_E_INVALID, "batch too large for one launch");
    a.count = count;
    // Two bootstraps per workgroup exactly where two-wave workgroups would double up on half of the SIMDs: between one
    // and two bootstraps per CU (measured per 1024-coefficient launch: 6.5 ms against 9.8).  Up to one per CU the
    // two-wave form is faster (5.7 against 6.5 ms), beyond two per CU too (9.8-11.1 against 11.1).
    const bool pair = lanes_log2_for((int)p.log_n_poly) == 6 && count > (size_t)ctx->cu_count && count <= 2 * (size_t)ctx->cu_count;
    dim3 grid((unsigned)(pair ? (count + 1) / 2 : count));
    if (p.k >= 2 && !(p.k == 2 && p.log_n_poly == 10 && ctx->group == 2 && p.l_bsk == 1)) {
        // every (k >= 2, N, l, key bits per step) but the one with kernels of its own: k + 1 waves per bootstrap (fbs_blind_rotate_glwe.hip).
        // Bootstraps per workgroup by launch size: one up to one bootstrap per CU, two up to two, the throughput shape beyond; a launch
        // longer than a round of the throughput shape whose last round would be at most two bootstraps per CU: whole rounds first, the
        // rest as a launch of its own (k = 3, N = 512: 1 024 = 768 + 256 bootstraps in 4.1 + 1.8 ms against two rounds' 7.4)
        a.psi_pow = reinterpret_cast<const double *>(ctx->d_psi_pow);
        const size_t cus = (size_t)ctx->cu_count;
        const int full = glwe_full_fpw(p.log_n_poly, p.k);
        const size_t per_round = (size_t)full * cus, rest_n = per_round ? count % per_round : 0;
        const bool small_ok = ctx->tune.br_cu_max_per_cu >= 1 && ctx->tune.br_glwe_fpw == 0;
        const bool cut = small_ok && ctx->tune.br_whole_cu && count > per_round && rest_n != 0 && rest_n <= (full > 2 ? 2 : 1) * cus;
        if (cut) {
            a.count = count - rest_n;
            a.gv.count = a.count;
        }
        int fpw = (int)ctx->tune.br_glwe_fpw;
        if (fpw == 0) fpw = !small_ok ? full : a.count <= cus ? 1 : (a.count <= 2 * cus && full > 2) ? 2 : full;
        hipEvent_t c0, c1;
        prof_begin(ctx, 1, stream, &c0, &c1);
        if (!launch_blind_rotate_glwe(ctx, a, fpw, stream, &ctx->prof.kernel[1])) {
            if (c0) ctx->prof.pool.push_back({c0, c1});
            return set_error(ctx, FBS_E_INVALID, "no blind-rotation kernel for this GLWE dimension and polynomial size");
        }
        prof_end(ctx, 1, stream, c0, c1);
        FBS_HIP(ctx, hipGetLastError());
        if (!cut) return FBS_OK;
        GateView rest = gv;
        rest.f_begin += count - rest_n;
        rest.count = rest_n;
        if (rest.out_rows) rest.out_rows += (count - rest_n) * (size_t)(rest.row_words ? rest.row_words : ctx->D + 1);
        return dev_blind_rotate(ctx, tv, rest, d_ms, stream);
    }
    if (p.k == 2) {   // GLWE dimension 2 at N = 1024, two key bits per step, one level: the shape the selector picks for p <= 15
        a.psi_pow = reinterpret_cast<const double *>(ctx->d_psi_pow);
        // a launch longer than a round of four-bootstrap workgroups whose last round would be (far) from full: whole rounds first,
        // the rest as a launch of its own in the twelve-wave shape (1 124 = 1 024 + 100: 7.3 + 2.1 ms against two rounds' 14.5)
        const size_t per_round = 4 * (size_t)ctx->cu_count, rest_n = count % per_round;
        const bool cut = count > per_round && rest_n != 0 && rest_n <= K2_CU_ROUNDS * (size_t)ctx->cu_count && ctx->tune.br_whole_cu &&
                         ctx->tune.br_k2_shape == 0 && ctx->tune.br_cu_kernel && ctx->tune.br_cu_max_per_cu >= 1;
        if (cut) {
            a.count = count - rest_n;
            a.gv.count = a.count;
        }
        hipEvent_t c0, c1;
        prof_begin(ctx, 1, stream, &c0, &c1);
        if (!launch_blind_rotate_k2(ctx, a, stream, &ctx->prof.kernel[1])) {
            if (c0) ctx->prof.pool.push_back({c0, c1});   // (the event pair goes back: nothing was recorded between them)
            return set_error(ctx, FBS_E_INVALID, "no blind-rotation kernel for this k = 2 shape");
        }
        prof_end(ctx, 1, stream, c0, c1);
        FBS_HIP(ctx, hipGetLastError());
        if (!cut) return FBS_OK;
        GateView rest = gv;
        rest.f_begin += count - rest_n;
        rest.count = rest_n;
        if (rest.out_rows) rest.out_rows += (count - rest_n) * (size_t)(rest.row_words ? rest.row_words : ctx->D + 1);
        return dev_blind_rotate(ctx, tv, rest, d_ms, stream);
    }
    if (ctx->group == 2) {
        a.psi_pow = reinterpret_cast<const double *>(ctx->d_psi_pow);
        // launches of at most one bootstrap per CU: the whole-CU shape (2.65-2.9 ms per bootstrap against 3.3-3.4; two rounds of
        // it are no faster than two bootstraps side by side in the four-wave kernel: 5.44 against 5.35 ms per 512)
        // Two gadget levels (the 128-bit sets for p = 31): the whole-CU shape for every launch, round after round -- the
        // two-waves-per-polynomial kernel spills 100 registers there (22.7 ms per 1024 bootstraps against 17.1)
        if ((count <= (size_t)ctx->cu_count || p.l_bsk == 2) && ctx->tune.br_cu_max_per_cu >= 1) {
            hipEvent_t c0, c1;
            prof_begin(ctx, 1, stream, &c0, &c1);
            if (launch_blind_rotate_cu_pairs(ctx, a, stream, &ctx->prof.kernel[1])) {
                prof_end(ctx, 1, stream, c0, c1);
                FBS_HIP(ctx, hipGetLastError());
                return FBS_OK;
            }
            if (c0) ctx->prof.pool.push_back({c0, c1});
        }
        const int dig2 = p.l_bsk == 1 ? 4 : p.beta_bsk <= 7 ? 3 : 0;
        hipEvent_t e0, e1;
        prof_begin(ctx, 1, stream, &e0, &e1);
#define LAUNCH_PAIRS(L, DIG)                                                                                          \
    do {                                                                                                               \
        ctx->prof.kernel[1] = "k_blind_rotate_pairs<" #L "," + std::to_string(lanes_log2_for(L)) + "," #DIG ">";        \
        hipLaunchKernelGGL((k_blind_rotate_pairs<L, lanes_log2_for(L), DIG>), dim3((unsigned)count),                   \
                           dim3(2 << lanes_log2_for(L)), 0, stream, a);                                                \
    } while (0)
#define PAIRS_FOR(L)                                                                                                   \
    case L:                                                                                                            \
        if (dig2 == 4) LAUNCH_PAIRS(L, 4);                                                                             \
        else if (dig2 == 3) LAUNCH_PAIRS(L, 3);                                                                        \
        else LAUNCH_PAIRS(L, 0);                                                                                       \
        break;
        switch (p.log_n_poly) {
            PAIRS_FOR(10)
            PAIRS_FOR(11)
            PAIRS_FOR(12)
            default: return set_error(ctx, FBS_E_INVALID, "two key bits per step: N = 1024, 2048 or 4096 only");
        }
#undef PAIRS_FOR
#undef LAUNCH_PAIRS
        prof_end(ctx, 1, stream, e0, e1);
        FBS_HIP(ctx, hipGetLastError());
        return FBS_OK;
    }
    const int by_beta = p.beta_bsk <= 7 ? 3 : p.beta_bsk <= 9 ? 2 : 1;
    const int dig = p.l_bsk > 5 ? 0 : p.l_bsk == 1 ? 4 : p.l_bsk == 2 ? 4 + by_beta : by_beta;
    // at most one bootstrap per CU: the shape with twice the waves per bootstrap, where there is one (fbs_ntt.hpp)
    const bool small_launch = ctx->d_bsk_hat_small != nullptr && count <= (size_t)ctx->cu_count;
    hipEvent_t e0, e1;
    prof_begin(ctx, 1, stream, &e0, &e1);
    {
        // launches that leave most of the chip empty: one bootstrap on the eight waves of a CU (fbs_blind_rotate_cu.hip)
        // Up to TWO bootstraps per CU: the second round of workgroups follows the first CU by CU (512 bootstraps: 5.9 ms against
        // 6.5 ms for two bootstraps side by side in the two-waves-per-bootstrap kernel; 384: 6.0 against 6.5).  Beyond that
        // the small workgroups of k_blind_rotate win (768: 8.3 ms against three rounds of 2.95).
        if (ctx->d_bsk_hat_small && count <= (size_t)ctx->cu_count * (size_t)ctx->tune.br_cu_max_per_cu &&
            launch_blind_rotate_cu(ctx, a, stream, &ctx->prof.kernel[1])) {
            prof_end(ctx, 1, stream, e0, e1);
            FBS_HIP(ctx, hipGetLastError());
            return FBS_OK;
        }
    }
    const size_t whole = (p.log_n_poly == 10 && dig == 3) ? whole_cu_share(ctx, count, 4 * (size_t)ctx->cu_count) : 0;
    if (whole) {
        // the benchmark shape: four bootstraps = the eight waves of a CU in one workgroup
        a.count = whole;
        a.gv.count = whole;
        ctx->prof.kernel[1] = "k_blind_rotate<10,6,3,4>";
        hipLaunchKernelGGL((k_blind_rotate<10, 6, 3, 4>), dim3((unsigned)((whole + 3) / 4)), dim3((2 << 6) * 4), 0, stream, a);
        prof_end(ctx, 1, stream, e0, e1);
        FBS_HIP(ctx, hipGetLastError());
        if (whole == count) return FBS_OK;
        GateView rest = gv;                       // what did not fill a round: its own launch, in the shape its size asks for
        rest.f_begin += whole;
        rest.count = count - whole;
        if (rest.out_rows) rest.out_rows += whole * (size_t)(rest.row_words ? rest.row_words : ctx->D + 1);
        return dev_blind_rotate(ctx, tv, rest, d_ms, stream);
    } else if (p.log_n_poly == 10 && (dig == 6 || dig == 7) && count > 8 * (size_t)ctx->cu_count) {
        // the two-level 128-bit sets at N = 1024 in launches of more than two rounds: no taking turns (see TURNS)
        if (dig == 6) {
            ctx->prof.kernel[1] = "k_blind_rotate<10,6,6,1,false>";
            hipLaunchKernelGGL((k_blind_rotate<10, 6, 6, 1, false>), grid, dim3(2 << 6), 0, stream, a);
        } else {
            ctx->prof.kernel[1] = "k_blind_rotate<10,6,7,1,false>";
            hipLaunchKernelGGL((k_blind_rotate<10, 6, 7, 1, false>), grid, dim3(2 << 6), 0, stream, a);
        }
    } else
    switch (p.log_n_poly) {
#define LAUNCH_LL(L, LL_, DIG, FPW)                                                                                    \
    do {                                                                                                               \
        ctx->prof.kernel[1] = "k_blind_rotate<" #L "," + std::to_string(LL_) + "," #DIG "," #FPW ">";                   \
        hipLaunchKernelGGL((k_blind_rotate<L, LL_, DIG, FPW>), grid, dim3((2 << (LL_)) * FPW), 0, stream, a);          \
    } while (0)
#define LAUNCH_DIG(L, LL_, FPW)                                                                                        \
    do {                                                                                                               \
        if (dig == 4) LAUNCH_LL(L, LL_, 4, FPW);                                                                       \
        else if (dig == 5) LAUNCH_LL(L, LL_, 5, FPW);                                                                  \
        else if (dig == 6) LAUNCH_LL(L, LL_, 6, FPW);                                                                  \
        else if (dig == 7) LAUNCH_LL(L, LL_, 7, FPW);                                                                  \
        else if (dig == 3) LAUNCH_LL(L, LL_, 3, FPW);                                                                  \
        else if (dig == 2) LAUNCH_LL(L, LL_, 2, FPW);                                                                  \
        else if (dig == 1) LAUNCH_LL(L, LL_, 1, FPW);                                                                  \
        else LAUNCH_LL(L, LL_, 0, FPW);                                                                                \
    } while (0)
#define X(L)                                                                                                           \
    case L:                                                                                                            \
        if constexpr (lanes_log2_for_small_launch(L) != lanes_log2_for(L)) {                                           \
            if (small_launch) {                                                                                        \
                a.bsk_hat = reinterpret_cast<const double *>(ctx->d_bsk_hat_small);                                    \
                LAUNCH_DIG(L, lanes_log2_for_small_launch(L), 1);                                                      \
                break;                                                                                                 \
            }                                                                                                          \
        }                                                                                                              \
        if constexpr (lanes_log2_for(L) == 6) {                                                                        \
            if (pair) LAUNCH_DIG(L, lanes_log2_for(L), 2);                                                             \
            else LAUNCH_DIG(L, lanes_log2_for(L), 1);                                                                  \
        } else {                                                                                                       \
            LAUNCH_DIG(L, lanes_log2_for(L), 1);                                                                       \
        }                                                                                                              \
        break;
        FBS_FOR_EACH_SHAPE(X)
#undef X
#undef LAUNCH_DIG
#undef LAUNCH_LL
        default: return set_error(ctx, FBS_E_INVALID, "unsupported N");
    }
    prof_end(ctx, 1, stream, e0, e1);
    FBS_HIP(ctx, hipGetLastError());
    return FBS_OK;
}

// Every blind-rotation instantiation dev_blind_rotate can pick, by the rules of the dispatch above (fbs_kernel_catalog;
// tests/test_gpu_dispatch.py drives each one and checks it against the oracle)
void blind_rotate_catalog(std::vector<std::string> *out) {
    auto name = [](int L, int ll, int dig, int fpw) {
        return "k_blind_rotate<" + std::to_string(L) + "," + std::to_string(ll) + "," + std::to_string(dig) + "," + std::to_string(fpw) + ">";
    };
    for (int L : {8, 9, 10, 11, 12}) {
        const int ll = lanes_log2_for(L), small = lanes_log2_for_small_launch(L);
        for (int dig = 0; dig < 8; dig++) {
            out->push_back(name(L, ll, dig, 1));
            if (ll == 6) out->push_back(name(L, ll, dig, 2));
            if (small != ll) out->push_back(name(L, small, dig, 1));
        }
    }
    out->push_back("k_blind_rotate<10,6,3,4>");
    out->push_back("k_blind_rotate<10,6,6,1,false>");
    out->push_back("k_blind_rotate<10,6,7,1,false>");
    for (int L : {10, 11, 12})
        for (int dig : {0, 3, 4})
            out->push_back("k_blind_rotate_pairs<" + std::to_string(L) + "," + std::to_string(lanes_log2_for(L)) + "," + std::to_string(dig) + ">");
    blind_rotate_cu_catalog(out);
    blind_rotate_k2_catalog(out);
    blind_rotate_glwe_catalog(out);
}

int dev_polymul(fbs_ctx *ctx, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_c, hipStream_t stream) {
    const double n_inv = fq_centered(fq_inv(ctx->N));
    const double *twf = reinterpret_cast<const double *>(ctx->d_tw_fwd), *twi = reinterpret_cast<const double *>(ctx->d_tw_inv);
    switch (ctx->p.log_n_poly) {
#define X(L)                                                                                                               \
    case L:                                                                                                                \
        hipLaunchKernelGGL((k_polymul<L, lanes_log2_for(L)>), dim3(1), dim3(1 << lanes_log2_for(L)), 0, stream, d_a, d_b, \
                           d_c, twf, twi, n_inv);                                                                          \
        break;
        FBS_FOR_EACH_SHAPE(X)
#undef X
        default: return set_error(ctx, FBS_E_INVALID, "unsupported N");
    }
    FBS_HIP(ctx, hipGetLastError());
    return FBS_OK;
}

}  // namespace fbs

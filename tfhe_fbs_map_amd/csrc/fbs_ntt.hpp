// Negacyclic NTT mod q (fbs_field.hpp) of one polynomial spread over LANES = 2^LL lanes (one or more wavefronts),
// gfx950, exact FP64 arithmetic on integer-valued doubles.
//
// Each lane holds E = N/LANES coefficients in VGPRs.  A transform is a chain of "groups": log2(E) radix-2
// butterfly stages on registers, then an exchange through LDS that brings the next group of index bits into the
// lane.  With LANES = 64 the exchange is private to a wave (no barrier); with more lanes the waves of a polynomial
// meet at one s_barrier per exchange.  Exchanges ping-pong between two N-word buffers, so one barrier per
// exchange is enough: a buffer is rewritten only after every wave has passed the barrier that follows its last
// read of it.
//
// Forward = Cooley-Tukey with the twist folded into the twiddles (tw[i] = psi^bitrev(i)), natural order in,
// bit-reversed evaluation order out.  Lazy ranges: inputs |x| <= q, every stage adds a product below 0.75 q, so
// after LOGN <= 11 stages |x| < 9.3 q < 2^50 -- the bound fp_mulmod needs -- without a single range fix-up.
// Inverse = Gentleman-Sande with tw[i] = psi^-bitrev(i); sums double per stage, so each group starts by centring
// its registers (|x| <= q/2) and may run at most 4 stages (8 q before the last product).  The 1/N is folded into
// the bootstrapping key.  All twiddles are stored centred, as doubles.
#pragma once
#include <hip/hip_runtime.h>

#include "fbs_field.hpp"

namespace fbs {

// One twiddle table, two ways in: `lane` for per-lane gathers (global memory, or a copy in LDS), `uniform` for the
// stages whose twiddle index is the same in every lane of the polynomial -- read through the scalar cache into
// SGPRs (constant address space: the table is never written while a kernel runs), costing no VGPR and no LDS slot.
typedef const double __attribute__((address_space(4))) *uniform_doubles;
struct Twiddles {
    const double *lane;
    uniform_doubles uniform;
    __device__ __forceinline__ Twiddles(const double *lane_table, const double *global_table)
        : lane(lane_table), uniform((uniform_doubles)(uintptr_t)global_table) {}
};

// polynomials spread over up to 2^FBS_ONE_BUFFER_MAX_LL lanes exchange through ONE buffer (a second synchronisation
// per exchange instead of a ping-pong pair), which leaves room for twiddle tables in LDS
#ifndef FBS_ONE_BUFFER_MAX_LL
#define FBS_ONE_BUFFER_MAX_LL 8
#endif

// First Cooley-Tukey butterfly of a forward transform, (a, b) <- (a + w b, a - w b), by what is known about the inputs:
//   FIRST = 0  nothing (|a|, |b| <= q): the general product, 8 instructions;
//   FIRST = 1  |b| <= 2^8 (gadget digits, beta <= 9): b * w is exact in a double, 6 instructions;
//   FIRST = 2  |a|, |b| <= 2^6 (beta <= 7): a +- w b is an integer below 2^51 + 2^6, EXACT in one FMA each and left
//              UNREDUCED -- 2 instructions.  Every later butterfly reduces only the operand it multiplies, so all values of
//              the transform then sit near 2^51 and grow by less than 0.8 q a stage: below 2^51.2 after eleven stages,
//              inside the 2^52 that fp_mulmod (butterflies and key products alike) accepts, and every sum exact.
template <int FIRST>
__device__ __forceinline__ void first_butterfly(double &a, double &b, double w) {
    if constexpr (FIRST == 2) {
        const double u = a, d = b;
        a = __builtin_fma(d, w, u);
        b = __builtin_fma(-d, w, u);
    } else {
        const double u = a;
        const double v = FIRST == 1 ? fp_mulmod_exact(b, w) : fp_mulmod(b, w);
        a = u + v;
        b = u - v;
    }
}

template <int LOGN, int LL>
struct PolyNtt {
    static constexpr int N = 1 << LOGN;
    static constexpr int LANES = 1 << LL;
    static constexpr int E = N / LANES;
    static constexpr int LOGE = LOGN - LL;
    static constexpr int GROUPS = (LOGN + LOGE - 1) / LOGE;
    static_assert(LL >= 6 && LOGE >= 2 && LOGE <= 4 && LOGN <= 11, "unsupported shape (range analysis assumes <= 4 stages per group)");

    // lowest index bit held inside the lane during group g
    __host__ __device__ static constexpr int lo_of(int g) { return (LOGN - (g + 1) * LOGE) > 0 ? (LOGN - (g + 1) * LOGE) : 0; }

    // LDS word of coefficient j: index bits 4.. folded (XOR) into the bank-selecting bits 0..4
    __device__ static __forceinline__ uint32_t phys(uint32_t j) { return j ^ ((j >> 4) & 31u); }

    // coefficient index of register m of lane t during group G
    template <int G>
    __device__ static __forceinline__ uint32_t index_of(uint32_t t, int m) {
        constexpr int lo = lo_of(G);
        return ((t >> lo) << (lo + LOGE)) | ((uint32_t)m << lo) | (t & ((1u << lo) - 1u));
    }

    template <int G>
    __device__ static __forceinline__ void store_group(double *buf, uint32_t t, const double (&x)[E]) {
        const uint32_t base = phys(index_of<G>(t, 0));
#pragma unroll
        for (int m = 0; m < E; m++) buf[base ^ phys(index_of<G>(0, m))] = x[m];   // phys is XOR-linear
    }
    template <int G>
    __device__ static __forceinline__ void load_group(const double *buf, uint32_t t, double (&x)[E]) {
        const uint32_t base = phys(index_of<G>(t, 0));
#pragma unroll
        for (int m = 0; m < E; m++) x[m] = buf[base ^ phys(index_of<G>(0, m))];
    }

    // make LDS writes of the polynomial's lanes visible to its lanes
    __device__ static __forceinline__ void sync() {
        if constexpr (LL == 6) {
            // one wave: LDS operations complete in issue order; only the compiler must be held back, and only for
            // LDS ("local") accesses -- global loads (twiddles, key rows) stay free to be issued early
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
        } else {
            __syncthreads();
        }
    }

    // Exchange state: `bufs` = the N-word buffer(s) of this polynomial, `pp` = which of two is next.  A polynomial
    // that lives in ONE wave needs no ping-pong (its LDS operations are ordered): stride = 0 keeps a single buffer.
    struct Xchg {
        double *bufs;
        uint32_t pp;
        uint32_t stride = N;
        __device__ __forceinline__ double *next() {
            double *b = bufs + (pp ? stride : 0);
            pp ^= 1u;
            return b;
        }
    };

    // distinct twiddles of stage s for this lane: one per block of registers the stage pairs up, j = m >> (bit + 1)
    // R: node of the twiddle tree the transform hangs from (1 = a whole polynomial; W + w when it is part w of a polynomial W
    // times the size, WavesNtt in fbs_ntt_split.hpp).  Only the wave-uniform reads index the big tree with it; `tw.lane`
    // always points at a table of this transform's own tree.
    template <int G>
    __device__ static __forceinline__ void load_stage(int s, uint32_t t, const Twiddles &tw, double (&w)[E / 2], uint32_t R = 1) {
        constexpr int lo = lo_of(G);
        const int bit = LOGN - 1 - s - lo;     // register-index bit paired by this stage
        const int sh = lo + LOGE - LOGN + s;   // how far the lane's high part reaches into the block id
        const uint32_t hi_part = t >> lo;
#pragma unroll
        for (int j = 0; j < E / 2; j++) {
            if (j >= (E >> (bit + 1))) continue;
            // lo >= LL: the lane's high part is empty, the index depends on the register only
            w[j] = lo >= LL ? tw.uniform[(R << s) + (uint32_t)j] : tw.lane[(1u << s) + ((hi_part << sh) | (uint32_t)j)];
        }
    }

    // Cooley-Tukey stages of group G on registers.  The twiddles of a stage are requested before the butterflies of
    // the stage before it, so their (LDS or scalar-cache) latency hides behind arithmetic.
    // FIRST describes the inputs of the transform and with them the very first stage (see `first_butterfly`)
    template <int G, int FIRST>
    __device__ static __forceinline__ void fwd_group(double (&x)[E], uint32_t t, const Twiddles &tw, uint32_t R = 1) {
        constexpr int lo = lo_of(G);
        constexpr int s_begin = G * LOGE;
        constexpr int s_end = (G + 1) * LOGE < LOGN ? (G + 1) * LOGE : LOGN;
        double w[LOGE + 1][E / 2];
        load_stage<G>(s_begin, t, tw, w[0], R);
#pragma unroll
        for (int s = s_begin; s < s_end; s++) {
            const int bit = LOGN - 1 - s - lo;   // register-index bit paired by this stage
            const int hm = 1 << bit;
            if (s + 1 < s_end) load_stage<G>(s + 1, t, tw, w[s + 1 - s_begin], R);
#pragma unroll
            for (int m = 0; m < E; m++) {
                if (m & hm) continue;
                const double wv = w[s - s_begin][m >> (bit + 1)];
                if (s == 0) {
                    first_butterfly<FIRST>(x[m], x[m + hm], wv);
                    continue;
                }
                const double u = x[m];
                const double v = fp_mulmod(x[m + hm], wv);
                x[m] = u + v;
                x[m + hm] = u - v;
            }
        }
    }
    // Gentleman-Sande stages of group G, last stage first
    template <int G>
    __device__ static __forceinline__ void inv_group(double (&x)[E], uint32_t t, const Twiddles &tw, uint32_t R = 1) {
        constexpr int lo = lo_of(G);
        constexpr int s_begin = G * LOGE;
        constexpr int s_end = (G + 1) * LOGE < LOGN ? (G + 1) * LOGE : LOGN;
        double w[LOGE + 1][E / 2];
        load_stage<G>(s_end - 1, t, tw, w[s_end - 1 - s_begin], R);
#pragma unroll
        for (int m = 0; m < E; m++) x[m] = fp_center(x[m]);
#pragma unroll
        for (int s = s_end - 1; s >= s_begin; s--) {
            const int bit = LOGN - 1 - s - lo;
            const int hm = 1 << bit;
            if (s > s_begin) load_stage<G>(s - 1, t, tw, w[s - 1 - s_begin], R);
#pragma unroll
            for (int m = 0; m < E; m++) {
                if (m & hm) continue;
                const double u = x[m], v = x[m + hm];
                x[m] = u + v;
                x[m + hm] = fp_mulmod(u - v, w[s - s_begin][m >> (bit + 1)]);
            }
        }
    }

    struct NoHook {
        __device__ __forceinline__ void operator()() const {}
    };
    // `before_last` runs right before the butterflies of the last group: the place to issue global loads whose
    // results are wanted when the transform ends (one group of butterflies ~ one L2 round trip)
    template <int G, int FIRST, class Hook>
    __device__ static __forceinline__ void fwd_from(double (&x)[E], Xchg &xc, uint32_t t, const Twiddles &tw, Hook &&before_last,
                                                    uint32_t R = 1) {
        if constexpr (G + 1 == GROUPS) before_last();
        fwd_group<G, FIRST>(x, t, tw, R);
        if constexpr (G + 1 < GROUPS) {
            double *buf = xc.next();
            if constexpr (LL <= FBS_ONE_BUFFER_MAX_LL) sync();   // one buffer: the stores stay behind the reads that filled x
            store_group<G>(buf, t, x);
            sync();
            load_group<G + 1>(buf, t, x);
            fwd_from<G + 1, FIRST>(x, xc, t, tw, before_last, R);
        }
    }
    template <int G>
    __device__ static __forceinline__ void inv_from(double (&x)[E], Xchg &xc, uint32_t t, const Twiddles &tw, uint32_t R = 1) {
        inv_group<G>(x, t, tw, R);
        if constexpr (G > 0) {
            double *buf = xc.next();
            if constexpr (LL <= FBS_ONE_BUFFER_MAX_LL) sync();   // one buffer: the stores stay behind the reads that filled x
            store_group<G>(buf, t, x);
            sync();
            load_group<G - 1>(buf, t, x);
            inv_from<G - 1>(x, xc, t, tw, R);
        }
    }

    // coefficients (group-0 layout: register m of lane t = coefficient t + LANES*m, |x| <= q) -> evaluations
    // (last-group layout, |x| < 9.3 q)
    __device__ static __forceinline__ void forward(double (&x)[E], Xchg &xc, uint32_t t, const Twiddles &tw) {
        fwd_from<0, 0>(x, xc, t, tw, NoHook{});
    }
    // the same with a promise about the inputs (FIRST, see first_butterfly)
    template <int FIRST, class Hook>
    __device__ static __forceinline__ void forward(double (&x)[E], Xchg &xc, uint32_t t, const Twiddles &tw, Hook &&before_last,
                                                   uint32_t R = 1) {
        fwd_from<0, FIRST>(x, xc, t, tw, before_last, R);
    }
    // evaluations (last-group layout, |x| < 2^52) -> N * coefficients (group-0 layout, |x| <= 8 q).  BOUNDED (a promise
    // of |x| <= 8 q) is accepted for interface parity with SplitNtt and not used: four-stage groups need the centring.
    // interface parity with SplitNtt (which takes its wave-uniform inverse twiddles from the caller): nothing to carry
    struct InvUniform {};
    __device__ static __forceinline__ InvUniform inverse_uniform(uint32_t, const Twiddles &, uint32_t = 1) { return {}; }
    template <bool BOUNDED = false>
    __device__ static __forceinline__ void inverse(double (&x)[E], Xchg &xc, uint32_t t, const Twiddles &tw, const InvUniform &,
                                                   uint32_t R = 1) {
        inv_from<GROUPS - 1>(x, xc, t, tw, R);
    }
    template <bool BOUNDED = false>
    __device__ static __forceinline__ void inverse(double (&x)[E], Xchg &xc, uint32_t t, const Twiddles &tw) {
        inv_from<GROUPS - 1>(x, xc, t, tw);
    }

    static constexpr int LANE_TABLE_OFFSET = 0;   // per-lane twiddle gathers read the uploaded table from its start
    static constexpr bool HAS_EVAL_POSITION = false;
    static constexpr int EVAL_GROUP_LOG2 = 0;   // which evaluation point a register holds is not published for this shape
    __device__ static __forceinline__ uint32_t eval_position_lane(uint32_t) { return 0; }
    __host__ __device__ static constexpr uint32_t eval_position_reg(int) { return 0; }
    // word (of the partner's exchange buffer) where thread t parks register m of a polynomial handed over between components
    __device__ static __forceinline__ uint32_t handoff_word(uint32_t t, int m) { return (uint32_t)m * LANES + t; }

    // Key storage: the evaluation held in register m of lane t after forward() sits at word
    // ((m/2)*LANES + t)*2 + (m&1) of its polynomial, so the lanes read a polynomial with E/2 fully
    // coalesced 16-byte loads.
    __host__ __device__ static constexpr uint32_t key_word(uint32_t t, int m) {
        return (((uint32_t)(m >> 1) * LANES + t) << 1) | (uint32_t)(m & 1);
    }
};

// lanes per polynomial for each supported size (measured on MI355X, 1024-bootstrap batches): 16 coefficients per lane
// wherever the polynomial has them.
//   N <= 1024: one wave per polynomial: exchanges are wave-private, no barrier inside a transform
//              (N = 1024: 14.3 ms against 17.7 ms for two waves with E = 8, at the time both were measured);
//   N  = 2048: two waves per polynomial, E = 16, two exchanges per transform through one buffer (two workgroup barriers
//              each) with the twiddle tables in the LDS a ping-pong pair would have taken: 23.7 ms; with the ping-pong
//              pair and twiddles from global memory 27.5; four waves with E = 8 and three exchanges 39.7 (l = 4).
// A second shape for launches that leave most of the chip empty (at most one bootstrap per CU): N = 1024 as FOUR waves
// per polynomial with 4 coefficients per lane (WavesNtt<10, 2>, fbs_ntt_split.hpp) -- one bootstrap is the eight waves a
// CU holds of this kernel.  (Round 2: two waves per polynomial with 8 coefficients per lane, the generic PolyNtt<10, 7>
// with two barriers per exchange: 4.8 ms per bootstrap.)  It needs its own transformed copy of the bootstrapping key (the
// evaluation order differs).  N = 2048 likewise: four waves per polynomial with 8 coefficients per lane (WavesNtt<11, 2>) beside
// the main shape's two waves with 16.  Same as the main shape where there is no such alternative.
#ifndef FBS_SMALL_LAUNCH_LL_1024
#define FBS_SMALL_LAUNCH_LL_1024 8
#endif
__host__ __device__ constexpr int lanes_log2_for_small_launch(int log_n) {
    return log_n == 10 ? FBS_SMALL_LAUNCH_LL_1024 : log_n == 11 ? 8 : (log_n <= 10 ? 6 : log_n - 4);
}

#ifdef FBS_COEFS_PER_LANE_LOG2   // experiments: force 2^k coefficients per lane everywhere it is possible
__host__ __device__ constexpr int lanes_log2_for(int log_n) {
    return log_n - FBS_COEFS_PER_LANE_LOG2 < 6 ? 6 : log_n - FBS_COEFS_PER_LANE_LOG2;
}
#else
__host__ __device__ constexpr int lanes_log2_for(int log_n) { return log_n <= 10 ? 6 : log_n - 4; }
#endif

}  // namespace fbs

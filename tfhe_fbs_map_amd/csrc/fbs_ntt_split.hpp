// Negacyclic NTT of one polynomial held by ONE wavefront with 16 coefficients per lane, organised so that the wave
// hides its own LDS latency.
//
// After the first Cooley-Tukey stage (registers m and m+8) the polynomial falls apart into two independent
// half-size transforms: registers 0..7 ("half 0") and registers 8..15 ("half 1").  Each half then runs the schedule of
// fbs_ntt.hpp with 8 coefficients per lane -- groups of 3 butterfly stages in registers, an exchange through LDS
// between groups -- but the two halves are issued OUT OF PHASE: while the reads of one half's exchange are in flight
// the other half computes.  Same arithmetic, same twiddle table (the half rooted at node 2+h of the twiddle tree:
// stage s' of half h uses tw[((2+h) << s') + block]), one more exchange per half (3 instead of 2, each half as wide),
// no extra registers: a half's registers hold either the values it is about to store or the ones it just asked for.
//
// Interface and conventions are those of PolyNtt (fbs_ntt.hpp): forward = lazy CT, natural order in (register m of
// lane t = coefficient t + 64 m), inverse = GS with centring per group, 1/N folded into the key, evaluation order
// defined by "whatever forward() leaves in register m of lane t" -- the key transform uses the same code.
#pragma once
#include <type_traits>

#include "fbs_ntt.hpp"
#include "fbs_ntt_lane.hpp"

namespace fbs {

template <int LOGN, int LL>
struct SplitNtt {
    static constexpr int N = 1 << LOGN;
    static constexpr int LANES = 1 << LL;
    static constexpr int E = N / LANES;
    static constexpr int EH = E / 2;           // registers per half
    static constexpr int LOGM = LOGN - 1;      // a half is a transform of size N/2 ...
    static constexpr int LOGEH = 3;            // ... with 3 index bits in registers
    static constexpr int GROUPS = LOGM / LOGEH;
    static_assert(LL == 6 && E == 16 && LOGM % LOGEH == 0, "one wave, 16 coefficients per lane");
    using Base = PolyNtt<LOGN, LL>;
    using Xchg = typename Base::Xchg;

    __device__ static __forceinline__ void sync() { Base::sync(); }
    __host__ __device__ static constexpr uint32_t key_word(uint32_t t, int m) { return Base::key_word(t, m); }
    static constexpr int LANE_TABLE_OFFSET = 0;
    // Where the evaluation held in register m of lane t after forward() sits in the output ARRAY of the textbook in-place
    // Cooley-Tukey transform (position P holds the value at psi^(2 bitrev(P) + 1)): half h = m / 8 is array half h, and
    // the last group's layout is local index 8 t + r.  Lane and register contribute disjoint bits.
    static constexpr bool HAS_EVAL_POSITION = true;
    // the exponents 2 bitrev(P) + 1 of the points a wave holds in one register are 2^EVAL_GROUP_LOG2 * (a permutation of the
    // lanes) + a wave-uniform remainder
    static constexpr int EVAL_GROUP_LOG2 = 2;
    __device__ static __forceinline__ uint32_t eval_position_lane(uint32_t t) { return t << 3; }
    __host__ __device__ static constexpr uint32_t eval_position_reg(int m) { return (uint32_t)(m >> 3) * (N / 2) + (uint32_t)(m & 7); }
    __device__ static __forceinline__ uint32_t handoff_word(uint32_t t, int m) { return (uint32_t)m * LANES + t; }
    // input / output layout of the coefficient domain: register m of lane t = coefficient t + LANES * m
    template <int G>
    __device__ static __forceinline__ uint32_t index_of(uint32_t t, int m) {
        static_assert(G == 0, "only the coefficient-domain layout is public");
        return t + (uint32_t)LANES * (uint32_t)m;
    }

    // ---- one half: local index i (LOGM bits), local stage s' = parent stage - 1 --------------------------------
    __host__ __device__ static constexpr int lo_of(int g) { return LOGM - (g + 1) * LOGEH; }
    // LDS word of local index i.  gfx950 banks 8-byte reads per 32 lanes over 32 word positions and 8-byte writes per
    // 16 lanes over 16; the three layouts walk the lanes through index bits {0..5}, {0..2, 6..8} and {3..8}.  XOR-ing
    // bits 4..6 into 0..2, bit 6 into 3 and bit 7 into 4 makes every one of those walks a bijection on the bank bits
    // (checked exhaustively in tests/test_lds_swizzle.py): no bank conflicts in any exchange.
    __device__ static __forceinline__ uint32_t phys(uint32_t i) { return i ^ ((i >> 4) & 7u) ^ (((i >> 6) & 3u) << 3); }
    template <int G>
    __device__ static __forceinline__ uint32_t local_index(uint32_t t, int r) {
        constexpr int lo = lo_of(G);
        return ((t >> lo) << (lo + LOGEH)) | ((uint32_t)r << lo) | (t & ((1u << lo) - 1u));
    }
    template <int G, int OFF>
    __device__ static __forceinline__ void store_half(double *buf, uint32_t t, const double (&x)[E]) {
        const uint32_t base = phys(local_index<G>(t, 0));
#pragma unroll
        for (int r = 0; r < EH; r++) buf[base ^ phys(local_index<G>(0, r))] = x[OFF + r];   // phys is XOR-linear
    }
    template <int G, int OFF>
    __device__ static __forceinline__ void load_half(const double *buf, uint32_t t, double (&x)[E]) {
        const uint32_t base = phys(local_index<G>(t, 0));
#pragma unroll
        for (int r = 0; r < EH; r++) x[OFF + r] = buf[base ^ phys(local_index<G>(0, r))];
    }
    // All twiddles of group G of half H for this lane: stage k of the group pairs blocks of 2^(LOGEH-k) registers and
    // needs one twiddle per block.  They are requested by the CALLER, right behind the exchange that leads into the
    // group and ahead of the other half's work: LDS answers a wave in order, so a request issued later would make
    // its consumer wait for everything the other half has in flight.
    struct GroupTw {
        double w[LOGEH][EH / 2];
    };
    // R: node of the twiddle tree this transform hangs from -- 1 for a whole polynomial, 2 + h when it is half h of a
    // polynomial twice the size (PairNtt below); wave-uniform.  Only the scalar (wave-uniform) reads index the big tree
    // with it: `tw.lane` always points at a table of THIS transform's own tree (node 1 = its root), so that per-lane
    // gather addresses keep compile-time offsets -- with a run-time root every (group, stage, half) costs a VGPR for
    // its own address, and the blind rotation has none to spare.
    template <int G, int H>
    __device__ static __forceinline__ void load_twiddles(uint32_t t, const Twiddles &tw, GroupTw &g, uint32_t R = 1) {
        constexpr int lo = lo_of(G);
        const uint32_t hi_part = t >> lo;
#pragma unroll
        for (int k = 0; k < LOGEH; k++) {
            const int s = G * LOGEH + k;
            const int bit = LOGM - 1 - s - lo;
            const int sh = lo + LOGEH - LOGM + s;
            const uint32_t root = (uint32_t)(2 + H) << s, uroot = (2u * R + (uint32_t)H) << s;
#pragma unroll
            for (int j = 0; j < EH / 2; j++) {
                if (j >= (EH >> (bit + 1))) continue;
                // lo >= LL: no lane bit reaches the block index -- the twiddle is wave-uniform (scalar cache, SGPRs)
                g.w[k][j] = lo >= LL ? tw.uniform[uroot + (uint32_t)j] : tw.lane[root + ((hi_part << sh) | (uint32_t)j)];
            }
        }
    }
    template <int G, int H>
    __device__ static __forceinline__ void fwd_group(double (&x)[E], const GroupTw &g) {
        constexpr int lo = lo_of(G), OFF = H * EH;
#pragma unroll
        for (int k = 0; k < LOGEH; k++) {
            const int bit = LOGM - 1 - (G * LOGEH + k) - lo, hm = 1 << bit;
#pragma unroll
            for (int r = 0; r < EH; r++) {
                if (r & hm) continue;
                const double u = x[OFF + r];
                const double v = fp_mulmod(x[OFF + r + hm], g.w[k][r >> (bit + 1)]);
                x[OFF + r] = u + v;
                x[OFF + r + hm] = u - v;
            }
        }
    }
    // CENTRE = false: the caller promises |x| <= 8 q on entry.  Three Gentleman-Sande stages then feed fp_mulmod at most
    // 8 * 8 q = 2^52 * (q / 2^46) < 2^52, inside the bound under which it stays exact (fbs_field.hpp; its result is then
    // below 0.8 q rather than 0.75 q), and leave sums below 64 q < 2^52, which the next group's centring accepts.
    template <int G, int H, bool CENTRE = true>
    __device__ static __forceinline__ void inv_group(double (&x)[E], const GroupTw &g) {
        constexpr int lo = lo_of(G), OFF = H * EH;
        if constexpr (CENTRE) {
#pragma unroll
            for (int r = 0; r < EH; r++) x[OFF + r] = fp_center(x[OFF + r]);
        }
#pragma unroll
        for (int k = LOGEH - 1; k >= 0; k--) {
            const int bit = LOGM - 1 - (G * LOGEH + k) - lo, hm = 1 << bit;
#pragma unroll
            for (int r = 0; r < EH; r++) {
                if (r & hm) continue;
                const double u = x[OFF + r], v = x[OFF + r + hm];
                x[OFF + r] = u + v;
                x[OFF + r + hm] = fp_mulmod(u - v, g.w[k][r >> (bit + 1)]);
            }
        }
    }
    // registers -> LDS -> registers of the next group's layout; the reads are only ISSUED here
    template <int FROM, int TO, int H>
    __device__ static __forceinline__ void exchange(double (&x)[E], double *region, uint32_t t) {
        sync();   // the stores stay behind every earlier read of this region
        store_half<FROM, H * EH>(region, t, x);
        sync();
        load_half<TO, H * EH>(region, t, x);
    }

    // The requests just issued must leave NOW: left to itself the scheduler sinks them below the other half's
    // arithmetic (shorter live ranges), which is exactly the latency this schedule exists to hide.
    __device__ static __forceinline__ void pin() { __builtin_amdgcn_sched_barrier(0); }

    struct NoHook {
        __device__ __forceinline__ void operator()() const {}
    };

    // FIRST: what is known about the inputs (first_butterfly, fbs_ntt.hpp)
    template <int FIRST, class Hook>
    __device__ static __forceinline__ void forward(double (&x)[E], Xchg &xc, uint32_t t, const Twiddles &tw, Hook &&before_last,
                                                   uint32_t R = 1) {
        double *half0 = xc.next(), *half1 = half0 + N / 2;
        {
            const double w0 = tw.uniform[R];
#pragma unroll
            for (int r = 0; r < EH; r++) first_butterfly<FIRST>(x[r], x[r + EH], w0);
        }
        static_assert(GROUPS == 3, "written out for three groups per half");
        GroupTw ta, tb;
        load_twiddles<0, 0>(t, tw, ta, R);
        load_twiddles<0, 1>(t, tw, tb, R);
        fwd_group<0, 0>(x, ta);
        exchange<0, 1, 0>(x, half0, t);
        load_twiddles<1, 0>(t, tw, ta, R);
        pin();
        fwd_group<0, 1>(x, tb);
        exchange<0, 1, 1>(x, half1, t);
        load_twiddles<1, 1>(t, tw, tb, R);
        pin();
        fwd_group<1, 0>(x, ta);
        exchange<1, 2, 0>(x, half0, t);
        load_twiddles<2, 0>(t, tw, ta, R);
        pin();
        fwd_group<1, 1>(x, tb);
        exchange<1, 2, 1>(x, half1, t);
        load_twiddles<2, 1>(t, tw, tb, R);
        pin();
        before_last();
        fwd_group<2, 0>(x, ta);
        fwd_group<2, 1>(x, tb);
    }
    __device__ static __forceinline__ void forward(double (&x)[E], Xchg &xc, uint32_t t, const Twiddles &tw) {
        forward<0>(x, xc, t, tw, NoHook{});
    }
    // evaluations (|x| < 2^52; BOUNDED: |x| <= 8 q, which spares the first centring pass) -> N * coefficients (|x| <= 8 q)
    // The wave-uniform twiddles of the inverse (last group of each half + the joining stage).  Scalar loads share the
    // LDS counter and return out of order, so whoever waits for one waits for everything in flight: a caller with slack
    // ahead of the transform (a barrier, say) requests them there with inverse_uniform() and passes them in.
    struct InvUniform {
        GroupTw a, b;
        double w0;
    };
    __device__ static __forceinline__ InvUniform inverse_uniform(uint32_t t, const Twiddles &tw, uint32_t R = 1) {
        InvUniform u;
        load_twiddles<0, 0>(t, tw, u.a, R);
        load_twiddles<0, 1>(t, tw, u.b, R);
        u.w0 = tw.uniform[R];
        return u;
    }
    template <bool BOUNDED = false>
    __device__ static __forceinline__ void inverse(double (&x)[E], Xchg &xc, uint32_t t, const Twiddles &tw) {
        inverse<BOUNDED>(x, xc, t, tw, inverse_uniform(t, tw));
    }
    template <bool BOUNDED = false>
    __device__ static __forceinline__ void inverse(double (&x)[E], Xchg &xc, uint32_t t, const Twiddles &tw, const InvUniform &uni,
                                                   uint32_t R = 1) {
        double *half0 = xc.next(), *half1 = half0 + N / 2;
        GroupTw ta, tb;
        load_twiddles<2, 0>(t, tw, ta, R);
        load_twiddles<2, 1>(t, tw, tb, R);
        const GroupTw &t0a = uni.a, &t0b = uni.b;
        const double w0 = uni.w0;
        inv_group<2, 0, !BOUNDED>(x, ta);
        exchange<2, 1, 0>(x, half0, t);
        load_twiddles<1, 0>(t, tw, ta, R);
        pin();
        inv_group<2, 1, !BOUNDED>(x, tb);
        exchange<2, 1, 1>(x, half1, t);
        load_twiddles<1, 1>(t, tw, tb, R);
        pin();
        inv_group<1, 0>(x, ta);
        exchange<1, 0, 0>(x, half0, t);
        pin();
        inv_group<1, 1>(x, tb);
        exchange<1, 0, 1>(x, half1, t);
        pin();
        inv_group<0, 0>(x, t0a);
        inv_group<0, 1>(x, t0b);
        // last Gentleman-Sande stage joins the halves: inputs <= 4 q each, outputs <= 8 q and < 0.75 q
#pragma unroll
        for (int r = 0; r < EH; r++) {
            const double u = x[r], v = x[r + EH];
            x[r] = u + v;
            x[r + EH] = fp_mulmod(u - v, w0);
        }
    }
};

// Negacyclic NTT of one polynomial of size N = W M held by W = 2^LOGW wavefronts (M = 1024: N = 2048 on two waves,
// N = 4096 on four), as LOGW butterfly stages across the waves plus one wave-private SplitNtt of size M per wave.
//
// The first LOGW Cooley-Tukey stages pair registers of one thread (natural layout: register m of thread t = coefficient
// t + 64 W m; stage s pairs m with m + 8 >> s) and split the polynomial into W independent size-M transforms hanging
// from nodes W .. 2W - 1 of the twiddle tree.  Wave w takes sub-transform w: ONE trip through LDS re-deals the values
// (every thread writes its 16 / W entries of each part, wave w reads part w in SplitNtt's layout -- register m' of lane
// l = entry l + 64 m'), two workgroup barriers; everything after that is wave-private (no barrier, the split schedule
// of SplitNtt hides its own LDS latency).  The generic two-wave transform (PolyNtt<11,7>) needs four barriers per
// transform, spilled 35-54 registers inside the blind rotation and conflicted in LDS; this one has the register budget
// of the N = 1024 kernel.  The inverse is the mirror image.  `bufs` of the exchange state is the polynomial's N-word LDS
// region; wave w uses words [w M, (w+1) M) of it as its private SplitNtt buffer -- the very words only it reads in the
// re-deal, so no barrier is needed between the re-deal and the private transform.
// Twiddles: `uniform` is the table of the whole polynomial (N entries, tw[i] = psi^bitrev(i)); `lane` points at the W
// parts' OWN tables back to back, M entries each: entry i of part w = entry ((W + w) << d) + (i - 2^d) of the big table,
// d = floor(log2 i) (host_twiddles appends them to the big table: LANE_TABLE_OFFSET).
template <int LOGN, int LOGW>
struct WavesNtt {
    static constexpr int N = 1 << LOGN;
    static constexpr int W = 1 << LOGW;
    static constexpr int LL = 6 + LOGW;
    static constexpr int LANES = 64 * W;
    static constexpr int E = N / LANES;
    static constexpr int LOGE = LOGN - LL;
    static constexpr int EP = E / W;          // registers per part after the cross stages
    static constexpr int M = N / W;
    // the wave-private transform of a part: the split schedule at 16 coefficients per lane; the plain grouped schedule at 4 or
    // 8 (N = 1024 or 2048 on four waves: one bootstrap on the eight waves of a CU, the shape of launches that leave most of
    // the chip empty)
    // (N = 1024 on four waves: 256-point parts at 4 coefficients per lane, most of whose data movement is between registers and
    // lanes -- LaneNtt256, fbs_ntt_lane.hpp)
    using Half = typename std::conditional<E == 16, SplitNtt<(E == 16 ? LOGN - LOGW : 10), 6>,
                                           typename std::conditional<E == 4 && M == 256, LaneNtt256,
                                                                     typename std::conditional<E == 8 && M == 512, LaneNtt512, PolyNtt<LOGN - LOGW, 6>>::type>::type>::type;
    static_assert((E == 16 || E == 8 || E == 4) && Half::E == E && E >= W && (LOGW == 1 || LOGW == 2), "two or four waves, 4, 8 or 16 coefficients per lane");
    // where the per-lane tables start inside the uploaded twiddle buffer: [N whole][N two halves][N four quarters]
    static constexpr int LANE_TABLE_OFFSET = LOGW * N;
    // array position of the evaluation in register m of thread t (see SplitNtt): wave w holds array part w
    static constexpr bool HAS_EVAL_POSITION = E == 16;   // (published for the split schedule only)
    static constexpr int EVAL_GROUP_LOG2 = 2 + LOGW;
    __device__ static __forceinline__ uint32_t eval_position_lane(uint32_t t) { return (t >> 6) * (uint32_t)M + ((t & 63u) << 3); }
    __host__ __device__ static constexpr uint32_t eval_position_reg(int m) { return (uint32_t)(m >> 3) * (M / 2) + (uint32_t)(m & 7); }
    __device__ static __forceinline__ Twiddles half_table(const Twiddles &tw, uint32_t w) {
        Twiddles sub = tw;
        sub.lane = tw.lane + w * (uint32_t)M;
        return sub;
    }

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
    __device__ static __forceinline__ void sync() { __syncthreads(); }
    __host__ __device__ static constexpr uint32_t key_word(uint32_t t, int m) {
        return (((uint32_t)(m >> 1) * LANES + t) << 1) | (uint32_t)(m & 1);
    }
    template <int G>
    __device__ static __forceinline__ uint32_t index_of(uint32_t t, int m) {
        static_assert(G == 0, "only the coefficient-domain layout is public");
        return t + (uint32_t)LANES * (uint32_t)m;
    }
    // Where thread t parks register m of a polynomial handed over between the two components: inside the words its own wave
    // uses privately afterwards, so that no other wave has to be waited for before the inverse transform starts.
    __device__ static __forceinline__ uint32_t handoff_word(uint32_t t, int m) { return (t >> 6) * (uint32_t)M + (t & 63u) + 64u * (uint32_t)m; }
    // which part this wave owns (wave-uniform, and known to the compiler as such)
    __device__ static __forceinline__ uint32_t wave_of(uint32_t t) { return __builtin_amdgcn_readfirstlane(t >> 6); }

    struct NoHook {
        __device__ __forceinline__ void operator()() const {}
    };
    template <int FIRST, class Hook>
    __device__ static __forceinline__ void forward(double (&x)[E], Xchg &xc, uint32_t t, const Twiddles &tw, Hook &&before_last) {
        const uint32_t w = wave_of(t), l = t & 63u;
        // cross stages: stage s pairs register m with m + (E/2 >> s) inside blocks of E >> s registers; block b uses tw[2^s + b]
#pragma unroll
        for (int s = 0; s < LOGW; s++) {
            const int half = (E / 2) >> s;
#pragma unroll
            for (int m = 0; m < E; m++) {
                if (m & half) continue;
                const double wv = tw.uniform[(1u << s) + (uint32_t)(m >> (LOGE - s))];
                if (s == 0) {
                    first_butterfly<FIRST>(x[m], x[m + half], wv);
                } else {
                    const double u = x[m], v = fp_mulmod(x[m + half], wv);
                    x[m] = u + v;
                    x[m + half] = u - v;
                }
            }
        }
        double *region = xc.bufs;
        __syncthreads();   // whoever used the region before (the other waves' private transforms, the caller) is done
#pragma unroll
        for (int q = 0; q < W; q++)
#pragma unroll
            for (int r = 0; r < EP; r++) region[q * M + t + (uint32_t)LANES * r] = x[q * EP + r];
        __syncthreads();
        double *mine = region + w * M;
#pragma unroll
        for (int m = 0; m < E; m++) x[m] = mine[l + 64u * m];
        typename Half::Xchg hx{mine, 0, 0};
        Half::template forward<0>(x, hx, l, half_table(tw, w), before_last, (uint32_t)W + w);
    }
    __device__ static __forceinline__ void forward(double (&x)[E], Xchg &xc, uint32_t t, const Twiddles &tw) {
        forward<0>(x, xc, t, tw, NoHook{});
    }

    struct InvUniform {
        typename Half::InvUniform half;
        double w[W - 1];   // the cross stages' twiddles, tw[1 .. W - 1]
    };
    __device__ static __forceinline__ InvUniform inverse_uniform(uint32_t t, const Twiddles &tw) {
        InvUniform u;
        u.half = Half::inverse_uniform(t & 63u, tw, (uint32_t)W + wave_of(t));   // wave-uniform reads only: the big table
#pragma unroll
        for (int i = 0; i < W - 1; i++) u.w[i] = tw.uniform[1 + i];
        return u;
    }
    template <bool BOUNDED = false>
    __device__ static __forceinline__ void inverse(double (&x)[E], Xchg &xc, uint32_t t, const Twiddles &tw) {
        inverse<BOUNDED>(x, xc, t, tw, inverse_uniform(t, tw));
    }
    // evaluations (|x| < 2^52; BOUNDED: |x| <= 8 q) -> N * coefficients, |x| <= 8 W q
    template <bool BOUNDED = false>
    __device__ static __forceinline__ void inverse(double (&x)[E], Xchg &xc, uint32_t t, const Twiddles &tw, const InvUniform &uni) {
        const uint32_t w = wave_of(t), l = t & 63u;
        double *region = xc.bufs, *mine = region + w * M;
        typename Half::Xchg hx{mine, 0, 0};
        Half::template inverse<BOUNDED>(x, hx, l, half_table(tw, w), uni.half, (uint32_t)W + w);   // <= 8 q, the part's coefficient layout
        Half::sync();      // the stores below stay behind the last reads of the private transform (same wave, same words)
#pragma unroll
        for (int m = 0; m < E; m++) mine[l + 64u * m] = x[m];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < W; q++)
#pragma unroll
            for (int r = 0; r < EP; r++) x[q * EP + r] = region[q * M + t + (uint32_t)LANES * r];
        // Gentleman-Sande stages join the parts, last cross stage first: inputs <= 8 q, sums double per stage (<= 8 W q <=
        // 32 q, inside what fp_mulmod and the caller's centring accept), products < 0.8 q
#pragma unroll
        for (int s = LOGW - 1; s >= 0; s--) {
            const int half = (E / 2) >> s;
#pragma unroll
            for (int m = 0; m < E; m++) {
                if (m & half) continue;
                const double u = x[m], v = x[m + half];
                x[m] = u + v;
                x[m + half] = fp_mulmod(u - v, uni.w[(1 << s) - 1 + (m >> (LOGE - s))]);
            }
        }
    }
};
template <int LOGN>
using PairNtt = WavesNtt<LOGN, 1>;

// the transform used for a shape: the split schedule where a wave holds a whole polynomial at 16 coefficients per lane,
// two or four wave-private split transforms behind one or two cross stages where two or four waves hold a larger one
#ifndef FBS_PAIR_NTT
#define FBS_PAIR_NTT 1   // experiments: 0 falls back to the generic two-wave transform for N = 2048
#endif
template <int LOGN, int LL, int KIND = (LL == 6 && LOGN - LL == 4) ? 1 : (FBS_PAIR_NTT && LL == 7 && LOGN == 11) ? 2 : (LL == 8 && (LOGN == 12 || LOGN == 11 || LOGN == 10)) ? 3 : 0>
struct NttFor {
    using type = PolyNtt<LOGN, LL>;
};
template <int LOGN, int LL>
struct NttFor<LOGN, LL, 1> {
    using type = SplitNtt<LOGN, LL>;
};
template <int LOGN, int LL>
struct NttFor<LOGN, LL, 2> {
    using type = WavesNtt<LOGN, 1>;
};
template <int LOGN, int LL>
struct NttFor<LOGN, LL, 3> {
    using type = WavesNtt<LOGN, 2>;
};

}  // namespace fbs

// Negacyclic sub-transform of 256 points held by ONE wavefront with 4 coefficients per lane, with most of its data movement
// done between REGISTERS AND LANES instead of through LDS.  gfx950 only (v_permlane32_swap_b32, v_permlane16_swap_b32).
//
// It is the wave-private part of a 1024-point transform dealt over four waves (WavesNtt<10, 2>, fbs_ntt_split.hpp: the shape
// of one bootstrap on the eight waves of a CU).  Eight butterfly stages on an 8-bit index j = (j7 .. j0); a lane holds the
// four values that differ in TWO index bits (the "register bits"), the other six select the lane.  A butterfly stage needs
// its bit in the registers.  The grouped schedule of PolyNtt<8, 6> changes the register bits by writing everything to LDS and
// reading it back in another order, three times per transform -- at 4 coefficients per lane that is more LDS traffic than
// arithmetic (measured in the one-bootstrap-per-CU kernel: LDS busy 49 % of the time, 37 % of that bank conflicts, and a
// dependent LDS round trip every 32 butterflies).  Here:
//   * v_permlane32_swap a, b  exchanges lanes 32..63 of a with lanes 0..31 of b: afterwards the pair (a, b) of every lane
//     holds what differed in LANE bit 5 before, and lane bit 5 says which of the two registers a value came from -- a
//     transposition of one register bit with lane bit 5, one instruction per 32-bit half of a register pair;
//     v_permlane16_swap does the same with lane bit 4;
//   * so: two stages on the register bits, transpose both register bits with lane bits 5 and 4, two more stages -- four stages
//     without touching memory; ONE exchange through LDS (conflict-free swizzle, checked in tests/test_lds_swizzle.py) deals the
//     four remaining bits into the registers and lane bits 5, 4; four more stages the same way.
// One LDS round trip per transform instead of three, and every per-lane twiddle is the same in every transform a lane ever
// runs, so a kernel loads them once (`Tw`, 9 doubles per direction) instead of gathering them from a table in LDS.
//
// Index bookkeeping (forward; the inverse is the mirror image).  m = register, ln = lane.
//   layout A  m = (j7 j6), ln = (j5 .. j0)                      stages 0, 1 (bits 7, 6; twiddles wave-uniform)
//   swap32 on (x0,x2), (x1,x3); swap16 on (x0,x1), (x2,x3)
//   layout B  m = (j5 j4), ln = (j7 j6 j3 j2 j1 j0)             stages 2, 3 (bits 5, 4)
//   LDS
//   layout C  m = (j3 j2), ln = (j1 j0 j7 j6 j5 j4)             stages 4, 5 (bits 3, 2)
//   swap32, swap16
//   layout D  m = (j1 j0), ln = (j3 j2 j7 j6 j5 j4)             stages 6, 7 (bits 1, 0)
// Stage s pairs the values that differ in bit 7 - s and multiplies by node 2^s + (j >> (8 - s)) of the part's twiddle tree.
// The evaluation order is "whatever forward() leaves in register m of lane ln"; the key transform runs the same code.
#pragma once
#include "fbs_ntt.hpp"

namespace fbs {

struct LaneNtt256 {
    static constexpr int N = 256, LANES = 64, E = 4, LL = 6;
    static constexpr int GROUPS = 2;

    // transposition of a register bit with lane bit 5 / lane bit 4 (see above)
    __device__ static __forceinline__ void swap32(double &a, double &b) {
        const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
        a = __hiloint2double((int)hi[0], (int)lo[0]);
        b = __hiloint2double((int)hi[1], (int)lo[1]);
    }
    __device__ static __forceinline__ void swap16(double &a, double &b) {
        const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
        a = __hiloint2double((int)hi[0], (int)lo[0]);
        b = __hiloint2double((int)hi[1], (int)lo[1]);
    }
    // both register bits <-> lane bits 5, 4 (its own inverse)
    __device__ static __forceinline__ void transpose(double (&x)[E]) {
        swap32(x[0], x[2]);
        swap32(x[1], x[3]);
        swap16(x[0], x[1]);
        swap16(x[2], x[3]);
    }

    // LDS word of index j: XOR-linear, conflict-free for the four access patterns of the exchange (layout B and C, written 16
    // lanes at a time, read 32 at a time)
    __host__ __device__ static constexpr uint32_t phys(uint32_t j) { return j ^ ((j >> 4) & 15u) ^ (((j >> 6) & 1u) << 4); }
    __device__ static __forceinline__ uint32_t index_b(uint32_t ln, int m) { return ((ln >> 4) << 6) | ((uint32_t)m << 4) | (ln & 15u); }
    __device__ static __forceinline__ uint32_t index_c(uint32_t ln, int m) { return ((ln & 15u) << 4) | ((uint32_t)m << 2) | (ln >> 4); }

    // the wave-private exchange is ordered by the LDS pipe itself; only the compiler must be held back
    __device__ static __forceinline__ void sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local");
    }

    // twiddles of one direction for one lane: stages 0, 1 wave-uniform (they end up in scalar registers), the rest per lane
    struct Tw {
        double u0, u1a, u1b;
        double t2, t3a, t3b, t4, t5a, t5b, t6, t7a, t7b;
    };
    // `part`: the part's own table (node i of ITS tree at entry i; global memory or a copy in LDS); `big`: the table of the whole
    // polynomial, read through the scalar cache at the part's root R
    __device__ static __forceinline__ Tw load(const double *part, uniform_doubles big, uint32_t R, uint32_t ln) {
        Tw w;
        w.u0 = big[R];
        w.u1a = big[2u * R];
        w.u1b = big[2u * R + 1u];
        const uint32_t top = ln >> 4, hi = ln & 15u, six = (hi << 2) | top;
        w.t2 = part[4u + top];
        w.t3a = part[8u + 2u * top];
        w.t3b = part[8u + 2u * top + 1u];
        w.t4 = part[16u + hi];
        w.t5a = part[32u + 2u * hi];
        w.t5b = part[32u + 2u * hi + 1u];
        w.t6 = part[64u + six];
        w.t7a = part[128u + 2u * six];
        w.t7b = part[128u + 2u * six + 1u];
        return w;
    }

    template <int FIRST = 0>
    __device__ static __forceinline__ void ct(double &a, double &b, double w) {
        if constexpr (FIRST != 0) {
            first_butterfly<FIRST>(a, b, w);
        } else {
            const double u = a, v = fp_mulmod(b, w);
            a = u + v;
            b = u - v;
        }
    }
    __device__ static __forceinline__ void gs(double &a, double &b, double w) {
        const double u = a, v = b;
        a = u + v;
        b = fp_mulmod(u - v, w);
    }

    struct NoHook {
        __device__ __forceinline__ void operator()() const {}
    };
    // NP polynomials side by side (same twiddles, one 256-word exchange buffer each): layout A in, layout D out.
    // FIRST: what is known about the inputs (first_butterfly); `before_last` runs ahead of the last four stages.
    template <int NP, int FIRST, class Hook>
    __device__ static __forceinline__ void forward_multi(double (&x)[NP][E], double *const (&bufs)[NP], uint32_t ln, const Tw &w,
                                                         Hook &&before_last) {
#pragma unroll
        for (int p = 0; p < NP; p++) {
            ct<FIRST>(x[p][0], x[p][2], w.u0);
            ct<FIRST>(x[p][1], x[p][3], w.u0);
        }
#pragma unroll
        for (int p = 0; p < NP; p++) {
            ct(x[p][0], x[p][1], w.u1a);
            ct(x[p][2], x[p][3], w.u1b);
        }
#pragma unroll
        for (int p = 0; p < NP; p++) transpose(x[p]);
#pragma unroll
        for (int p = 0; p < NP; p++) {
            ct(x[p][0], x[p][2], w.t2);
            ct(x[p][1], x[p][3], w.t2);
        }
#pragma unroll
        for (int p = 0; p < NP; p++) {
            ct(x[p][0], x[p][1], w.t3a);
            ct(x[p][2], x[p][3], w.t3b);
        }
        {
            const uint32_t wb = phys(index_b(ln, 0)), rc = phys(index_c(ln, 0));
            sync();   // the stores stay behind every earlier read of these buffers
#pragma unroll
            for (int p = 0; p < NP; p++)
#pragma unroll
                for (int m = 0; m < E; m++) bufs[p][wb ^ phys((uint32_t)m << 4)] = x[p][m];
            sync();
#pragma unroll
            for (int p = 0; p < NP; p++)
#pragma unroll
                for (int m = 0; m < E; m++) x[p][m] = bufs[p][rc ^ phys((uint32_t)m << 2)];
        }
        before_last();
#pragma unroll
        for (int p = 0; p < NP; p++) {
            ct(x[p][0], x[p][2], w.t4);
            ct(x[p][1], x[p][3], w.t4);
        }
#pragma unroll
        for (int p = 0; p < NP; p++) {
            ct(x[p][0], x[p][1], w.t5a);
            ct(x[p][2], x[p][3], w.t5b);
        }
#pragma unroll
        for (int p = 0; p < NP; p++) transpose(x[p]);
#pragma unroll
        for (int p = 0; p < NP; p++) {
            ct(x[p][0], x[p][2], w.t6);
            ct(x[p][1], x[p][3], w.t6);
        }
#pragma unroll
        for (int p = 0; p < NP; p++) {
            ct(x[p][0], x[p][1], w.t7a);
            ct(x[p][2], x[p][3], w.t7b);
        }
    }
    // evaluations (layout D, |x| < 2^52) -> 256 * coefficients of the part (layout A, |x| <= 8 q); `w` = load() of the INVERSE table
    // (`mid` runs half-way, between the two halves of the exchange)
    template <class Hook = NoHook>
    __device__ static __forceinline__ void inverse_one(double (&x)[E], double *buf, uint32_t ln, const Tw &w, Hook &&mid = NoHook{}) {
#pragma unroll
        for (int m = 0; m < E; m++) x[m] = fp_center(x[m]);
        gs(x[0], x[1], w.t7a);
        gs(x[2], x[3], w.t7b);
        gs(x[0], x[2], w.t6);
        gs(x[1], x[3], w.t6);
        transpose(x);
        gs(x[0], x[1], w.t5a);
        gs(x[2], x[3], w.t5b);
        gs(x[0], x[2], w.t4);
        gs(x[1], x[3], w.t4);   // <= 8 q, products < 0.75 q
        {
            const uint32_t wc = phys(index_c(ln, 0)), rb = phys(index_b(ln, 0));
            sync();
#pragma unroll
            for (int m = 0; m < E; m++) buf[wc ^ phys((uint32_t)m << 2)] = x[m];
            mid();
            sync();
#pragma unroll
            for (int m = 0; m < E; m++) x[m] = fp_center(buf[rb ^ phys((uint32_t)m << 4)]);
        }
        gs(x[0], x[1], w.t3a);
        gs(x[2], x[3], w.t3b);
        gs(x[0], x[2], w.t2);
        gs(x[1], x[3], w.t2);
        transpose(x);
        gs(x[0], x[1], w.u1a);
        gs(x[2], x[3], w.u1b);
        gs(x[0], x[2], w.u0);
        gs(x[1], x[3], w.u0);
    }

    // ---- the interface WavesNtt expects of the wave-private transform of a part (fbs_ntt_split.hpp) -------------------------
    struct Xchg {
        double *bufs;
        uint32_t pp;
        uint32_t stride = N;
    };
    template <int FIRST, class Hook>
    __device__ static __forceinline__ void forward(double (&x)[E], Xchg &xc, uint32_t ln, const Twiddles &tw, Hook &&before_last,
                                                   uint32_t R = 1) {
        const Tw w = load(tw.lane, tw.uniform, R, ln);
        double y[1][E];
#pragma unroll
        for (int m = 0; m < E; m++) y[0][m] = x[m];
        double *b[1] = {xc.bufs};
        forward_multi<1, FIRST>(y, b, ln, w, before_last);
#pragma unroll
        for (int m = 0; m < E; m++) x[m] = y[0][m];
    }
    struct InvUniform {};
    __device__ static __forceinline__ InvUniform inverse_uniform(uint32_t, const Twiddles &, uint32_t = 1) { return {}; }
    template <bool BOUNDED = false>
    __device__ static __forceinline__ void inverse(double (&x)[E], Xchg &xc, uint32_t ln, const Twiddles &tw, const InvUniform &,
                                                   uint32_t R = 1) {
        inverse_one(x, xc.bufs, ln, load(tw.lane, tw.uniform, R, ln));
    }
};

// The same for 512 points at 8 coefficients per lane: the wave-private part of a 2048-point transform dealt over four waves
// (WavesNtt<11, 2>).  Nine stages on j = (j8 .. j0), three register bits (r2 r1 r0):
//   layout A  m = (j8 j7 j6), ln = (j5 .. j0)                      stages 0, 1, 2 (twiddles wave-uniform)
//   swap32 r2 <-> lane 5, swap16 r1 <-> lane 4
//   layout B  m = (j5 j4 j6), ln = (j8 j7 j3 j2 j1 j0)             stages 3, 4 (bits 5, 4)
//   LDS
//   layout C  m = (j3 j2 j1), ln = (j0 j8 j7 j6 j5 j4)             stages 5, 6, 7 (bits 3, 2, 1)
//   swap32 r2 <-> lane 5
//   layout D  m = (j0 j2 j1), ln = (j3 j8 j7 j6 j5 j4)             stage 8 (bit 0)
// Seventeen per-lane twiddles per direction (two, four; one, two, four; four): a kernel keeps the forward ones in registers and
// reads the inverse ones from a [k][lane] table in LDS (`TwLane`, conflict-free by construction).
struct LaneNtt512 {
    static constexpr int N = 512, LANES = 64, E = 8, LL = 6;
    static constexpr int GROUPS = 2;
    static constexpr int LANE_TW = 17;

    __device__ static __forceinline__ void swap32(double &a, double &b) { LaneNtt256::swap32(a, b); }
    __device__ static __forceinline__ void swap16(double &a, double &b) { LaneNtt256::swap16(a, b); }
    __device__ static __forceinline__ void sync() { LaneNtt256::sync(); }
    __host__ __device__ static constexpr uint32_t phys(uint32_t j) {
        return j ^ ((j >> 5) & 15u) ^ (((j >> 4) & 1u) << 3) ^ (((j >> 7) & 1u) << 4);
    }
    __device__ static __forceinline__ uint32_t index_b(uint32_t ln, int m) {
        return ((ln >> 4) << 7) | ((uint32_t)(m & 1) << 6) | ((uint32_t)(m >> 2) << 5) | ((uint32_t)((m >> 1) & 1) << 4) | (ln & 15u);
    }
    __device__ static __forceinline__ uint32_t index_c(uint32_t ln, int m) { return ((ln & 31u) << 4) | ((uint32_t)m << 1) | (ln >> 5); }

    struct Uniform {
        double u0, u1[2], u2[4];
    };
    struct TwLane {
        double t3[2], t4[4], t5, t6[2], t7[4], t8[4];
    };
    struct Tw {
        Uniform u;
        TwLane t;
    };
    // node of the part's tree for every per-lane twiddle, in TwLane order (k = 0 .. 16)
    __device__ static __forceinline__ uint32_t lane_node(uint32_t ln, int k) {
        const uint32_t top = ln >> 4, hi = ln & 31u, l5 = ln >> 5;
        if (k < 2) return 8u + ((top << 1) | (uint32_t)k);                                  // t3[r0]
        if (k < 6) return 16u + ((top << 2) | (uint32_t)(k - 2));                           // t4[(r0 << 1) | r2]
        if (k < 7) return 32u + hi;                                                         // t5
        if (k < 9) return 64u + ((hi << 1) | (uint32_t)(k - 7));                            // t6[r2]
        if (k < 13) return 128u + ((hi << 2) | (uint32_t)(k - 9));                          // t7[(r2 << 1) | r1]
        return 256u + ((hi << 3) | (l5 << 2) | (uint32_t)(k - 13));                         // t8[(r1 << 1) | r0]
    }
    __device__ static __forceinline__ Uniform load_uniform(uniform_doubles big, uint32_t R) {
        Uniform u;
        u.u0 = big[R];
        u.u1[0] = big[2u * R];
        u.u1[1] = big[2u * R + 1u];
#pragma unroll
        for (int i = 0; i < 4; i++) u.u2[i] = big[4u * R + (uint32_t)i];
        return u;
    }
    // per-lane twiddles straight from the part's own table (global memory or LDS)
    __device__ static __forceinline__ TwLane load_lane(const double *part, uint32_t ln) {
        double v[LANE_TW];
#pragma unroll
        for (int k = 0; k < LANE_TW; k++) v[k] = part[lane_node(ln, k)];
        return unpack(v);
    }
    // ... or from a table laid out [k][lane] (what a kernel builds once in LDS: every read is 64 consecutive words)
    __device__ static __forceinline__ TwLane load_lane_table(const double *table, uint32_t ln) {
        double v[LANE_TW];
#pragma unroll
        for (int k = 0; k < LANE_TW; k++) v[k] = table[(uint32_t)k * 64u + ln];
        return unpack(v);
    }
    __device__ static __forceinline__ TwLane unpack(const double (&v)[LANE_TW]) {
        TwLane t;
        t.t3[0] = v[0], t.t3[1] = v[1];
#pragma unroll
        for (int i = 0; i < 4; i++) t.t4[i] = v[2 + i], t.t7[i] = v[9 + i], t.t8[i] = v[13 + i];
        t.t5 = v[6];
        t.t6[0] = v[7], t.t6[1] = v[8];
        return t;
    }

    template <int FIRST = 0>
    __device__ static __forceinline__ void ct(double &a, double &b, double w) { LaneNtt256::ct<FIRST>(a, b, w); }
    __device__ static __forceinline__ void gs(double &a, double &b, double w) { LaneNtt256::gs(a, b, w); }

    struct NoHook {
        __device__ __forceinline__ void operator()() const {}
    };
    template <int NP, int FIRST, class Hook>
    __device__ static __forceinline__ void forward_multi(double (&x)[NP][E], double *const (&bufs)[NP], uint32_t ln, const Uniform &u,
                                                         const TwLane &w, Hook &&before_last) {
#pragma unroll
        for (int p = 0; p < NP; p++)
#pragma unroll
            for (int m = 0; m < 4; m++) ct<FIRST>(x[p][m], x[p][m + 4], u.u0);                       // stage 0: bit 8 = r2
#pragma unroll
        for (int p = 0; p < NP; p++)
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int m = 0; m < 2; m++) ct(x[p][4 * h + m], x[p][4 * h + m + 2], u.u1[h]);      // stage 1: bit 7 = r1, block r2
#pragma unroll
        for (int p = 0; p < NP; p++)
#pragma unroll
            for (int q = 0; q < 4; q++) ct(x[p][2 * q], x[p][2 * q + 1], u.u2[q]);                    // stage 2: bit 6 = r0, block (r2 r1)
#pragma unroll
        for (int p = 0; p < NP; p++) {
#pragma unroll
            for (int m = 0; m < 4; m++) swap32(x[p][m], x[p][m + 4]);                                // r2 <-> lane 5
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int m = 0; m < 2; m++) swap16(x[p][4 * h + m], x[p][4 * h + m + 2]);            // r1 <-> lane 4
        }
#pragma unroll
        for (int p = 0; p < NP; p++)
#pragma unroll
            for (int m = 0; m < 4; m++) ct(x[p][m], x[p][m + 4], w.t3[m & 1]);                        // stage 3: bit 5 = r2, block (.., r0)
#pragma unroll
        for (int p = 0; p < NP; p++)
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int m = 0; m < 2; m++) ct(x[p][4 * h + m], x[p][4 * h + m + 2], w.t4[(m << 1) | h]);   // stage 4: bit 4 = r1, block (.., r0, r2)
        {
            const uint32_t wb = phys(index_b(ln, 0)), rc = phys(index_c(ln, 0));
            sync();
#pragma unroll
            for (int p = 0; p < NP; p++)
#pragma unroll
                for (int m = 0; m < E; m++) bufs[p][wb ^ phys(index_b(0, m))] = x[p][m];
            sync();
#pragma unroll
            for (int p = 0; p < NP; p++)
#pragma unroll
                for (int m = 0; m < E; m++) x[p][m] = bufs[p][rc ^ phys(index_c(0, m))];
        }
        before_last();
#pragma unroll
        for (int p = 0; p < NP; p++)
#pragma unroll
            for (int m = 0; m < 4; m++) ct(x[p][m], x[p][m + 4], w.t5);                               // stage 5: bit 3 = r2
#pragma unroll
        for (int p = 0; p < NP; p++)
#pragma unroll
            for (int h = 0; h < 2; h++)
#pragma unroll
                for (int m = 0; m < 2; m++) ct(x[p][4 * h + m], x[p][4 * h + m + 2], w.t6[h]);       // stage 6: bit 2 = r1, block (.., r2)
#pragma unroll
        for (int p = 0; p < NP; p++)
#pragma unroll
            for (int q = 0; q < 4; q++) ct(x[p][2 * q], x[p][2 * q + 1], w.t7[q]);                    // stage 7: bit 1 = r0, block (.., r2, r1)
#pragma unroll
        for (int p = 0; p < NP; p++)
#pragma unroll
            for (int m = 0; m < 4; m++) swap32(x[p][m], x[p][m + 4]);                                // r2 <-> lane 5
#pragma unroll
        for (int p = 0; p < NP; p++)
#pragma unroll
            for (int m = 0; m < 4; m++) ct(x[p][m], x[p][m + 4], w.t8[m]);                            // stage 8: bit 0 = r2, block (.., l5, r1, r0)
    }
    // evaluations (layout D, |x| < 2^52) -> 512 * coefficients of the part (layout A, |x| <= 4 q); inverse tables
    template <class Hook = NoHook>
    __device__ static __forceinline__ void inverse_one(double (&x)[E], double *buf, uint32_t ln, const Uniform &u, const TwLane &w,
                                                       Hook &&mid = NoHook{}) {
#pragma unroll
        for (int m = 0; m < E; m++) x[m] = fp_center(x[m]);
#pragma unroll
        for (int m = 0; m < 4; m++) gs(x[m], x[m + 4], w.t8[m]);
#pragma unroll
        for (int m = 0; m < 4; m++) swap32(x[m], x[m + 4]);
#pragma unroll
        for (int q = 0; q < 4; q++) gs(x[2 * q], x[2 * q + 1], w.t7[q]);
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int m = 0; m < 2; m++) gs(x[4 * h + m], x[4 * h + m + 2], w.t6[h]);
#pragma unroll
        for (int m = 0; m < 4; m++) gs(x[m], x[m + 4], w.t5);   // four stages since the centring: <= 8 q
        {
            const uint32_t wc = phys(index_c(ln, 0)), rb = phys(index_b(ln, 0));
            sync();
#pragma unroll
            for (int m = 0; m < E; m++) buf[wc ^ phys(index_c(0, m))] = x[m];
            mid();
            sync();
#pragma unroll
            for (int m = 0; m < E; m++) x[m] = buf[rb ^ phys(index_b(0, m))];
        }
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int m = 0; m < 2; m++) gs(x[4 * h + m], x[4 * h + m + 2], w.t4[(m << 1) | h]);
#pragma unroll
        for (int m = 0; m < 4; m++) gs(x[m], x[m + 4], w.t3[m & 1]);
        // six stages since the centring: <= 32 q (every product's input stayed below 2^52); centre again, so that the three
        // stages left and the caller's two joining stages end at 16 q
#pragma unroll
        for (int m = 0; m < E; m++) x[m] = fp_center(x[m]);
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int m = 0; m < 2; m++) swap16(x[4 * h + m], x[4 * h + m + 2]);
#pragma unroll
        for (int m = 0; m < 4; m++) swap32(x[m], x[m + 4]);
#pragma unroll
        for (int q = 0; q < 4; q++) gs(x[2 * q], x[2 * q + 1], u.u2[q]);
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int m = 0; m < 2; m++) gs(x[4 * h + m], x[4 * h + m + 2], u.u1[h]);
#pragma unroll
        for (int m = 0; m < 4; m++) gs(x[m], x[m + 4], u.u0);   // three stages since the centring: <= 4 q
    }

    // ---- the interface WavesNtt expects of the wave-private transform of a part ------------------------------------------
    struct Xchg {
        double *bufs;
        uint32_t pp;
        uint32_t stride = N;
    };
    template <int FIRST, class Hook>
    __device__ static __forceinline__ void forward(double (&x)[E], Xchg &xc, uint32_t ln, const Twiddles &tw, Hook &&before_last,
                                                   uint32_t R = 1) {
        double y[1][E];
#pragma unroll
        for (int m = 0; m < E; m++) y[0][m] = x[m];
        double *b[1] = {xc.bufs};
        forward_multi<1, FIRST>(y, b, ln, load_uniform(tw.uniform, R), load_lane(tw.lane, ln), before_last);
#pragma unroll
        for (int m = 0; m < E; m++) x[m] = y[0][m];
    }
    struct InvUniform {};
    __device__ static __forceinline__ InvUniform inverse_uniform(uint32_t, const Twiddles &, uint32_t = 1) { return {}; }
    template <bool BOUNDED = false>
    __device__ static __forceinline__ void inverse(double (&x)[E], Xchg &xc, uint32_t ln, const Twiddles &tw, const InvUniform &,
                                                   uint32_t R = 1) {
        inverse_one(x, xc.bufs, ln, load_uniform(tw.uniform, R), load_lane(tw.lane, ln));
    }
};

}  // namespace fbs

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
    __device__ static __forceinline__ void inverse_one(double (&x)[E], double *buf, uint32_t ln, const Tw &w) {
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

}  // namespace fbs

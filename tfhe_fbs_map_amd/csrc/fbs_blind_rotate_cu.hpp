// What the whole-CU blind-rotation kernels share (fbs_blind_rotate_cu.hip: one k = 1 bootstrap on the eight waves of a CU;
// fbs_blind_rotate_k2.hip: one k = 2 bootstrap on six or twelve): the twiddles of a wave-private part transform.
#pragma once
#include "fbs_blind_rotate.hpp"

namespace fbs {

// The twiddles a wave needs for its part, both directions, and the two calls the kernel makes with them (PARTS = waves per
// polynomial: part w hangs from node PARTS + w of the polynomial's twiddle tree).  Every transform a
// lane ever runs uses the same ones.  256-point parts (N = 1024): 9 + 9 per-lane doubles, all in registers.  512-point parts
// (N = 2048): 17 + 17 -- the forward ones stay in registers, the inverse ones are read once per step from a [part][k][lane]
// table the workgroup builds in LDS (64 consecutive words per read).
// LEAN (two workgroups per CU, 128 registers per thread): the inverse twiddles are not kept -- `prefetch_inverse` asks for them
// (18 registers' worth, L1 / L2 hits) when the forward transforms are done, ahead of the products that cover their latency.
template <class Part, bool LEAN, int PARTS = 4>
struct CuTwiddles;
template <bool LEAN, int PARTS>
struct CuTwiddles<LaneNtt256, LEAN, PARTS> {
    static constexpr int LDS_WORDS = 0;
    LaneNtt256::Tw f, i;
    const double *inv_part;
    uniform_doubles inv_big;
    uint32_t root, lane;
    // big_*: the table of the whole polynomial (wave-uniform reads at the part's root 4 + w); tw_*: the four parts' own tables
    __device__ __forceinline__ void init(uniform_doubles big_f, uniform_doubles big_i, const double *tw_fwd, const double *tw_inv, uint32_t w,
                                         uint32_t ln, double *) {
        f = LaneNtt256::load(tw_fwd + w * 256u, big_f, (uint32_t)PARTS + w, ln);
        inv_part = tw_inv + w * 256u, inv_big = big_i, root = (uint32_t)PARTS + w, lane = ln;
        if constexpr (!LEAN) i = LaneNtt256::load(inv_part, inv_big, root, lane);
    }
    template <int NL>
    __device__ __forceinline__ void forward(double (&x)[NL][4], double *const (&bufs)[NL], uint32_t ln) const {
        LaneNtt256::forward_multi<NL, 0>(x, bufs, ln, f, LaneNtt256::NoHook{});
    }
    __device__ __forceinline__ void prefetch_inverse() {
        if constexpr (LEAN) i = LaneNtt256::load(inv_part, inv_big, root, lane);
    }
    template <class Hook>
    __device__ __forceinline__ void inverse(double (&x)[4], double *buf, uint32_t ln, Hook &&mid) const {
        LaneNtt256::inverse_one(x, buf, ln, i, mid);
    }
};
template <bool LEAN, int PARTS>
struct CuTwiddles<LaneNtt512, LEAN, PARTS> {
    static constexpr int LDS_WORDS = PARTS * LaneNtt512::LANE_TW * 64;
    LaneNtt512::Uniform uf, ui;
    LaneNtt512::TwLane f;
    const double *inv_table;
    __device__ __forceinline__ void init(uniform_doubles big_f, uniform_doubles big_i, const double *tw_fwd, const double *tw_inv, uint32_t w,
                                         uint32_t ln, double *lds) {
        uf = LaneNtt512::load_uniform(big_f, (uint32_t)PARTS + w);
        ui = LaneNtt512::load_uniform(big_i, (uint32_t)PARTS + w);
        f = LaneNtt512::load_lane(tw_fwd + w * 512u, ln);
        for (uint32_t e = threadIdx.x; e < (uint32_t)LDS_WORDS; e += blockDim.x) {   // (made visible by the barrier that follows)
            const uint32_t part = e / (LaneNtt512::LANE_TW * 64u), k = e / 64u % LaneNtt512::LANE_TW, lane = e & 63u;
            uint32_t node = 0;
#pragma unroll
            for (int kk = 0; kk < LaneNtt512::LANE_TW; kk++)
                if ((uint32_t)kk == k) node = LaneNtt512::lane_node(lane, kk);
            lds[e] = tw_inv[part * 512u + node];
        }
        inv_table = lds + w * (LaneNtt512::LANE_TW * 64u);
    }
    template <int NL>
    __device__ __forceinline__ void forward(double (&x)[NL][8], double *const (&bufs)[NL], uint32_t ln) const {
        LaneNtt512::forward_multi<NL, 0>(x, bufs, ln, uf, f, LaneNtt512::NoHook{});
    }
    __device__ __forceinline__ void prefetch_inverse() {}
    template <class Hook>
    __device__ __forceinline__ void inverse(double (&x)[8], double *buf, uint32_t ln, Hook &&mid) const {
        LaneNtt512::inverse_one(x, buf, ln, ui, LaneNtt512::load_lane_table(inv_table, ln), mid);
    }
};

}  // namespace fbs

// Device-side addressing of ciphertexts through a GateView (see fbs_internal.hpp).
#pragma once
#include "fbs_internal.hpp"

namespace fbs {

__device__ __forceinline__ const uint64_t *gate_in(const GateView &gv, size_t f, uint32_t ct_words) {
    size_t g = f / gv.s_count, s = gv.s_begin + f % gv.s_count;
    size_t slot = gv.src_slot ? gv.src_slot[g] : g;
    return gv.in_base + (slot * gv.T + s) * ct_words;
}
__device__ __forceinline__ uint64_t *gate_out(const GateView &gv, size_t f, uint32_t ct_words) {
    size_t g = f / gv.s_count, s = gv.s_begin + f % gv.s_count;
    size_t slot = gv.dst_slot ? gv.dst_slot[g] : g;
    return gv.out_base + (slot * gv.T + s) * ct_words;
}

}  // namespace fbs

// Device-side addressing of ciphertexts through a GateView (see fbs_internal.hpp).
#pragma once
#include "fbs_internal.hpp"

namespace fbs {

// input ciphertext of key switch r of the launch (r < ks_count)
__device__ __forceinline__ const uint64_t *ks_in(const GateView &gv, size_t r, uint32_t ct_words) {
    const size_t fk = gv.ks_begin + r;
    const size_t u = fk / gv.s_count, s = gv.s_begin + fk % gv.s_count;
    const size_t slot = gv.src_slot ? gv.src_slot[u] : u;
    return gv.in_base + (slot * gv.T + s) * ct_words;
}
// bootstrap i of the launch (i < count): its gate, and the row of the modulus-switched scratch it rotates by
__device__ __forceinline__ void gate_of(const GateView &gv, size_t i, size_t *gate, size_t *ms_row) {
    const size_t f = gv.f_begin + i;
    const size_t g = f / gv.s_count, s = f % gv.s_count;
    const size_t u = gv.source_of ? gv.source_of[g] : g;
    *gate = g;
    *ms_row = u * gv.s_count + s - gv.ks_begin;
}
__device__ __forceinline__ uint64_t *gate_out(const GateView &gv, size_t i, uint32_t ct_words) {
    if (gv.out_rows) return gv.out_rows + i * (gv.row_words ? gv.row_words : ct_words);
    const size_t f = gv.f_begin + i;
    const size_t g = f / gv.s_count, s = gv.s_begin + f % gv.s_count;
    const size_t slot = gv.dst_slot ? gv.dst_slot[g] : g;
    return gv.out_base + (slot * gv.T + s) * ct_words;
}

// fused programs: the scratch row a rotation of TV_0 leaves its whole accumulator in, or null for an ordinary gate
// (`acc_words` = (k + 1) N: the whole GLWE accumulator, mask polynomials then body)
__device__ __forceinline__ uint64_t *gate_acc(const GateView &gv, size_t i, uint32_t acc_words) {
    const bool in_row = gv.out_rows && gv.row_words;   // a fused level cut across GPUs: the accumulator goes into the gate's row
    if (!gv.acc_rows && !in_row) return nullptr;
    const size_t f = gv.f_begin + i;
    const size_t g = f / gv.s_count, s = f % gv.s_count;
    const uint32_t d = gv.dst_slot[g];
    if (!(d & 0x80000000u)) return nullptr;
    if (in_row) return gv.out_rows + i * gv.row_words;
    return gv.acc_rows + ((size_t)(d & 0x7FFFFFFFu) * gv.s_count + s) * acc_words;
}

}  // namespace fbs

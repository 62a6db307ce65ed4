// What the blind-rotation kernels of fbs_blind_rotate.hip and fbs_blind_rotate_cu.hip share: launch arguments, the issue-priority
// hand-over of the two waves of a SIMD, and the launcher of the one-bootstrap-per-CU shape.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "fbs_gate.hpp"
#include "fbs_internal.hpp"
#include "fbs_ntt.hpp"
#include "fbs_ntt_split.hpp"

namespace fbs {

struct BrArgs {
    GateView gv;
    const uint32_t *ms;      // [count][n+1], values in [0, 2N)
    const double *bsk_hat;   // [n][rows][2][N]  centred, NTT order, times 1/N
    const double *tw_fwd, *tw_inv;
    const uint64_t *tvs;     // [tables][N]
    const uint64_t *post;    // [tables]
    uint32_t n, l, beta, ct_words, n_tables;
    size_t count;            // bootstraps in this launch
    const double *psi_pow;   // [N] psi^x, centred (two key bits per step only)
};

// Issue priority of the two waves that share a SIMD.  They sit in wave slots 0 and 1 of it (HW_ID bits 3:0); left alone,
// the SIMD issues the older one first whenever both are ready.  PRIO = 1: the priority is raised on even steps in one slot
// and on odd steps in the other; PRIO = 7: it also changes hands in the middle of a step (before the inverse transform), so
// that within each half of a step one wave leads and the other fills its stalls, and neither leads for long.
// Measured (tools/selector_bench.py, one box, FBS/s without / PRIO 1 / PRIO 7): N = 1024 one polynomial per wave: p = 2
// 148.3 / 153.0 / 155.0 k, p = 4 125.0 / 126.8 / 129.7 k, and the benchmark shape in whole-CU workgroups 100.0 / 106.8 /
// 110.0 k; N = 2048 (two waves per polynomial): (15, 70) 100.1 / 101.0 / 36 k, (31, 325) 46.3 / 46.8 / 24 k -- a change of
// hands between the barriers of a multi-wave transform stalls the polynomial's other wave.
#ifndef FBS_PRIO_ONE_WAVE
#define FBS_PRIO_ONE_WAVE 7    // polynomials that live in one wave (N <= 1024)
#endif
#ifndef FBS_PRIO_MULTI_WAVE
#define FBS_PRIO_MULTI_WAVE 1  // polynomials spread over several waves
#endif
__device__ __forceinline__ uint32_t wave_slot_parity() {
    uint32_t hw_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
    return hw_id & 1u;
}
template <int PRIO>
__device__ __forceinline__ void lead_if(uint32_t turn, uint32_t slot) {
    if constexpr (PRIO != 0) {
        if ((turn ^ slot) & 1u) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    }
}

// fbs_blind_rotate_cu.hip: one bootstrap on the eight waves of a CU (N = 1024, at most four gadget levels).  Returns false when
// there is no instantiation for the context's parameters (the caller then takes the generic kernel); *kernel = its name.
bool launch_blind_rotate_cu(fbs_ctx *ctx, const BrArgs &a, hipStream_t stream, std::string *kernel);
// ... with two key bits per step (N = 2048, one gadget level)
bool launch_blind_rotate_cu_pairs(fbs_ctx *ctx, const BrArgs &a, hipStream_t stream, std::string *kernel);
void blind_rotate_cu_catalog(std::vector<std::string> *out);

}  // namespace fbs

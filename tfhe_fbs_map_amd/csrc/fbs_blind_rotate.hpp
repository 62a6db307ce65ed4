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

// Key words through BUFFER loads.  The resource (base address of the step's key rows, four SGPRs) and the distance to the
// polynomial wanted (one SGPR) are wave-uniform; every thread supplies ONE 32-bit byte offset, and register pairs further along the
// polynomial are reached through the instruction's immediate offset -- no 64-bit address arithmetic on the vector pipe and no
// register pair per address (flat global loads: 78 of the 2 873 vector instructions of a k_blind_rotate_pairs<11,7,4> step and
// more than thirty registers went into addresses).  `base` must be wave-uniform.
// -DFBS_EXP_HOT_KEYS=1 (experiments only: WRONG results): every step reads the key rows of step 0 or 1, which stay in L2 -- what a
// launch would take if no key word ever came from further away than L2
#ifndef FBS_EXP_HOT_KEYS
#define FBS_EXP_HOT_KEYS 0
#endif
#define FBS_KEY_STEP(i) (FBS_EXP_HOT_KEYS ? ((i) & 1u) : (i))
struct KeyRows {
    __amdgpu_buffer_rsrc_t rsrc;
    __device__ __forceinline__ explicit KeyRows(const double *base)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(base), 0, 0x7FFFFFFF, 0x00020000)) {}
    // 16 bytes at base + poly_bytes + thread_bytes (+ the immediate the compiler splits off thread_bytes' constant part)
    __device__ __forceinline__ double2 load(uint32_t thread_bytes, uint32_t poly_bytes) const {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, thread_bytes, poly_bytes, 0);
        double2 r;
        r.x = __hiloint2double((int)v.y, (int)v.x);
        r.y = __hiloint2double((int)v.w, (int)v.z);
        return r;
    }
};

// fbs_blind_rotate_cu.hip: one bootstrap on the eight waves of a CU (N = 1024, at most four gadget levels).  Returns false when
// there is no instantiation for the context's parameters (the caller then takes the generic kernel); *kernel = its name.
bool launch_blind_rotate_cu(fbs_ctx *ctx, const BrArgs &a, hipStream_t stream, std::string *kernel);
// ... with two key bits per step (N = 2048, one gadget level)
bool launch_blind_rotate_cu_pairs(fbs_ctx *ctx, const BrArgs &a, hipStream_t stream, std::string *kernel);
void blind_rotate_cu_catalog(std::vector<std::string> *out);
// fbs_blind_rotate_k2.hip: GLWE dimension k = 2 at N = 1024 (two key bits per step, one gadget level): three waves per bootstrap and
// one / two / four bootstraps per workgroup, or one bootstrap on the twelve waves of a workgroup (launches that leave most of the
// chip empty).  Returns false when the context is not of that shape.
constexpr size_t K2_CU_ROUNDS = 3;   // bootstraps per CU up to which a k = 2 launch takes the twelve-wave shape, round after round
bool launch_blind_rotate_k2(fbs_ctx *ctx, const BrArgs &a, hipStream_t stream, std::string *kernel);
void blind_rotate_k2_catalog(std::vector<std::string> *out);
// fbs_blind_rotate_glwe.hip: every other (k >= 2, N <= 1024, l, key bits per step): k + 1 waves per bootstrap, one wave per polynomial.
// Returns false when no instantiation is built for the context's (N, k).
// fpw: bootstraps per workgroup -- 1, 2, or anything else for the throughput shape (glwe_full_fpw of them)
bool launch_blind_rotate_glwe(fbs_ctx *ctx, const BrArgs &a, int fpw, hipStream_t stream, std::string *kernel);
int glwe_full_fpw(uint32_t log_n, uint32_t k);
void blind_rotate_glwe_catalog(std::vector<std::string> *out);

}  // namespace fbs

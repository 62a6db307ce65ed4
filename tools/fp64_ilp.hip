// FP64 FMA issue rate against instruction-level parallelism and waves per SIMD (gfx950): how many independent chains
// does a wave need before dependent v_fma_f64 stop costing issue slots?  hipcc --offload-arch=gfx950 -O3 fp64_ilp.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP>
__global__ __launch_bounds__(64) void chains(double *out, double b, double c, int iters) {
    double a[ILP];
#pragma unroll
    for (int i = 0; i < ILP; i++) a[i] = threadIdx.x + i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < ILP; i++) a[i] = __builtin_fma(a[i], b, c);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < ILP; i++) s += a[i];
    out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int ILP>
void run(double *d, int waves_per_simd) {
    const int iters = 20000;
    const int grid = 1024 * waves_per_simd;   // 256 CUs x 4 SIMDs
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    chains<ILP><<<grid, 64>>>(d, 0.999999, 1e-9, 10);
    hipEventRecord(e0);
    chains<ILP><<<grid, 64>>>(d, 0.999999, 1e-9, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_wave = (double)iters * 8 * ILP;
    // cycles per wave-instruction per SIMD at 2.4 GHz nominal
    printf("ILP %d waves/SIMD %d: %.3f ms, %.2f cycles per FMA per SIMD (2.4 GHz)\n", ILP, waves_per_simd, ms,
           ms * 1e-3 * 2.4e9 / (instr_per_wave * waves_per_simd));
}
int main() {
    double *d; hipMalloc(&d, 1024 * 8 * 64 * 8);
    for (int w : {1, 2, 4}) { run<1>(d, w); run<2>(d, w); run<3>(d, w); run<4>(d, w); run<6>(d, w); run<8>(d, w); }
    return 0;
}

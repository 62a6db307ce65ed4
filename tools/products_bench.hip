// Issue rate of the PRODUCTS phase of k_blind_rotate_cu_pairs in isolation (gfx950): the bundle words (three exact products
// and their lazy sum per key word) times the digits' evaluations, on registers only -- no loads, no LDS, no barriers -- at two
// waves per SIMD.  If this runs near the FP64 issue ceiling (4.8-5.4 cycles per instruction, tools/fp64_ilp.hip), what slows
// the phase down in the kernel is not its arithmetic.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I../tfhe_fbs_map_amd/csrc products_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "fbs_field.hpp"
using namespace fbs;
template <int NL>
__global__ __launch_bounds__(512, 2) void products(const double *in, double *out, int iters) {
    constexpr int E = 8;
    double x[NL][E], own[E], other[E];
    double2 ko[3][NL][E / 2], kt[3][NL][E / 2];
    const double *p = in + threadIdx.x;
    int q = 0;
#pragma unroll
    for (int lv = 0; lv < NL; lv++)
#pragma unroll
        for (int m = 0; m < E; m++) x[lv][m] = p[512 * q++];
#pragma unroll
    for (int jj = 0; jj < 3; jj++)
#pragma unroll
        for (int lv = 0; lv < NL; lv++)
#pragma unroll
            for (int j = 0; j < E / 2; j++) {
                ko[jj][lv][j] = double2{p[512 * q], p[512 * (q + 1)]};
                kt[jj][lv][j] = double2{p[512 * (q + 2)], p[512 * (q + 3)]};
                q += 4;
            }
#pragma unroll
    for (int m = 0; m < E; m++) own[m] = other[m] = 0.0;
    double mo[3] = {p[512 * q], p[512 * (q + 1)], p[512 * (q + 2)]};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int jj = 0; jj < 3; jj++) mo[jj] = fp_center(mo[jj] + 12345.0);   // (changes every iteration: nothing to hoist)
#pragma unroll
        for (int m = 0; m < E; m++) {
            double so = 0.0, st = 0.0;
#pragma unroll
            for (int lv = 0; lv < NL; lv++) {
                double wo = 0.0, wt = 0.0;
#pragma unroll
                for (int jj = 0; jj < 3; jj++) {
                    const double2 a = ko[jj][lv][m >> 1], b = kt[jj][lv][m >> 1];
                    wo += fp_mulmod((m & 1) ? a.y : a.x, mo[jj]);
                    wt += fp_mulmod((m & 1) ? b.y : b.x, mo[jj]);
                }
                so += fp_mulmod(x[lv][m], wo);
                st += fp_mulmod(x[lv][m], wt);
            }
            own[m] = fp_center(own[m] + so);
            other[m] = fp_center(other[m] + st);
        }
    }
    double s = 0;
#pragma unroll
    for (int m = 0; m < E; m++) s += own[m] + other[m];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int NL>
void run(const double *in, double *out) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    products<NL><<<256, 512>>>(in, out, 10);
    hipEventRecord(e0);
    products<NL><<<256, 512>>>(in, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per iteration and thread: 8 registers x NL levels x 2 columns x (3 products of 6 + 3 adds + 1 product of 6 + 1 add) + 8 x 2 x 4 + 3 x 4
    const double instr = (double)iters * (8.0 * NL * 2 * (3 * 6 + 3 + 6 + 1) + 8 * 2 * 4 + 12);
    printf("NL %d: %.3f ms, %.2f cycles per instruction per SIMD at two waves per SIMD (2.4 GHz nominal), by the source's count of %.0f per iteration\n",
           NL, ms, ms * 1e-3 * 2.4e9 / (instr * 2), instr / iters);
}
int main() {
    double *in, *out;
    hipMalloc(&in, 512 * 256 * 8);
    hipMalloc(&out, 256 * 512 * 8);
    double *h = new double[512 * 256];
    for (int i = 0; i < 512 * 256; i++) h[i] = (double)((i * 2654435761u) % 1000003) * 7919.0 - 3.0e9;
    hipMemcpy(in, h, 512 * 256 * 8, hipMemcpyHostToDevice);
    run<1>(in, out);
    run<2>(in, out);
    return 0;
}

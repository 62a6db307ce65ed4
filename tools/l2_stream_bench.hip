// How fast can ONE CU pull a stream out of L2 (gfx950)?  Every workgroup (one per CU, 512 threads) reads the SAME buffer from the
// start, 16 bytes per lane and load, U loads in flight per thread -- the access pattern of the bootstrapping key in the whole-CU
// blind-rotation kernels, where every CU walks the same key rows in step (L2 hits after the first touch) and nothing is reused
// inside a CU.   hipcc --offload-arch=gfx950 -O3 l2_stream_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int U>
__global__ __launch_bounds__(512, 2) void stream(const double2 *in, double *out, size_t words_per_pass, int passes) {
    double s = 0;
    for (int p = 0; p < passes; p++)
        for (size_t base = 0; base + 512 * U <= words_per_pass; base += 512 * U) {
            double2 v[U];
#pragma unroll
            for (int u = 0; u < U; u++) v[u] = in[base + u * 512 + threadIdx.x];
#pragma unroll
            for (int u = 0; u < U; u++) s += v[u].x + v[u].y;
        }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int U>
void run(const double2 *in, double *out, size_t bytes, int grid) {
    const size_t words = bytes / 16;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    stream<U><<<grid, 512>>>(in, out, words, 1);
    hipEventRecord(e0);
    stream<U><<<grid, 512>>>(in, out, words, 4);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%3d workgroups, %2d loads of 16 B in flight per thread, buffer %4zu MB: %.1f GB/s per CU, %.2f TB/s over the chip\n", grid, U,
           bytes >> 20, 4.0 * bytes / (ms * 1e-3) / 1e9, 4.0 * bytes * grid / (ms * 1e-3) / 1e12);
}
int main() {
    const size_t bytes = 150u << 20;
    double2 *in; double *out;
    hipMalloc(&in, bytes); hipMalloc(&out, 256 * 512 * 8);
    hipMemset(in, 0, bytes);
    for (int grid : {256, 64, 8}) { run<4>(in, out, bytes, grid); run<12>(in, out, bytes, grid); run<24>(in, out, bytes, grid); }
    return 0;
}

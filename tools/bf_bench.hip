// Butterfly-throughput probe: Goldilocks u64 butterfly vs exact FP64-FMA butterfly modulo a 46-bit prime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include "../tfhe_fbs_map_amd/csrc/fbs_field.hpp"
using namespace fbs;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(128) void k_int(uint64_t *io, const uint64_t *tw, int rounds) {
    uint64_t x[16];
    for (int m = 0; m < 16; m++) x[m] = io[(blockIdx.x * 128 + threadIdx.x) * 16 + m];
    for (int r = 0; r < rounds; r++) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int hm = 8 >> s;
#pragma unroll
            for (int m = 0; m < 16; m++) {
                if (m & hm) continue;
                const uint64_t w = tw[(r & 7) * 16 + (s * 4 + (m & 3))];
                const uint64_t u = x[m], v = gl_mul(x[m + hm], w);
                x[m] = gl_add_lc(u, v);
                x[m + hm] = gl_sub_lc(u, v);
            }
        }
    }
    for (int m = 0; m < 16; m++) io[(blockIdx.x * 128 + threadIdx.x) * 16 + m] = x[m];
}

__device__ __forceinline__ double mulmod(double x, double w, double P, double pinv) {
    double h = x * w;
    double l = __builtin_fma(x, w, -h);
    double q = __builtin_rint(h * pinv);
    double r = __builtin_fma(-q, P, h);
    return r + l;
}
__global__ __launch_bounds__(128) void k_fp(double *io, const double *tw, int rounds, double P, double pinv) {
    double x[16];
    for (int m = 0; m < 16; m++) x[m] = io[(blockIdx.x * 128 + threadIdx.x) * 16 + m];
    for (int r = 0; r < rounds; r++) {
#pragma unroll
        for (int s = 0; s < 4; s++) {
            const int hm = 8 >> s;
#pragma unroll
            for (int m = 0; m < 16; m++) {
                if (m & hm) continue;
                const double w = tw[(r & 7) * 16 + (s * 4 + (m & 3))];
                const double u = x[m], v = mulmod(x[m + hm], w, P, pinv);
                x[m] = u + v;
                x[m + hm] = u - v;
            }
        }
        // every 4 stages: bring everything back to (-P/2, P/2)
#pragma unroll
        for (int m = 0; m < 16; m++) x[m] = __builtin_fma(-__builtin_rint(x[m] * pinv), P, x[m]);
    }
    for (int m = 0; m < 16; m++) io[(blockIdx.x * 128 + threadIdx.x) * 16 + m] = x[m];
}

int main() {
    const int blocks = 1024, threads = 128, rounds = 2000;
    const size_t n = (size_t)blocks * threads * 16;
    uint64_t *d_i, *d_tw; double *d_f, *d_tf;
    CHECK(hipMalloc(&d_i, n * 8)); CHECK(hipMalloc(&d_f, n * 8)); CHECK(hipMalloc(&d_tw, 128 * 8)); CHECK(hipMalloc(&d_tf, 128 * 8));
    const double P = 70368744161281.0;   // 2^46 - 2^14 + 1? placeholder odd value; only timing matters here
    std::vector<uint64_t> hi(n), ht(128); std::vector<double> hf(n), htf(128);
    for (size_t i = 0; i < n; i++) { hi[i] = (i * 0x9E3779B97F4A7C15ull) % GQ; hf[i] = (double)((i * 0x9E3779B97F4A7C15ull) % (uint64_t)P); }
    for (int i = 0; i < 128; i++) { ht[i] = (0x1234567ull * (i + 3) * 0x9E3779B97F4A7C15ull) % GQ; htf[i] = (double)(ht[i] % (uint64_t)P) - P / 2; }
    CHECK(hipMemcpy(d_i, hi.data(), n * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_f, hf.data(), n * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_tw, ht.data(), 1024, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_tf, htf.data(), 1024, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; rep++) {
        float ms;
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k_int, dim3(blocks), dim3(threads), 0, 0, d_i, d_tw, rounds); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        double bf = (double)blocks * threads * rounds * 32;
        printf("u64 Goldilocks : %.2f ms  %.1f G butterflies/s\n", ms, bf / ms / 1e6);
        CHECK(hipEventRecord(e0)); hipLaunchKernelGGL(k_fp, dim3(blocks), dim3(threads), 0, 0, d_f, d_tf, rounds, P, 1.0 / P); CHECK(hipEventRecord(e1)); CHECK(hipDeviceSynchronize());
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("f64 46-bit     : %.2f ms  %.1f G butterflies/s\n", ms, bf / ms / 1e6);
    }
    return 0;
}

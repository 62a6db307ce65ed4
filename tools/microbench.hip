// Instruction-throughput probe for gfx950: cycles per wave64 instruction per SIMD for the integer ops the
// Goldilocks butterfly is made of.  Not part of the product; results are quoted in DESIGN.md.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// 8 independent chains, 8 instructions each per asm block; ITER blocks
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int OP>
__global__ void probe(uint64_t *out, int iters) {
    uint32_t a[8], b[8], c[8], d[8];
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 7 + i; b[i] = threadIdx.x * 13 + 5 * i + 1; c[i] = i + threadIdx.x; d[i] = 3 * i; }
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#define ONE(i) \
        if constexpr (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])); \
        else if constexpr (OP == 1) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])); \
        else if constexpr (OP == 2) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])); \
        else if constexpr (OP == 3) { uint64_t acc = ((uint64_t)c[i] << 32) | a[i]; asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc) : "v"(b[i]), "v"(d[i]) : "vcc"); a[i] = (uint32_t)acc; c[i] = (uint32_t)(acc >> 32); } \
        else if constexpr (OP == 4) asm volatile("v_add_co_u32 %0, vcc, %0, %1\n v_addc_co_u32 %2, vcc, %2, %3, vcc" : "+v"(a[i]), "+v"(c[i]) : "v"(b[i]), "v"(d[i]) : "vcc"); \
        else if constexpr (OP == 5) { uint64_t acc = ((uint64_t)c[i] << 32) | a[i]; uint64_t o = ((uint64_t)d[i] << 32) | b[i]; asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(acc) : "v"(o)); a[i] = (uint32_t)acc; c[i] = (uint32_t)(acc >> 32); } \
        else if constexpr (OP == 6) { uint64_t x = ((uint64_t)c[i] << 32) | a[i]; uint64_t y = ((uint64_t)d[i] << 32) | b[i]; asm volatile("v_cmp_lt_u64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(a[i]) : "v"(x), "v"(y), "v"(b[i]) : "vcc"); } \
        else if constexpr (OP == 7) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i])); \
        else if constexpr (OP == 8) asm volatile("v_mul_hi_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(b[i])); \
        else if constexpr (OP == 9) asm volatile("v_alignbit_b32 %0, %0, %1, 7" : "+v"(a[i]) : "v"(b[i])); \
        else if constexpr (OP == 10) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(c[i])); \
        else if constexpr (OP == 11) { double x = __hiloint2double(c[i], a[i]); double y = __hiloint2double(d[i], b[i]); asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(y)); a[i] = __double2loint(x); c[i] = __double2hiint(x); } \
        else if constexpr (OP == 12) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b[i])); \
        else if constexpr (OP == 13) asm volatile("v_sub_co_u32 %0, vcc, %0, %1\n v_subb_co_u32 %2, vcc, %2, %3, vcc" : "+v"(a[i]), "+v"(c[i]) : "v"(b[i]), "v"(d[i]) : "vcc"); \
        else if constexpr (OP == 14) { uint64_t acc = ((uint64_t)c[i] << 32) | a[i]; asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(acc)); a[i] = (uint32_t)acc; c[i] = (uint32_t)(acc >> 32); } \
        else if constexpr (OP == 15) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
        REP8(ONE) REP8(ONE) REP8(ONE) REP8(ONE)
#undef ONE
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint32_t s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + c[i];
    if (threadIdx.x == 0) out[blockIdx.x * 2] = t1 - t0;
    out[blockIdx.x * 2 + 1] = s;
}

template <int OP>
int run(const char *name, int insts_per_one) {
    uint64_t *d;
    const int iters = 2000;
    printf("%-28s", name);
    for (int waves_per_simd : {1, 2, 4}) {
        int blocks = 256 * waves_per_simd;          // 256-thread blocks: 4 waves = one per SIMD
        CHECK(hipMalloc(&d, blocks * 16));
        hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, d, 10);
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe<OP>, dim3(blocks), dim3(256), 0, 0, d, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<uint64_t> h(blocks * 2);
        CHECK(hipMemcpy(h.data(), d, blocks * 16, hipMemcpyDeviceToHost));
        double cyc = 0; for (int b = 0; b < blocks; b++) cyc += h[2 * b]; cyc /= blocks;
        double n_inst = (double)iters * 32 * insts_per_one;
        // s_memtime counts at 100 MHz-ish constant clock? report both memtime ticks and wall-derived ns
        double ns_per_inst_per_simd = ms * 1e6 / (n_inst * waves_per_simd);
        printf("  w/simd=%d: %.2f tick/inst/wave  %.3f ns/inst/SIMD", waves_per_simd, cyc / n_inst, ns_per_inst_per_simd);
        CHECK(hipFree(d));
    }
    printf("\n");
    return 0;
}

int main() {
    run<0>("v_add_u32", 1);
    run<15>("v_xor_b32", 1);
    run<1>("v_mul_lo_u32", 1);
    run<2>("v_mul_hi_u32", 1);
    run<3>("v_mad_u64_u32", 1);
    run<4>("v_add_co+v_addc_co (pair)", 2);
    run<13>("v_sub_co+v_subb_co (pair)", 2);
    run<5>("v_lshl_add_u64", 1);
    run<6>("v_cmp_lt_u64+v_cndmask", 2);
    run<12>("v_cndmask_b32", 1);
    run<7>("v_mad_u32_u24", 1);
    run<8>("v_mul_hi_u32_u24", 1);
    run<9>("v_alignbit_b32", 1);
    run<10>("v_add3_u32", 1);
    run<14>("v_lshlrev_b64", 1);
    run<11>("v_fma_f64", 1);
    return 0;
}

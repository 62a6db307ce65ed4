// HIP kernels of libfbsexec.so other than the blind rotation (fbs_blind_rotate.hip), gfx950 only:
//
//   k_ks_digits / k_ks_gemm / k_ks_gemm_finish   LWE key switch kN -> n as an int8 GEMM on the matrix cores (batches > 64)
//   k_keyswitch_fp / k_keyswitch_lanes / k_keyswitch   the same on the FP64 / integer vector pipes (fallback, small batches)
//   k_ms_body                          the body word of the mean-compensated modulus switch q -> 2N
//   k_multi_extract                    tables cut out of a shared blind rotation (fused programs)
//   k_lincomb                         LinearProd over wire slots
//
// plus the profiling helpers and the launchers for those kernels.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>

#include "fbs_gate.hpp"
#include "fbs_internal.hpp"

namespace fbs {

// ---------------------------------------------------------------------------------------------
// key switch + modulus switch
//
// out = (0, b) - sum_j sum_v digit_v(a_j) * KSK[j][v]; BALANCED base-2^gamma digits (in [-B/2, B/2), carries
// propagated) of the closest multiple of q/2^(t*gamma) (top t*gamma bits of the 46-bit word, rounded).  The kernels
// never see a signed digit: adding B/2 at every digit position before cutting the word into bit fields gives
// u_v = digit_v + B/2 in [0, B), and sum_v digit_v K_v = sum_v u_v K_v - (B/2) sum_v K_v, whose second term summed over
// all j is one constant vector per key (`corr`, built at key upload).  Balanced digits carry about a quarter of the
// key-noise variance of unsigned ones at no cost in the loop.  Key words are < 2^46 and there are kN*t fields
// < 2^gamma per output, so a plain 64-bit accumulator holds the whole sum (checked in dev_supported) and is
// folded mod q once at the end; the result is switched to [0, 2N) on the spot.
// ---------------------------------------------------------------------------------------------
struct KsArgs {
    GateView gv;
    const uint64_t *ksk;   // [D*t][stride]
    const uint64_t *corr;  // [stride]  (B/2) * sum of all key rows, mod q
    uint32_t *ms;          // [count][n+1]
    // Mean-compensated modulus switch: the rounding errors eps_i of the n mask words reach the phase as sum eps_i s_i
    // through a BINARY key (mean 1/2), and the evaluator knows them: every kernel adds what its columns left into eps[f],
    // parks the unrounded body in body_raw[f], and k_ms_body rounds body - (sum eps) / 2 (and clears eps[f] for the next
    // launch).  The phase error left is sum eps_i (s_i - 1/2): n/4 roundings of variance instead of n/2.
    unsigned long long *eps;   // [count]  two's complement sums, zero on entry
    uint64_t *body_raw;        // [count]
    uint32_t offs;         // B/2 at every digit position
    uint32_t n, D, t, gamma, stride, ct_words, log2_2n;
    uint32_t cols_major;   // 1: blockIdx.x walks the column blocks (see dev_keyswitch), 0: the ciphertext tiles
    size_t row0;           // first key switch of this launch (launches are split when a grid dimension would overflow)
    size_t count;          // key switches in all (gv.ks_count)
};

// rounded top t*gamma bits plus the digit offsets; bits above t*gamma (the rounding carry: a multiple of q) are never looked at
__device__ __forceinline__ uint32_t ks_round(uint64_t w, uint32_t tg, uint32_t offs) {
    return (uint32_t)(((w >> (FQ_BITS - 1 - tg)) + 1) >> 1) + offs;
}
__device__ __forceinline__ uint64_t ks_finish(uint64_t acc, uint64_t body, uint64_t corr) {
    return fq_add(fq_sub(body, acc % FQ), corr);
}
// One word r of the switched ciphertext to [0, 2N) (q treated as 2^46).  Mask words (col < n): rounded, stored, the
// rounding error added to *eps.  The body (col == n) is parked unrounded for k_ms_body.
__device__ __forceinline__ void ms_store(const KsArgs &a, size_t f, uint32_t col, uint64_t r, int64_t *eps) {
    if (col == a.n) {
        a.body_raw[f] = r;
        return;
    }
    const uint32_t sh = FQ_BITS - a.log2_2n;   // bits dropped
    const uint64_t m = ((r >> (sh - 1)) + 1) >> 1;
    *eps += (int64_t)r - (int64_t)(m << sh);
    a.ms[f * (a.n + 1) + col] = (uint32_t)m & ((1u << a.log2_2n) - 1u);
}
__device__ __forceinline__ void ms_flush(const KsArgs &a, size_t f, int64_t eps) {
    if (eps) atomicAdd(a.eps + f, (unsigned long long)eps);
}
__global__ __launch_bounds__(256) void k_ms_body(uint32_t *ms, unsigned long long *eps, const uint64_t *body_raw, uint32_t n,
                                                 uint32_t log2_2n, size_t count) {
    const size_t f = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (f >= count) return;
    const int64_t e = (int64_t)eps[f];
    eps[f] = 0;
    const uint64_t body = fq_sub(body_raw[f], fq_from_i64(e >> 1));   // (the halving floors)
    const uint32_t sh = FQ_BITS - log2_2n;
    ms[f * (n + 1) + n] = (uint32_t)(((body >> (sh - 1)) + 1) >> 1) & ((1u << log2_2n) - 1u);
}

// Small batches: workgroup = FB ciphertexts x 256 output columns; lanes are columns, the key row is read once
// (coalesced) and used FB times.
template <int FB>
__global__ __launch_bounds__(256) void k_keyswitch(KsArgs a) {
    extern __shared__ uint32_t abar[];   // [FB][D]
    __shared__ const uint64_t *ct_ptr[FB];
    const size_t f0 = (size_t)blockIdx.x * FB;
    if (threadIdx.x < FB) ct_ptr[threadIdx.x] = f0 + threadIdx.x < a.count ? ks_in(a.gv, f0 + threadIdx.x, a.ct_words) : nullptr;
    __syncthreads();
    const uint32_t col = blockIdx.y * 256u + threadIdx.x;
    const uint32_t tg = a.t * a.gamma;

    for (uint32_t idx = threadIdx.x; idx < FB * a.D; idx += 256) {
        const uint32_t f = idx / a.D, j = idx % a.D;
        abar[idx] = ct_ptr[f] ? ks_round(ct_ptr[f][j], tg, a.offs) : 0u;
    }
    __syncthreads();

    uint64_t acc[FB];
#pragma unroll
    for (int f = 0; f < FB; f++) acc[f] = 0;
    const uint32_t dmask = (1u << a.gamma) - 1u;
    const uint64_t *kcol = a.ksk + col;   // stride is padded to a multiple of 256: always in bounds

    for (uint32_t j = 0; j < a.D; j++) {
        uint32_t ab[FB];
#pragma unroll
        for (int f = 0; f < FB; f++) ab[f] = abar[f * a.D + j];
        for (uint32_t v = 0; v < a.t; v++) {
            const uint64_t kw = kcol[((size_t)j * a.t + v) * a.stride];
            const uint32_t sh = a.gamma * (a.t - 1 - v);
#pragma unroll
            for (int f = 0; f < FB; f++) acc[f] += (uint64_t)((ab[f] >> sh) & dmask) * kw;
        }
    }
    if (col > a.n) return;
#pragma unroll
    for (int f = 0; f < FB; f++) {
        if (f0 + f >= a.count) break;
        const uint64_t body = col == a.n ? ct_ptr[f][a.D] : 0;
        int64_t eps = 0;
        ms_store(a, f0 + f, col, ks_finish(acc[f], body, a.corr[col]), &eps);
        ms_flush(a, f0 + f, eps);
    }
}

// Batches: LANES are ciphertexts (64 or 128 per workgroup) and the key words a wave needs are wave-uniform, so they
// arrive through the scalar cache (s_load_dwordx16) instead of being re-fetched by every ciphertext tile.  A
// workgroup owns COLS output columns; its waves each take a share of the kN mask words (more waves in flight to
// cover the scalar-load latency, which is what bounds this kernel) and their partial sums meet in LDS.  The rounded mask words are
// staged per tile in LDS, transposed on the way in so that both the global read (along the ciphertext row) and
// the LDS read (along the ciphertexts) are contiguous.
// CPL: ciphertexts per lane (every scalar key load then feeds CPL times the multiply-adds); WAVES: waves per workgroup,
// each taking 1/WAVES of the mask words (more waves in flight per ciphertext tile to cover the scalar-load latency)
template <int COLS, int CPL, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_keyswitch_lanes(KsArgs a) {
    constexpr int JT = 64 / WAVES;               // mask words per wave per staging round (tile = 16 KB * CPL of LDS)
    constexpr int CTS = 64 * CPL;                // ciphertexts per workgroup
    constexpr int THREADS = 64 * WAVES;
    constexpr int TILE_WORDS = WAVES * JT * CTS > (WAVES / 2) * COLS * 2 * CTS ? WAVES * JT * CTS : (WAVES / 2) * COLS * 2 * CTS;
    __shared__ uint32_t tile[TILE_WORDS];        // [slice][j in tile][ciphertext]; reused for the final reduction
    __shared__ const uint64_t *ct_ptr[CTS];      // row base of the ciphertexts (the slot lookup costs a 64-bit division)
    static_assert(WAVES >= 2 && (WAVES & (WAVES - 1)) == 0, "tree reduction over the waves");
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const size_t f0 = a.row0 + (size_t)(a.cols_major ? blockIdx.y : blockIdx.x) * CTS;
    // column-major grid: workgroups with equal blockIdx.x mod 8 share an XCD (round-robin dispatch); give each XCD
    // PAIRS of adjacent column blocks, so that both 64-byte halves of every 128-byte key line are used from one L2
    const uint32_t bx = blockIdx.x;
    const uint32_t col0 = (a.cols_major ? (((bx & 7u) << 1) | ((bx >> 3) & 1u) | ((bx >> 4) << 4)) : blockIdx.y) * COLS;
    if (col0 > a.n) return;                      // padding column block (the grid is padded to a multiple of the 8 XCDs)
    const uint32_t tg = a.t * a.gamma;
    const uint32_t dmask = (1u << a.gamma) - 1u;
    const uint32_t slice_len = (a.D + WAVES - 1) / WAVES;    // mask words per wave

    // two accumulators per column: digit x low word (one v_mad_u64_u32) and digit x high 14 bits (one 24-bit mad);
    // neither can overflow (dev_supported), and there is no carry to move between registers in the loop
    uint64_t acc_lo[CPL][COLS];
    uint32_t acc_hi[CPL][COLS];
#pragma unroll
    for (int u = 0; u < CPL; u++)
#pragma unroll
        for (int c = 0; c < COLS; c++) {
            acc_lo[u][c] = 0;
            acc_hi[u][c] = 0;
        }
    for (uint32_t q = threadIdx.x; q < CTS; q += THREADS) ct_ptr[q] = f0 + q < a.count ? ks_in(a.gv, f0 + q, a.ct_words) : nullptr;

    for (uint32_t r0 = 0; r0 < slice_len; r0 += JT) {
        __syncthreads();
        // the workgroup stages WAVES slices x JT words x CTS ciphertexts: consecutive threads read consecutive words
        for (uint32_t idx = threadIdx.x; idx < WAVES * JT * CTS; idx += THREADS) {
            const uint32_t jj = idx % JT, q = (idx / JT) % CTS, sl = idx / (JT * CTS);
            const uint32_t j = sl * slice_len + r0 + jj;
            uint32_t v = 0;
            const uint64_t *row = ct_ptr[q];
            if (row && r0 + jj < slice_len && j < a.D) v = ks_round(row[j], tg, a.offs);
            tile[(sl * JT + jj) * CTS + q] = v;
        }
        __syncthreads();
        // this wave's mask words of the round.  The key-row address depends only on (wave, jj, v): the compiler keeps
        // it in SGPRs and fetches the COLS words with one s_load_dwordx16 (checked in the ISA: a hand-rolled
        // "next row" prefetch made it fall back to 64-lane vector loads of one address and ran 40 % slower).
        for (uint32_t jj = 0; jj < JT; jj++) {
            const uint32_t j = wave * slice_len + r0 + jj;
            if (r0 + jj >= slice_len || j >= a.D) break;          // wave-uniform
            uint32_t ab[CPL];
#pragma unroll
            for (int u = 0; u < CPL; u++) ab[u] = tile[(wave * JT + jj) * CTS + lane + 64 * u];
            // constant address space: the key is read-only for the life of the kernel and the address is wave-uniform,
            // so these become scalar loads (s_load_dwordx16).  As plain global loads they are 64-lane broadcasts that
            // saturate the vector memory address path (measured: 1.2 ms instead of 0.x ms per 1024-batch).
            typedef const uint64_t __attribute__((address_space(4))) *const_words;
            const const_words krow = (const_words)(uintptr_t)(a.ksk + (size_t)j * a.t * a.stride + col0);
#pragma unroll 4
            for (uint32_t v = 0; v < a.t; v++) {
                uint32_t d[CPL];
#pragma unroll
                for (int u = 0; u < CPL; u++) d[u] = (ab[u] >> (a.gamma * (a.t - 1 - v))) & dmask;
#pragma unroll
                for (int c = 0; c < COLS; c++) {
                    const uint64_t kw = krow[(size_t)v * a.stride + c];
#pragma unroll
                    for (int u = 0; u < CPL; u++) {
                        acc_lo[u][c] += (uint64_t)d[u] * (uint32_t)kw;
                        acc_hi[u][c] += __umul24(d[u], (uint32_t)(kw >> 32));
                    }
                }
            }
        }
    }
    // the partial sums of the waves meet in LDS, halving the number of waves that still hold one each round;
    // wave 0 folds mod q and mod-switches
    uint64_t sum[CPL][COLS];
#pragma unroll
    for (int u = 0; u < CPL; u++)
#pragma unroll
        for (int c = 0; c < COLS; c++) sum[u][c] = acc_lo[u][c] + ((uint64_t)acc_hi[u][c] << 32);
    uint64_t *park = reinterpret_cast<uint64_t *>(tile);
#pragma unroll
    for (int half = WAVES / 2; half >= 1; half /= 2) {
        __syncthreads();
        if (wave >= (uint32_t)half && wave < 2u * half) {
#pragma unroll
            for (int u = 0; u < CPL; u++)
#pragma unroll
                for (int c = 0; c < COLS; c++) park[((wave - half) * COLS + c) * CTS + lane + 64 * u] = sum[u][c];
        }
        __syncthreads();
        if (wave < (uint32_t)half) {
#pragma unroll
            for (int u = 0; u < CPL; u++)
#pragma unroll
                for (int c = 0; c < COLS; c++) sum[u][c] += park[(wave * COLS + c) * CTS + lane + 64 * u];
        }
    }
    if (wave) return;
#pragma unroll
    for (int u = 0; u < CPL; u++) {
        const size_t f = f0 + lane + 64 * u;
        if (f >= a.count) continue;
        const uint64_t body = ct_ptr[lane + 64 * u][a.D];
        int64_t eps = 0;
#pragma unroll
        for (int c = 0; c < COLS; c++) {
            const uint32_t col = col0 + c;
            if (col > a.n) break;
            ms_store(a, f, col, ks_finish(sum[u][c], col == a.n ? body : 0, a.corr[col]), &eps);
        }
        ms_flush(a, f, eps);
    }
}

// The same on the FP64 pipe.  A balanced digit (|d| <= 2^(gamma-1)) times a CENTRED key word (|k| < 2^45) is exact in a
// double, and one v_fma_f64 with the key word as its scalar operand does multiply and accumulate: one VALU instruction
// per (digit, column, ciphertext) where the integer form needs 2.5 (v_mad_u64_u32 for the low word, a 24-bit multiply and
// an add for the high one).  Sums stay exact below 2^53: the accumulators are centred (|.| <= q/2) every `words_per_fold`
// mask words (chosen by the launcher so that 2^45 + words * t * 2^(44 + gamma) <= 2^53).  Measured per 1024-batch at
// P1024: 0.66 -> 0.48 ms.  Key layout: [D*t][stride] centred doubles.
template <int COLS, int CPL, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k_keyswitch_fp(KsArgs a, const double *__restrict__ ksk_f, uint32_t words_per_fold) {
    constexpr int JT = 64 / WAVES;
    constexpr int CTS = 64 * CPL;
    constexpr int THREADS = 64 * WAVES;
    constexpr int TILE_WORDS = WAVES * JT * CTS > (WAVES / 2) * COLS * 2 * CTS ? WAVES * JT * CTS : (WAVES / 2) * COLS * 2 * CTS;
    __shared__ uint32_t tile[TILE_WORDS];
    __shared__ const uint64_t *ct_ptr[CTS];
    static_assert(WAVES >= 2 && (WAVES & (WAVES - 1)) == 0, "tree reduction over the waves");
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const size_t f0 = a.row0 + (size_t)blockIdx.y * CTS;
    const uint32_t bx = blockIdx.x;   // XCD-aware column order, as in k_keyswitch_lanes
    const uint32_t col0 = (((bx & 7u) << 1) | ((bx >> 3) & 1u) | ((bx >> 4) << 4)) * COLS;
    if (col0 > a.n) return;
    const uint32_t tg = a.t * a.gamma;
    const uint32_t slice_len = (a.D + WAVES - 1) / WAVES;

    double acc[CPL][COLS];
#pragma unroll
    for (int u = 0; u < CPL; u++)
#pragma unroll
        for (int c = 0; c < COLS; c++) acc[u][c] = 0.0;
    for (uint32_t q = threadIdx.x; q < CTS; q += THREADS) ct_ptr[q] = f0 + q < a.count ? ks_in(a.gv, f0 + q, a.ct_words) : nullptr;

    uint32_t since_fold = 0;
    for (uint32_t r0 = 0; r0 < slice_len; r0 += JT) {
        __syncthreads();
        for (uint32_t idx = threadIdx.x; idx < WAVES * JT * CTS; idx += THREADS) {
            const uint32_t jj = idx % JT, q = (idx / JT) % CTS, sl = idx / (JT * CTS);
            const uint32_t j = sl * slice_len + r0 + jj;
            // fields digit + B/2 with their top bits flipped = the balanced digits in two's complement, ready for a signed
            // bit-field extract (a missing ciphertext or word: all digits zero)
            uint32_t v = 0;
            const uint64_t *row = ct_ptr[q];
            if (row && r0 + jj < slice_len && j < a.D) v = ks_round(row[j], tg, a.offs) ^ a.offs;
            tile[(sl * JT + jj) * CTS + q] = v;
        }
        __syncthreads();
        for (uint32_t jj = 0; jj < JT; jj++) {
            const uint32_t j = wave * slice_len + r0 + jj;
            if (r0 + jj >= slice_len || j >= a.D) break;          // wave-uniform
            uint32_t ab[CPL];
#pragma unroll
            for (int u = 0; u < CPL; u++) ab[u] = tile[(wave * JT + jj) * CTS + lane + 64 * u];
            typedef const double __attribute__((address_space(4))) *const_doubles;   // wave-uniform, read-only: scalar loads
            const const_doubles krow = (const_doubles)(uintptr_t)(ksk_f + (size_t)j * a.t * a.stride + col0);
#pragma unroll 4
            for (uint32_t v = 0; v < a.t; v++) {
                double d[CPL];
#pragma unroll
                for (int u = 0; u < CPL; u++)
                    d[u] = (double)(int)__builtin_amdgcn_sbfe(ab[u], a.gamma * (a.t - 1 - v), a.gamma);   // balanced digit
#pragma unroll
                for (int c = 0; c < COLS; c++) {
                    const double kw = krow[(size_t)v * a.stride + c];
#pragma unroll
                    for (int u = 0; u < CPL; u++) acc[u][c] = __builtin_fma(d[u], kw, acc[u][c]);
                }
            }
            if (++since_fold == words_per_fold) {
                since_fold = 0;
#pragma unroll
                for (int u = 0; u < CPL; u++)
#pragma unroll
                    for (int c = 0; c < COLS; c++) acc[u][c] = fp_center(acc[u][c]);
            }
        }
    }
#pragma unroll
    for (int u = 0; u < CPL; u++)
#pragma unroll
        for (int c = 0; c < COLS; c++) acc[u][c] = fp_center(acc[u][c]);
    // partial sums of the waves meet in LDS (each below q/2: eight of them stay far below 2^52)
    double *park = reinterpret_cast<double *>(tile);
#pragma unroll
    for (int half = WAVES / 2; half >= 1; half /= 2) {
        __syncthreads();
        if (wave >= (uint32_t)half && wave < 2u * half) {
#pragma unroll
            for (int u = 0; u < CPL; u++)
#pragma unroll
                for (int c = 0; c < COLS; c++) park[((wave - half) * COLS + c) * CTS + lane + 64 * u] = acc[u][c];
        }
        __syncthreads();
        if (wave < (uint32_t)half) {
#pragma unroll
            for (int u = 0; u < CPL; u++)
#pragma unroll
                for (int c = 0; c < COLS; c++) acc[u][c] += park[(wave * COLS + c) * CTS + lane + 64 * u];
        }
    }
    if (wave) return;
#pragma unroll
    for (int u = 0; u < CPL; u++) {
        const size_t f = f0 + lane + 64 * u;
        if (f >= a.count) continue;
        const uint64_t body = ct_ptr[lane + 64 * u][a.D];
        int64_t eps = 0;
#pragma unroll
        for (int c = 0; c < COLS; c++) {
            const uint32_t col = col0 + c;
            if (col > a.n) break;
            // out = body - sum: the sum is signed here, so no correction vector
            const uint64_t sum = fp_to_u64(fp_canon(acc[u][c]));
            ms_store(a, f, col, fq_sub(col == a.n ? body : 0, sum), &eps);
        }
        ms_flush(a, f, eps);
    }
}

// ---------------------------------------------------------------------------------------------
// The key switch as an int8 GEMM on the matrix cores (v_mfma_i32_32x32x32_i8, gfx950).
//
//     out[f][col] = body - sum_{j, v} digit_v(a_j of ciphertext f) * K[(j, v)][col]   (mod q)
//
// is a [ciphertexts] x [kN t] by [kN t] x [n + 1] product.  The balanced digits are tiny (|d| <= 2^(gamma-1)); a centred key
// word (|K| < 2^45) is cut into SIX balanced base-256 limbs, K = sum_b limb_b 256^b with limb_b in [-128, 128), so that
//     C[f][b][col] = sum_kappa digit[f][kappa] * limb_b[kappa][col]
// is an int8 x int8 -> int32 GEMM with N = 6 (n + 1) columns, EXACT (|C| <= 2^(gamma-1) 2^7 kN t < 2^31, checked by the
// launcher), and out = body - sum_b C_b 256^b mod q is recombined by the finishing kernel, which also does the modulus
// switch.  Same integers as the other key-switch kernels, so the same ciphertexts.  Per 1024 bootstraps at P1024 this is
// 6.4e10 int8 multiply-adds x 2 -- 13 us at the dense int8 rate -- where the FP64 form spends 8.3e7 wave-FMAs on the vector
// pipe (0.47 ms measured); the key shrinks from 8 to 6 bytes per word.
//
// Operand fragments are stored the way a wave reads them: tile (32 rows or columns) x k-step (32 values of kappa) x lane x
// 16 bytes, lane l = 32 h + r holding row/column r and the 16 values kappa = 32 ks + 16 h + i.  A and B use the SAME
// assignment of (h, i) to kappa, which is all the instruction needs for a dot product over kappa.  kappa = v * kN + j
// (digit-major), so that the 16 bytes one thread of the digit kernel stores are 16 consecutive mask words of one level.
// ---------------------------------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

struct KsGemm {
    uint32_t D, t, gamma, n, cols_pad;   // cols_pad: n + 1 rounded up to 64; the GEMM has 6 * cols_pad columns (limb-major)
    uint32_t ksteps;                     // D * t / 32
};

// key -> limb fragments, once per key.  One thread per (kappa, column): six bytes, one per limb plane.
__global__ __launch_bounds__(256) void k_ks_limbs(const uint64_t *ksk, uint32_t stride, KsGemm g, int8_t *B) {
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const uint32_t col = (uint32_t)(idx % g.cols_pad);
    const size_t kappa = idx / g.cols_pad;
    if (kappa >= (size_t)g.D * g.t) return;
    const uint32_t v = (uint32_t)(kappa / g.D), j = (uint32_t)(kappa % g.D);
    int64_t x = 0;
    if (col <= g.n) {
        const uint64_t w = ksk[((size_t)j * g.t + v) * stride + col];
        x = w > FQ / 2 ? (int64_t)w - (int64_t)FQ : (int64_t)w;   // centred
    }
    const uint32_t ks = (uint32_t)(kappa >> 5), h = (uint32_t)(kappa >> 4) & 1u, i = (uint32_t)kappa & 15u;
    for (uint32_t b = 0; b < 6; b++) {
        const int64_t limb = ((x + 128) & 255) - 128;   // balanced: in [-128, 128)
        x = (x - limb) >> 8;
        const uint32_t nn = b * g.cols_pad + col;
        B[(((size_t)(nn >> 5) * g.ksteps + ks) * 64 + h * 32 + (nn & 31)) * 16 + i] = (int8_t)limb;
    }
}

// ciphertexts -> digit fragments.  Workgroup = 32 ciphertexts x 8 blocks of 16 mask words; a thread rounds its 16 words and
// stores one 16-byte fragment per level.  Rows past the batch are zero.
__global__ __launch_bounds__(256) void k_ks_digits(KsArgs a, KsGemm g, size_t f0, size_t rows, int8_t *A) {
    const uint32_t r = threadIdx.x & 31, wb = blockIdx.y * 8 + (threadIdx.x >> 5);
    const size_t m = (size_t)blockIdx.x * 32 + r;
    if (wb * 16 >= g.D) return;
    const uint32_t tg = g.t * g.gamma, dmask = (1u << g.gamma) - 1u, half = 1u << (g.gamma - 1);
    uint32_t ab[16];
    if (m < rows) {
        const uint64_t *row = ks_in(a.gv, f0 + m, a.ct_words) + (size_t)wb * 16;
#pragma unroll
        for (int i = 0; i < 16; i++) ab[i] = ks_round(row[i], tg, a.offs);
    }
    for (uint32_t v = 0; v < g.t; v++) {
        const uint32_t sh = g.gamma * (g.t - 1 - v);
        uint32_t packed[4] = {0, 0, 0, 0};
        if (m < rows) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int d = (int)((ab[i] >> sh) & dmask) - (int)half;   // balanced digit
                packed[i >> 2] |= ((uint32_t)d & 255u) << (8 * (i & 3));
            }
        }
        const size_t kappa = (size_t)v * g.D + (size_t)wb * 16;
        const uint32_t ks = (uint32_t)(kappa >> 5), h = (uint32_t)(kappa >> 4) & 1u;
        uint4 *dst = reinterpret_cast<uint4 *>(A + (((size_t)blockIdx.x * g.ksteps + ks) * 64 + h * 32 + r) * 16);
        *dst = make_uint4(packed[0], packed[1], packed[2], packed[3]);
    }
}

// C += A B over k-steps [blockIdx.z * klen, ...): four waves, 2 x 2, each MT x NT tiles of 32 x 32.  Fragments come straight
// from global memory (16 bytes per lane, a kilobyte per wave and fragment, contiguous); both matrices are a few tens of MB
// and live in L2 / Infinity Cache.  Partial sums of the k-slices meet in C through integer atomics.
template <int MT, int NT>
__global__ __launch_bounds__(256) void k_ks_gemm(const v4i *__restrict__ A, const v4i *__restrict__ B, int *__restrict__ C,
                                                 uint32_t ksteps, uint32_t klen, uint32_t ldc) {
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t mt0 = (blockIdx.y * 2 + (wave >> 1)) * MT, nt0 = (blockIdx.x * 2 + (wave & 1)) * NT;
    const uint32_t k0 = blockIdx.z * klen;
    const uint32_t k1 = k0 + klen < ksteps ? k0 + klen : ksteps;
    v16i acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[i][j][e] = 0;
    const v4i *pa[MT], *pb[NT];
#pragma unroll
    for (int i = 0; i < MT; i++) pa[i] = A + ((size_t)(mt0 + i) * ksteps + k0) * 64 + lane;
#pragma unroll
    for (int j = 0; j < NT; j++) pb[j] = B + ((size_t)(nt0 + j) * ksteps + k0) * 64 + lane;
    // a ring of three fragment sets: the loads of k-step k + 2 are in flight while the products of k-step k run (an L2 round
    // trip is several times the 4 x 32 cycles the matrix pipe spends on one step)
    const uint32_t steps = k1 - k0;
    v4i fa[3][MT], fb[3][NT];
    auto fetch = [&](int buf, uint32_t step) {
        const size_t at = (size_t)(step < steps ? step : steps - 1) * 64;   // past the end: the last step again, unused
#pragma unroll
        for (int i = 0; i < MT; i++) fa[buf][i] = pa[i][at];
#pragma unroll
        for (int j = 0; j < NT; j++) fb[buf][j] = pb[j][at];
    };
    auto multiply = [&](int buf) {
#pragma unroll
        for (int i = 0; i < MT; i++)
#pragma unroll
            for (int j = 0; j < NT; j++) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[buf][i], fb[buf][j], acc[i][j], 0, 0, 0);
    };
    fetch(0, 0);
    fetch(1, 1);
    for (uint32_t step = 0; step < steps; step += 3) {
        fetch(2, step + 2);
        multiply(0);
        fetch(0, step + 3);
        if (step + 1 < steps) multiply(1);
        fetch(1, step + 4);
        if (step + 2 < steps) multiply(2);
    }
    // C/D layout of the 32 x 32 forms: column = lane & 31, row = (e & 3) + 8 (e >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < MT; i++)
#pragma unroll
        for (int j = 0; j < NT; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
                const uint32_t row = (mt0 + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const uint32_t col = (nt0 + j) * 32 + (lane & 31);
                atomicAdd(C + (size_t)row * ldc + col, acc[i][j][e]);
            }
}

// limbs -> word mod q -> modulus switch.  A wave takes four ciphertext rows with its lanes on 64 consecutive columns
// (coalesced reads of the six limb planes -- all 24 requested before the first is used -- and coalesced stores of the
// switched words), adds the rounding errors of a row up across the lanes (one atomic per row and wave), and leaves the
// part of C it read zero for the next launch.
__global__ __launch_bounds__(256) void k_ks_gemm_finish(KsArgs a, KsGemm g, size_t f0, size_t rows, int *C, uint32_t ldc) {
    constexpr int ROWS = 4;   // rows per wave
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t col = blockIdx.x * 64 + lane;
    const size_t m0 = ((size_t)blockIdx.y * 4 + wave) * ROWS;
    int limb[ROWS][6];
#pragma unroll
    for (int i = 0; i < ROWS; i++) {
        const size_t m = m0 + i < rows ? m0 + i : rows - 1;
        int *crow = C + m * ldc + col;
#pragma unroll
        for (int b = 0; b < 6; b++) limb[i][b] = col <= g.n ? crow[b * g.cols_pad] : 0;
    }
#pragma unroll
    for (int i = 0; i < ROWS; i++) {
        const size_t m = m0 + i;
        if (m >= rows) return;   // (wave-uniform)
        const size_t f = f0 + m;
        int64_t eps = 0;
        if (col <= g.n) {
            int *crow = C + m * ldc + col;
#pragma unroll
            for (int b = 0; b < 6; b++) crow[b * g.cols_pad] = 0;
            const int64_t lo = (int64_t)limb[i][0] + ((int64_t)limb[i][1] << 8) + ((int64_t)limb[i][2] << 16);   // |.| < 2^48
            const int64_t hi = (int64_t)limb[i][3] + ((int64_t)limb[i][4] << 8) + ((int64_t)limb[i][5] << 16);
            const uint64_t sum = fq_add(fq_from_i64(lo), fq_mul(fq_from_i64(hi), 1ull << 24));
            const uint64_t body = col == g.n ? ks_in(a.gv, f, a.ct_words)[g.D] : 0;
            ms_store(a, f, col, fq_sub(body, sum), &eps);
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) eps += __shfl_xor(eps, d);
        if (lane == 0) ms_flush(a, f, eps);
    }
}

// ---------------------------------------------------------------------------------------------
// linear combination over wire slots: out = sum coef_i * wire_i + const (exact FP64 products, lazy sum)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lincomb(uint64_t *wires, size_t T, size_t s_begin, size_t s_count, uint32_t ct_words,
                                                 const uint32_t *dst, const uint32_t *term_off, const uint32_t *srcs,
                                                 const double *coefs, const uint64_t *consts) {
    const size_t b = blockIdx.x;                 // flattened (output, sample): grid.x reaches 2^31, grid.y only 65535
    const uint32_t g = (uint32_t)(b / s_count);
    const size_t s = s_begin + b % s_count;
    const uint32_t t0 = term_off[g], t1 = term_off[g + 1];
    uint64_t *out = wires + ((size_t)dst[g] * T + s) * ct_words;
    for (uint32_t j = threadIdx.x; j < ct_words; j += 256) {
        double accv = (j == ct_words - 1) ? fp_from_u64(consts[g]) : 0.0;
        for (uint32_t t = t0; t < t1; t++) {
            const double v = fp_from_u64(wires[((size_t)srcs[t] * T + s) * ct_words + j]);
            accv += fp_mulmod(v, coefs[t]);          // each product is below 0.75 q in magnitude
            if (((t - t0) & 15u) == 15u) accv = fp_center(accv);
        }
        out[j] = fp_to_u64(fp_canon(accv));
    }
}

// rows of a contiguous array -> wire slots (the gate-sharded multi-GPU mode publishes a level's bootstraps as one
// contiguous all-gathered array; this puts them where the next level's linear combinations look for them)
__global__ __launch_bounds__(256) void k_scatter_rows(uint64_t *wires, size_t T, size_t s_begin, size_t s_count, uint32_t ct_words,
                                                      const uint32_t *dst_slot, const uint64_t *rows, size_t f_begin, uint32_t row_words) {
    const size_t f = f_begin + blockIdx.x;
    const size_t g = f / s_count, s = s_begin + f % s_count;
    if (dst_slot[g] & 0x80000000u) return;   // a shared rotation's row holds an accumulator: k_multi_extract reads it
    uint64_t *out = wires + ((size_t)dst_slot[g] * T + s) * ct_words;
    const uint64_t *in = rows + (size_t)blockIdx.x * row_words;
    for (uint32_t j = threadIdx.x; j < ct_words; j += 256) out[j] = in[j];
}

// Several tables on one blind rotation (fused programs): the rotation of TV_0 left the GLWE accumulator (A_0 .. A_(k-1), B) in a scratch
// row; table F's ciphertext is SampleExtract_0((A, B) * D_F) + post_F with D_F the small integer polynomial of
// host_build_tv_diff, given as (position, value) pairs.  Coefficient m of X^i * P is P[m - i], negated when it wrapped;
// the extracted ciphertext holds A'_0, -A'_(N-1), .., -A'_1 and B'_0.  d * word < 2^16 * 2^46 summed in 64 bits (the loader
// fuses only tables with sum |d| < 2^16), one reduction per output word.  HBM-bound copy work, a few hundred KB per row.
__global__ __launch_bounds__(256) void k_multi_extract(const uint64_t *acc_rows, uint64_t *wires, size_t T, size_t s_begin,
                                                       size_t s_count, uint32_t N, uint32_t k, const uint32_t *x_row, const uint32_t *x_table,
                                                       const uint32_t *x_dst, const uint32_t *diff_pos, const int32_t *diff_val,
                                                       const uint32_t *diff_n, uint32_t diff_cap, const uint64_t *post) {
    const size_t e = blockIdx.x / s_count, s = blockIdx.x % s_count;
    const uint32_t tab = x_table[e], D = k * N;
    const uint64_t *acc = acc_rows + ((size_t)x_row[e] * s_count + s) * (size_t)(k + 1) * N;   // A_0 .. A_(k-1), B
    uint64_t *out = wires + ((size_t)x_dst[e] * T + s_begin + s) * (D + 1);
    const uint32_t *pos = diff_pos + (size_t)tab * diff_cap;
    const int32_t *val = diff_val + (size_t)tab * diff_cap;
    const uint32_t w = diff_n[tab];
    for (uint32_t j = threadIdx.x; j <= D; j += 256) {
        const uint32_t c = j / N, jj = j - c * N;               // word jj of mask polynomial c; j == D: the body
        const uint64_t *P = acc + (size_t)c * N;                // (c == k for the body: the accumulator's last polynomial)
        const uint32_t m = jj == 0 ? 0u : N - jj;               // coefficient of the product this word comes from
        int64_t sum = 0;
        for (uint32_t i = 0; i < w; i++) {
            const uint32_t at = pos[i];
            const int64_t term = (int64_t)val[i] * (int64_t)(m >= at ? P[m - at] : P[m + N - at]);
            sum += m >= at ? term : -term;
        }
        if (jj != 0) sum = -sum;
        int64_t r = sum % (int64_t)FQ;
        if (r < 0) r += (int64_t)FQ;
        out[j] = j == D ? fq_add((uint64_t)r, post[tab]) : (uint64_t)r;
    }
}

// trivial ciphertexts (0, ..., 0, body): constant outputs of a program
__global__ __launch_bounds__(256) void k_fill_trivial(uint64_t *out, uint32_t ct_words, uint64_t body) {
    uint64_t *row = out + (size_t)blockIdx.x * ct_words;
    for (uint32_t j = threadIdx.x; j < ct_words; j += 256) row[j] = j == ct_words - 1 ? body : 0;
}

// ---------------------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------------------
void prof_begin(fbs_ctx *ctx, int, hipStream_t s, hipEvent_t *e0, hipEvent_t *e1) {
    *e0 = *e1 = nullptr;
    if (!ctx->prof.on) return;
    if (ctx->prof.pool.empty()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        ctx->prof.pool.push_back({a, b});
    }
    auto pr = ctx->prof.pool.back();
    ctx->prof.pool.pop_back();
    *e0 = pr.first;
    *e1 = pr.second;
    (void)hipEventRecord(*e0, s);
}
void prof_end(fbs_ctx *ctx, int which, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    if (!e0) return;
    (void)hipEventRecord(e1, s);
    ctx->prof.pending[which].push_back({e0, e1, ctx->prof.kernel[which]});
}

int dev_supported(const fbs_ctx *ctx) {
    const fbs_params &p = ctx->p;
    if (p.k >= 2 && !glwe_shape_built(p.log_n_poly, p.k))
        return set_error(ctx, FBS_E_INVALID, "GLWE dimensions k >= 2 are built for k = 2, 3, 4 at N = 256 and 512 and k = 2, 3 at N = 1024");
    if (p.k < 1) return set_error(ctx, FBS_E_INVALID, "need k >= 1");
    if (p.log_n_poly < 8 || p.log_n_poly > 12)
        return set_error(ctx, FBS_E_INVALID, "supported polynomial sizes are N = 256, 512, 1024, 2048, 4096");
    if (p.l_bsk < 1 || p.beta_bsk < 1 || p.l_bsk * p.beta_bsk > 30 || p.l_bsk * p.beta_bsk > FQ_BITS - 2)
        return set_error(ctx, FBS_E_INVALID, "need 1 <= l*beta <= 30");
    if (p.t_ksk < 1 || p.gamma_ksk < 1 || p.t_ksk * p.gamma_ksk > 31 || p.t_ksk * p.gamma_ksk > FQ_BITS - 2)
        return set_error(ctx, FBS_E_INVALID, "need 1 <= t*gamma <= 31");
    if (p.n < 1 || p.n > 4096) return set_error(ctx, FBS_E_INVALID, "need 1 <= n <= 4096");
    if (p.bsk_group == 2 && p.k == 1 && (p.log_n_poly < 10 || p.l_bsk > 5))
        return set_error(ctx, FBS_E_INVALID, "two key bits per step (bsk_group = 2) at k = 1 is built for N = 1024, 2048 and 4096, l <= 5");
    // lazy FP64 ranges (fbs_field.hpp): partial external products stay below 2^50 while (k+1)*l <= 20
    if ((p.k + 1) * p.l_bsk > 20) return set_error(ctx, FBS_E_INVALID, "need (k+1)*l <= 20");
    // 64-bit key-switch accumulators: D*t digits < 2^gamma times words < 2^46
    double bits = FQ_BITS + p.gamma_ksk + std::log2((double)p.t_ksk * ctx->D);
    if (bits > 63.9 || bits - 32.0 > 31.9)   // whole sum in 64 bits; the high-word partial sums in 32
        return set_error(ctx, FBS_E_INVALID, "key-switch accumulator would overflow");
    return FBS_OK;
}

static KsGemm gemm_shape(const fbs_ctx *ctx) {
    KsGemm g{};
    g.D = ctx->D;
    g.t = ctx->p.t_ksk;
    g.gamma = ctx->p.gamma_ksk;
    g.n = ctx->p.n;
    g.cols_pad = (ctx->p.n + 1 + 63) / 64 * 64;   // 6 * cols_pad columns: a multiple of the 128 a workgroup covers
    g.ksteps = ctx->D * ctx->p.t_ksk / 32;
    return g;
}

// limb fragments of the key-switching key (after the integer copy d_ksk is in place)
int dev_keyswitch_gemm_setup(fbs_ctx *ctx) {
    const KsGemm g = gemm_shape(ctx);
    const size_t bytes = (size_t)g.ksteps * 32 * 6 * g.cols_pad;
    if (ctx->d_ks_b) (void)hipFree(ctx->d_ks_b);
    ctx->d_ks_b = nullptr;
    FBS_HIP(ctx, hipMalloc(&ctx->d_ks_b, bytes));
    const size_t threads = (size_t)g.D * g.t * g.cols_pad;
    hipLaunchKernelGGL(k_ks_limbs, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, ctx->d_ksk, ctx->ksk_stride, g,
                       ctx->d_ks_b);
    FBS_HIP(ctx, hipGetLastError());
    FBS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return FBS_OK;
}

// digits -> GEMM -> recombination + modulus switch, in passes of at most 8192 ciphertexts (the scratch stays bounded)
constexpr size_t KS_GEMM_PASS = 8192;

static bool keyswitch_gemm_exact(const fbs_ctx *ctx) {
    // int8 GEMM on the matrix cores: exact while 2^(gamma-1) * 2^7 * kN t stays below 2^31
    return ctx->d_ks_b && std::ldexp((double)ctx->D * ctx->p.t_ksk, (int)ctx->p.gamma_ksk + 6) < 2147483648.0;
}

int dev_keyswitch_reserve(fbs_ctx *ctx, size_t count) {
    if (!keyswitch_gemm_exact(ctx)) return FBS_OK;
    const KsGemm g = gemm_shape(ctx);
    const size_t want = std::min<size_t>(KS_GEMM_PASS, (count + 127) / 128 * 128);
    if (ctx->ks_rows_capacity >= want) return FBS_OK;
    const uint32_t ldc = 6 * g.cols_pad;
    ctx->scratch_growths++;
    if (ctx->d_ks_a) (void)hipFree(ctx->d_ks_a);   // (hipFree waits for the kernels that may still use them)
    if (ctx->d_ks_c) (void)hipFree(ctx->d_ks_c);
    ctx->d_ks_a = nullptr;
    ctx->d_ks_c = nullptr;
    ctx->ks_rows_capacity = 0;
    FBS_HIP(ctx, hipMalloc(&ctx->d_ks_a, want * (size_t)g.ksteps * 32));
    FBS_HIP(ctx, hipMalloc(&ctx->d_ks_c, want * (size_t)ldc * 4));
    FBS_HIP(ctx, hipMemset(ctx->d_ks_c, 0, want * (size_t)ldc * 4));   // every launch leaves it zero again
    FBS_HIP(ctx, hipDeviceSynchronize());
    ctx->ks_rows_capacity = want;
    return FBS_OK;
}

int dev_keyswitch_rezero(fbs_ctx *ctx, hipStream_t stream) {
    if (ctx->d_ks_c && ctx->ks_rows_capacity)
        FBS_HIP(ctx, hipMemsetAsync(ctx->d_ks_c, 0, ctx->ks_rows_capacity * (size_t)(6 * gemm_shape(ctx).cols_pad) * 4, stream));
    return FBS_OK;
}

static int keyswitch_gemm(fbs_ctx *ctx, KsArgs &a, hipStream_t stream) {
    const KsGemm g = gemm_shape(ctx);
    constexpr size_t PASS = KS_GEMM_PASS;
    const uint32_t ldc = 6 * g.cols_pad;
    if (int rc = dev_keyswitch_reserve(ctx, a.count)) return rc;   // (a no-op after ensure_ms / fbs_ctx_reserve)
    ctx->prof.kernel[0] = "k_ks_gemm<2,2> (int8 MFMA)";
    for (size_t f0 = 0; f0 < a.count; f0 += PASS) {
        const size_t rows = std::min(PASS, a.count - f0);
        const unsigned m_blocks = (unsigned)((rows + 127) / 128), n_blocks = ldc / 128;
        hipLaunchKernelGGL(k_ks_digits, dim3(m_blocks * 4, (g.D / 16 + 7) / 8), dim3(256), 0, stream, a, g, f0, rows, ctx->d_ks_a);
        // enough workgroups to keep every CU busy: the k range is cut where the (m, n) grid alone is too small.  (2 x 4 tiles
        // per wave halve the fragment traffic but leave one wave per SIMD: 186 against 135 us at N = 2048, t = 7.)
        const unsigned want_split = (2u * (unsigned)ctx->cu_count + m_blocks * n_blocks - 1) / (m_blocks * n_blocks);
        unsigned split = std::max(1u, std::min({want_split, 16u, g.ksteps / 16u}));
        if (ctx->tune.ks_split > 0) split = (unsigned)ctx->tune.ks_split;   // (tuning)
        const uint32_t klen = (g.ksteps + split - 1) / split;
        hipLaunchKernelGGL((k_ks_gemm<2, 2>), dim3(n_blocks, m_blocks, (g.ksteps + klen - 1) / klen), dim3(256), 0, stream,
                           reinterpret_cast<const v4i *>(ctx->d_ks_a), reinterpret_cast<const v4i *>(ctx->d_ks_b), ctx->d_ks_c, g.ksteps, klen,
                           ldc);
        hipLaunchKernelGGL(k_ks_gemm_finish, dim3((g.n + 64) / 64, (unsigned)((rows + 15) / 16)), dim3(256), 0, stream, a, g, f0, rows,
                           ctx->d_ks_c, ldc);
    }
    return FBS_OK;
}

void keyswitch_catalog(std::vector<std::string> *out) {
    for (const char *k : {"k_ks_gemm<2,2> (int8 MFMA)", "k_keyswitch_fp<8,2,8>", "k_keyswitch_lanes<8,2,8>", "k_keyswitch_lanes<8,1,4>", "k_keyswitch<8>"})
        out->push_back(k);
}

int dev_keyswitch(fbs_ctx *ctx, const GateView &gv, uint32_t *d_ms, hipStream_t stream) {
    const fbs_params &p = ctx->p;
    KsArgs a{};
    a.gv = gv;
    a.ksk = ctx->d_ksk;
    a.corr = ctx->d_ks_corr;
    a.ms = d_ms;
    a.eps = ctx->d_ms_eps;
    a.body_raw = ctx->d_ms_body;
    for (uint32_t v = 0; v < p.t_ksk; v++) a.offs |= (1u << (p.gamma_ksk - 1)) << (v * p.gamma_ksk);
    a.n = p.n;
    a.D = ctx->D;
    a.t = p.t_ksk;
    a.gamma = p.gamma_ksk;
    a.stride = ctx->ksk_stride;
    a.ct_words = ctx->D + 1;
    a.log2_2n = p.log_n_poly + 1;
    a.count = gv.ks_count;
    if (a.count == 0) return FBS_OK;
    hipEvent_t e0, e1;
    prof_begin(ctx, 0, stream, &e0, &e1);
    // The int8 GEMM on the matrix cores serves every batch size (round 2 used it above 64 ciphertexts only; below, the integer
    // kernels took 0.54 ms for 32-64 ciphertexts and 2.7 ms for 1-16, the GEMM takes 0.04-0.06 ms: its cost is streaming the
    // key's 31 MB of limb fragments, whatever the number of rows).
    const size_t gemm_min = (size_t)ctx->tune.ks_gemm_min;
    const bool gemm_ok = ctx->tune.ks_mfma && keyswitch_gemm_exact(ctx);
    if (a.count >= gemm_min && gemm_ok) {
        const int rc = keyswitch_gemm(ctx, a, stream);
        if (rc != FBS_OK) return rc;
    } else if (a.count >= 32) {
        // lanes = ciphertexts: pays once a wave is at least half full
        constexpr int COLS = 8;
        // measured per 1024-batch: 64 ciphertexts x 4 waves 0.91 ms, 128 x 4 waves 0.81, 64 x 8 waves 1.10, 128 x 8 waves 0.65,
        // 256 x 8 waves 1.32, 128 x 16 waves 0.86, 256 x 16 waves 1.80
        // Workgroups are dealt to the 8 XCDs round-robin by linear id.  With the column blocks on grid.x, padded to a
        // multiple of 16 and dealt in adjacent pairs, XCD x only ever sees column blocks 2x, 2x+1 (mod 16): each key line is fetched by ONE XCD's L2
        // instead of all eight (the 50 MB key does not fit any L2), and what every XCD re-reads is the 8 times
        // smaller ciphertext batch.
        const bool cols_major = ctx->tune.ks_cols_major != 0;
        a.cols_major = cols_major ? 1u : 0u;
        const unsigned cols = (p.n + 1 + COLS - 1) / COLS;
        const unsigned cols_padded = cols_major ? (cols + 15u) / 16u * 16u : cols;
        // FP64 form: needs the centred-double copy of the key and room to accumulate at least one mask word exactly
        const double per_word = (double)p.t_ksk * std::ldexp(1.0, 44 + (int)p.gamma_ksk);
        const double room = std::ldexp(1.0, 53) - std::ldexp(1.0, 45);
        const bool allow_fp = ctx->tune.ks_fp != 0;
        if (a.count > 64 && cols_major && allow_fp && ctx->d_ksk_f && per_word <= room) {
            const uint32_t words_per_fold = (uint32_t)std::min(1024.0, std::floor(room / per_word));
            const size_t tiles = (a.count + 127) / 128;
            for (size_t t0 = 0; t0 < tiles; t0 += 65535) {
                const unsigned nt = (unsigned)std::min<size_t>(65535, tiles - t0);
                a.row0 = t0 * 128;
                // measured per 1024-batch: 8 columns x 128 ciphertexts x 8 waves 0.48 ms; 16 columns 0.77 (8 waves) / 0.67 (4 waves);
                // 8 columns x 4 waves 0.60; 256 ciphertexts 0.57
                ctx->prof.kernel[0] = "k_keyswitch_fp<8,2,8>";
                hipLaunchKernelGGL((k_keyswitch_fp<COLS, 2, 8>), dim3(cols_padded, nt), dim3(512), 0, stream, a,
                                   reinterpret_cast<const double *>(ctx->d_ksk_f), words_per_fold);
            }
        } else if (a.count > 64) {
            const size_t tiles = (a.count + 127) / 128;
            const size_t per_launch = cols_major ? 65535 : 0x7FFFFFFF;     // tiles sit on grid.y in the column-major form
            for (size_t t0 = 0; t0 < tiles; t0 += per_launch) {
                const unsigned nt = (unsigned)std::min(per_launch, tiles - t0);
                a.row0 = t0 * 128;
                ctx->prof.kernel[0] = "k_keyswitch_lanes<8,2,8>";
                hipLaunchKernelGGL((k_keyswitch_lanes<COLS, 2, 8>), cols_major ? dim3(cols_padded, nt) : dim3(nt, cols), dim3(512), 0,
                                   stream, a);
            }
        } else {
            ctx->prof.kernel[0] = "k_keyswitch_lanes<8,1,4>";
            hipLaunchKernelGGL((k_keyswitch_lanes<COLS, 1, 4>), cols_major ? dim3(cols_padded, 1) : dim3(1, cols), dim3(256), 0, stream, a);
        }
    } else {
        constexpr int FB = 8;
        dim3 grid((unsigned)((a.count + FB - 1) / FB), ctx->ksk_stride / 256);
        size_t shmem = (size_t)FB * ctx->D * sizeof(uint32_t);
        ctx->prof.kernel[0] = "k_keyswitch<8>";
        hipLaunchKernelGGL(k_keyswitch<FB>, grid, dim3(256), shmem, stream, a);
    }
    hipLaunchKernelGGL(k_ms_body, dim3((unsigned)((a.count + 255) / 256)), dim3(256), 0, stream, d_ms, a.eps, a.body_raw, p.n,
                       a.log2_2n, a.count);
    prof_end(ctx, 0, stream, e0, e1);
    FBS_HIP(ctx, hipGetLastError());
    return FBS_OK;
}

int dev_lincomb(fbs_ctx *ctx, uint64_t *d_wires, size_t T, size_t s_begin, size_t s_count, uint32_t n_out, const uint32_t *d_dst,
                const uint32_t *d_term_off, const uint32_t *d_srcs, const uint64_t *d_coefs, const uint64_t *d_consts,
                hipStream_t stream) {
    if (n_out == 0 || s_count == 0) return FBS_OK;
    const size_t blocks = (size_t)n_out * s_count;
    if (blocks > 0x7FFFFFFFull) return set_error(ctx, FBS_E_INVALID, "more than 2^31 linear combinations in one launch");
    hipEvent_t e0, e1;
    prof_begin(ctx, 2, stream, &e0, &e1);
    // d_coefs carries the coefficients as centred doubles (bit pattern in a uint64 array, see fbs_capi.cpp)
    ctx->prof.kernel[2] = "k_lincomb";
    hipLaunchKernelGGL(k_lincomb, dim3((unsigned)blocks), dim3(256), 0, stream, d_wires, T, s_begin, s_count, ctx->D + 1, d_dst,
                       d_term_off, d_srcs, reinterpret_cast<const double *>(d_coefs), d_consts);
    prof_end(ctx, 2, stream, e0, e1);
    FBS_HIP(ctx, hipGetLastError());
    return FBS_OK;
}

int dev_scatter_rows(fbs_ctx *ctx, uint64_t *d_wires, size_t T, size_t s_begin, size_t s_count, const uint32_t *d_dst_slot,
                     const uint64_t *d_rows, size_t f_begin, size_t count, uint32_t row_words, hipStream_t stream) {
    if (count == 0) return FBS_OK;
    if (count > 0x7FFFFFFFull) return set_error(ctx, FBS_E_INVALID, "more than 2^31 rows in one launch");
    hipLaunchKernelGGL(k_scatter_rows, dim3((unsigned)count), dim3(256), 0, stream, d_wires, T, s_begin, s_count, ctx->D + 1, d_dst_slot,
                       d_rows, f_begin, row_words ? row_words : ctx->D + 1);
    FBS_HIP(ctx, hipGetLastError());
    return FBS_OK;
}

int dev_multi_extract(fbs_ctx *ctx, const fbs_tvset *tv, const uint64_t *d_acc_rows, uint64_t *d_wires, size_t T, size_t s_begin,
                      size_t s_count, uint32_t n_extract, const uint32_t *d_x_row, const uint32_t *d_x_table,
                      const uint32_t *d_x_dst, hipStream_t stream) {
    const size_t blocks = (size_t)n_extract * s_count;
    if (blocks == 0) return FBS_OK;
    if (blocks > 0x7FFFFFFFull) return set_error(ctx, FBS_E_INVALID, "more than 2^31 rows in one launch");
    hipLaunchKernelGGL(k_multi_extract, dim3((unsigned)blocks), dim3(256), 0, stream, d_acc_rows, d_wires, T, s_begin, s_count, ctx->N, ctx->p.k,
                       d_x_row, d_x_table, d_x_dst, tv->d_diff_pos, tv->d_diff_val, tv->d_diff_n, tv->diff_cap, tv->d_post);
    FBS_HIP(ctx, hipGetLastError());
    return FBS_OK;
}

int dev_copy_out(fbs_ctx *ctx, const uint64_t *d_wires, size_t T, size_t s_begin, size_t count, int64_t slot, uint64_t body,
                 uint64_t *d_out, hipStream_t stream) {
    if (count == 0) return FBS_OK;
    const size_t ctw = ctx->D + 1;
    if (slot >= 0) {
        FBS_HIP(ctx, hipMemcpyAsync(d_out, d_wires + ((size_t)slot * T + s_begin) * ctw, count * ctw * 8, hipMemcpyDeviceToDevice, stream));
        return FBS_OK;
    }
    if (count > 0x7FFFFFFFull) return set_error(ctx, FBS_E_INVALID, "more than 2^31 rows in one launch");
    hipLaunchKernelGGL(k_fill_trivial, dim3((unsigned)count), dim3(256), 0, stream, d_out, (uint32_t)ctw, body);
    FBS_HIP(ctx, hipGetLastError());
    return FBS_OK;
}

}  // namespace fbs

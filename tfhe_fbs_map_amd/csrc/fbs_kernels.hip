// HIP kernels of libfbsexec.so, written for gfx950 (MI355X, CDNA4) only.
//
//   k_bsk_transform   one-time: bootstrapping key -> NTT domain, lane-interleaved, scaled by 1/N
//   k_keyswitch       LWE key switch kN -> n fused with the modulus switch q -> 2N
//   k_blind_rotate    the hot kernel: n CMUX steps (rotate, gadget-decompose, (k+1)l forward NTTs,
//                     multiply-accumulate against one bootstrapping-key row, k+1 inverse NTTs),
//                     then sample extraction and the table's post-add
//   k_lincomb         LinearProd over wire slots
//   k_polymul         debug: one negacyclic product through the same NTT code
//
// Mapping to the machine.  Everything here is 64-bit modular integer arithmetic: no MFMA.  One
// functional bootstrap is one workgroup of two wavefronts; wave c owns GLWE component c of the
// accumulator (k = 1: mask and body).  A polynomial of N coefficients is spread over the 64 lanes
// of its wave, E = N/64 coefficients per lane held in VGPRs; a radix-2 NTT runs log2(E) butterfly
// stages on registers, then exchanges through the wave's own N-word LDS buffer to bring the next
// group of index bits into the lane.  Waves only meet (s_barrier) twice per CMUX step, to hand
// each other the half of the external product that belongs to the other component.  The
// bootstrapping key row of a step is read once per workgroup with 16-byte coalesced loads; all
// workgroups walk the key in the same order, so after the first touch it is served from L2/MALL.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>

#include "fbs_internal.hpp"

namespace fbs {

// ---------------------------------------------------------------------------------------------
// device NTT over one wave
// ---------------------------------------------------------------------------------------------
template <int LOGN>
struct WaveNtt {
    static constexpr int N = 1 << LOGN;
    static constexpr int E = N / 64;
    static constexpr int LOGE = LOGN - 6;
    static constexpr int GROUPS = (LOGN + LOGE - 1) / LOGE;
    static_assert(LOGN >= 8 && LOGN <= 11, "supported polynomial sizes: 256..2048");

    // lowest index bit held inside the lane during group g
    __device__ static constexpr int lo_of(int g) { return (LOGN - (g + 1) * LOGE) > 0 ? (LOGN - (g + 1) * LOGE) : 0; }

    // LDS word of coefficient j: XOR-fold of index bits 4..8 into the bank-selecting bits 0..4 keeps
    // the three access patterns (lane bits = j[5:0], j[9:6|1:0], j[9:4]) conflict-free for b64.
    __device__ static __forceinline__ uint32_t phys(uint32_t j) { return j ^ ((j >> 4) & 31u); }

    // coefficient index of register m of `lane` during group g
    template <int G>
    __device__ static __forceinline__ uint32_t index_of(uint32_t lane, int m) {
        constexpr int lo = lo_of(G);
        return ((lane >> lo) << (lo + LOGE)) | ((uint32_t)m << lo) | (lane & ((1u << lo) - 1u));
    }

    template <int G>
    __device__ static __forceinline__ void store_group(uint64_t *buf, uint32_t lane, const uint64_t (&x)[E]) {
        const uint32_t base = phys(index_of<G>(lane, 0));
#pragma unroll
        for (int m = 0; m < E; m++) buf[base ^ phys(index_of<G>(0, m))] = x[m];   // phys is XOR-linear
    }
    template <int G>
    __device__ static __forceinline__ void load_group(const uint64_t *buf, uint32_t lane, uint64_t (&x)[E]) {
        const uint32_t base = phys(index_of<G>(lane, 0));
#pragma unroll
        for (int m = 0; m < E; m++) x[m] = buf[base ^ phys(index_of<G>(0, m))];
    }

    __device__ static __forceinline__ void wave_sync() {
        // LDS operations of one wave complete in issue order; only the compiler must be held back.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // Cooley-Tukey stages of group G on registers.  x loose in, loose out.
    template <int G>
    __device__ static __forceinline__ void fwd_group(uint64_t (&x)[E], uint32_t lane, const uint64_t *__restrict__ tw) {
        constexpr int lo = lo_of(G);
        constexpr int s_begin = G * LOGE;
        constexpr int s_end = (G + 1) * LOGE < LOGN ? (G + 1) * LOGE : LOGN;
        const uint32_t hi_part = lane >> lo;
#pragma unroll
        for (int s = s_begin; s < s_end; s++) {
            const int bit = LOGN - 1 - s - lo;   // register-index bit paired by this stage
            const int hm = 1 << bit;
            const int sh = lo + LOGE - LOGN + s;  // how far the lane's high part reaches into the block id
#pragma unroll
            for (int m = 0; m < E; m++) {
                if (m & hm) continue;
                const uint32_t blk = (hi_part << sh) | (uint32_t)(m >> (bit + 1));
                const uint64_t w = tw[(1u << s) + blk];
                const uint64_t u = x[m];
                const uint64_t v = gl_mul(x[m + hm], w);
                x[m] = gl_add_lc(u, v);
                x[m + hm] = gl_sub_lc(u, v);
            }
        }
    }
    // Gentleman-Sande stages of group G, last stage first.  x canonical in, canonical out.
    template <int G>
    __device__ static __forceinline__ void inv_group(uint64_t (&x)[E], uint32_t lane, const uint64_t *__restrict__ tw) {
        constexpr int lo = lo_of(G);
        constexpr int s_begin = G * LOGE;
        constexpr int s_end = (G + 1) * LOGE < LOGN ? (G + 1) * LOGE : LOGN;
        const uint32_t hi_part = lane >> lo;
#pragma unroll
        for (int s = s_end - 1; s >= s_begin; s--) {
            const int bit = LOGN - 1 - s - lo;
            const int hm = 1 << bit;
            const int sh = lo + LOGE - LOGN + s;
#pragma unroll
            for (int m = 0; m < E; m++) {
                if (m & hm) continue;
                const uint32_t blk = (hi_part << sh) | (uint32_t)(m >> (bit + 1));
                const uint64_t w = tw[(1u << s) + blk];
                const uint64_t u = x[m], v = x[m + hm];
                x[m] = gl_add(u, v);
                x[m + hm] = gl_mul(gl_sub(u, v), w);
            }
        }
    }

    template <int G>
    __device__ static __forceinline__ void fwd_from(uint64_t (&x)[E], uint64_t *buf, uint32_t lane, const uint64_t *tw) {
        fwd_group<G>(x, lane, tw);
        if constexpr (G + 1 < GROUPS) {
            wave_sync();
            store_group<G>(buf, lane, x);
            wave_sync();
            load_group<G + 1>(buf, lane, x);
            fwd_from<G + 1>(x, buf, lane, tw);
        }
    }
    template <int G>
    __device__ static __forceinline__ void inv_from(uint64_t (&x)[E], uint64_t *buf, uint32_t lane, const uint64_t *tw) {
        inv_group<G>(x, lane, tw);
        if constexpr (G > 0) {
            wave_sync();
            store_group<G>(buf, lane, x);
            wave_sync();
            load_group<G - 1>(buf, lane, x);
            inv_from<G - 1>(x, buf, lane, tw);
        }
    }

    // coefficients (group-0 layout, canonical) -> evaluations (last-group layout, loose)
    __device__ static __forceinline__ void forward(uint64_t (&x)[E], uint64_t *buf, uint32_t lane, const uint64_t *tw) {
        fwd_from<0>(x, buf, lane, tw);
    }
    // evaluations (last-group layout, canonical) -> N * coefficients (group-0 layout, canonical)
    __device__ static __forceinline__ void inverse(uint64_t (&x)[E], uint64_t *buf, uint32_t lane, const uint64_t *tw) {
        inv_from<GROUPS - 1>(x, buf, lane, tw);
    }

    // Key storage: evaluation held in register m of `lane` after forward() sits at word
    // ((m/2)*64 + lane)*2 + (m&1) of its polynomial: a wave reads a polynomial with E/2 fully
    // coalesced 16-byte loads.
    __device__ static __forceinline__ uint32_t key_word(uint32_t lane, int m) { return (((uint32_t)(m >> 1) * 64u + lane) << 1) | (uint32_t)(m & 1); }
};

// ---------------------------------------------------------------------------------------------
// one-time key transform
// ---------------------------------------------------------------------------------------------
template <int LOGN>
__global__ __launch_bounds__(64) void k_bsk_transform(const uint64_t *__restrict__ src, uint64_t *__restrict__ dst,
                                                      const uint64_t *__restrict__ tw_fwd, uint64_t n_inv, size_t polys) {
    using W = WaveNtt<LOGN>;
    __shared__ uint64_t buf[W::N];
    const uint32_t lane = threadIdx.x;
    for (size_t p = blockIdx.x; p < polys; p += gridDim.x) {
        uint64_t x[W::E];
#pragma unroll
        for (int m = 0; m < W::E; m++) x[m] = src[p * W::N + W::template index_of<0>(lane, m)];
        W::forward(x, buf, lane, tw_fwd);
        W::wave_sync();
#pragma unroll
        for (int m = 0; m < W::E; m++) dst[p * W::N + W::key_word(lane, m)] = gl_mul(gl_canon(x[m]), n_inv);
    }
}

template <int LOGN>
__global__ __launch_bounds__(64) void k_polymul(const uint64_t *a, const uint64_t *b, uint64_t *c,
                                                const uint64_t *tw_fwd, const uint64_t *tw_inv, uint64_t n_inv) {
    using W = WaveNtt<LOGN>;
    __shared__ uint64_t buf[W::N];
    const uint32_t lane = threadIdx.x;
    uint64_t x[W::E], y[W::E];
#pragma unroll
    for (int m = 0; m < W::E; m++) {
        x[m] = a[W::template index_of<0>(lane, m)];
        y[m] = b[W::template index_of<0>(lane, m)];
    }
    W::forward(x, buf, lane, tw_fwd);
    W::wave_sync();
    W::forward(y, buf, lane, tw_fwd);
    W::wave_sync();
#pragma unroll
    for (int m = 0; m < W::E; m++) x[m] = gl_mul(gl_mul(x[m], gl_canon(y[m])), n_inv);
    W::inverse(x, buf, lane, tw_inv);
#pragma unroll
    for (int m = 0; m < W::E; m++) c[W::template index_of<0>(lane, m)] = x[m];
}

// ---------------------------------------------------------------------------------------------
// key switch + modulus switch
// ---------------------------------------------------------------------------------------------
struct KsArgs {
    GateView gv;
    const uint64_t *ksk;   // [D*t][stride]
    uint32_t *ms;          // [count][n+1]
    uint32_t n, D, t, gamma, stride, ct_words, log2_2n;
    size_t count;          // n_gates * s_count
};

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

// out = (0, b) - sum_j sum_v digit_v(a_j) * KSK[j][v]; unsigned base-2^gamma digits of the closest
// multiple of q/2^(t*gamma).  Workgroup = FB ciphertexts x 256 output columns; the key row is read
// once and used FB times; 96-bit accumulators are folded mod q once at the end.
template <int FB>
__global__ __launch_bounds__(256) void k_keyswitch(KsArgs a) {
    extern __shared__ uint32_t abar[];   // [FB][D]
    const size_t f0 = (size_t)blockIdx.x * FB;
    const uint32_t col = blockIdx.y * 256u + threadIdx.x;
    const uint32_t tg = a.t * a.gamma;

    for (uint32_t idx = threadIdx.x; idx < FB * a.D; idx += 256) {
        uint32_t f = idx / a.D, j = idx % a.D;
        uint32_t v = 0;
        if (f0 + f < a.count) {
            uint64_t w = gate_in(a.gv, f0 + f, a.ct_words)[j];
            v = (uint32_t)(((w >> (63 - tg)) + 1) >> 1);
        }
        abar[idx] = v;
    }
    __syncthreads();

    uint32_t acc0[FB], acc1[FB], acc2[FB];
#pragma unroll
    for (int f = 0; f < FB; f++) acc0[f] = acc1[f] = acc2[f] = 0;
    const uint32_t dmask = (1u << a.gamma) - 1u;
    const uint64_t *kcol = a.ksk + col;   // stride is padded to a multiple of 256: always in bounds

    for (uint32_t j = 0; j < a.D; j++) {
        uint32_t ab[FB];
#pragma unroll
        for (int f = 0; f < FB; f++) ab[f] = abar[f * a.D + j];
        for (uint32_t v = 0; v < a.t; v++) {
            const uint64_t kw = kcol[((size_t)j * a.t + v) * a.stride];
            const uint32_t k0 = (uint32_t)kw, k1 = (uint32_t)(kw >> 32);
            const uint32_t sh = a.gamma * (a.t - 1 - v);
#pragma unroll
            for (int f = 0; f < FB; f++) {
                const uint32_t d = (ab[f] >> sh) & dmask;
                // (acc2:acc1:acc0) += d * (k1:k0)
                uint64_t lo = (uint64_t)d * k0 + acc0[f];
                acc0[f] = (uint32_t)lo;
                uint64_t mid = (uint64_t)d * k1 + acc1[f] + (lo >> 32);
                acc1[f] = (uint32_t)mid;
                acc2[f] += (uint32_t)(mid >> 32);
            }
        }
    }
    if (col > a.n) return;
    const uint32_t sh_ms = 64 - a.log2_2n - 1;
    const uint32_t mask = (1u << a.log2_2n) - 1u;
#pragma unroll
    for (int f = 0; f < FB; f++) {
        if (f0 + f >= a.count) break;
        uint64_t sum = gl_canon(gl_reduce128(((uint64_t)acc1[f] << 32) | acc0[f], acc2[f]));
        uint64_t body = col == a.n ? gate_in(a.gv, f0 + f, a.ct_words)[a.D] : 0;
        uint64_t r = gl_sub(body, sum);
        a.ms[(f0 + f) * (a.n + 1) + col] = (uint32_t)(((r >> sh_ms) + 1) >> 1) & mask;
    }
}

// Same arithmetic, mapped the other way round: LANES are ciphertexts (64 per workgroup) and the key words a
// wave needs are wave-uniform, so they arrive through the scalar cache (s_load_dwordx16) instead of being
// re-fetched by every ciphertext tile: KSK traffic drops from (count/8) x 50 MB to (count/64) x 41 MB per
// launch.  A workgroup owns COLS output columns; its four waves each take a quarter of the kN mask words
// (more waves in flight to cover the scalar-load latency) and their 96-bit partial sums meet in LDS.
// The rounded mask words are staged per tile in LDS, transposed on the way in so that both the global read
// (along the ciphertext row) and the LDS read (along the ciphertexts) are contiguous.
template <int COLS>
__global__ __launch_bounds__(256) void k_keyswitch_lanes(KsArgs a) {
    constexpr int JT = 32;                       // mask words per wave per staging round
    __shared__ uint32_t tile[4 * JT * 64];       // [slice][j in tile][ciphertext]; reused for the final reduction
    static_assert(4 * JT * 64 >= 3 * COLS * 3 * 64, "reduction scratch must fit in the staging tile");
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63;
    const size_t f0 = (size_t)blockIdx.x * 64;
    const size_t f = f0 + lane;
    const uint32_t col0 = blockIdx.y * COLS;
    const uint32_t tg = a.t * a.gamma;
    const uint32_t dmask = (1u << a.gamma) - 1u;
    const uint32_t slice_len = (a.D + 3) / 4;    // mask words per wave

    uint32_t acc0[COLS], acc1[COLS], acc2[COLS];
#pragma unroll
    for (int c = 0; c < COLS; c++) acc0[c] = acc1[c] = acc2[c] = 0;

    for (uint32_t r0 = 0; r0 < slice_len; r0 += JT) {
        __syncthreads();
        // 256 threads stage 4 slices x JT words x 64 ciphertexts: consecutive threads read consecutive words
        for (uint32_t idx = threadIdx.x; idx < 4 * JT * 64; idx += 256) {
            const uint32_t jj = idx % JT, q = (idx / JT) % 64, sl = idx / (JT * 64);
            const uint32_t j = sl * slice_len + r0 + jj;
            uint32_t v = 0;
            if (f0 + q < a.count && r0 + jj < slice_len && j < a.D) {
                const uint64_t w = gate_in(a.gv, f0 + q, a.ct_words)[j];
                v = (uint32_t)(((w >> (63 - tg)) + 1) >> 1);
            }
            tile[(sl * JT + jj) * 64 + q] = v;
        }
        __syncthreads();
        for (uint32_t jj = 0; jj < JT; jj++) {
            const uint32_t j = wave * slice_len + r0 + jj;
            if (r0 + jj >= slice_len || j >= a.D) break;          // wave-uniform
            const uint32_t ab = tile[(wave * JT + jj) * 64 + lane];
            const uint64_t *krow = a.ksk + (size_t)j * a.t * a.stride + col0;   // wave-uniform address
            for (uint32_t v = 0; v < a.t; v++) {
                const uint32_t d = (ab >> (a.gamma * (a.t - 1 - v))) & dmask;
#pragma unroll
                for (int c = 0; c < COLS; c++) {
                    const uint64_t kw = krow[(size_t)v * a.stride + c];
                    const uint64_t lo = (uint64_t)d * (uint32_t)kw + acc0[c];
                    acc0[c] = (uint32_t)lo;
                    const uint64_t mid = (uint64_t)d * (uint32_t)(kw >> 32) + acc1[c] + (lo >> 32);
                    acc1[c] = (uint32_t)mid;
                    acc2[c] += (uint32_t)(mid >> 32);
                }
            }
        }
    }
    // waves 1..3 park their partial sums; wave 0 adds them up (96-bit), folds mod q and mod-switches
    __syncthreads();
    if (wave) {
#pragma unroll
        for (int c = 0; c < COLS; c++) {
            uint32_t *slot = tile + (((wave - 1) * COLS + c) * 3) * 64 + lane;
            slot[0] = acc0[c];
            slot[64] = acc1[c];
            slot[128] = acc2[c];
        }
    }
    __syncthreads();
    if (wave || f >= a.count) return;
    const uint32_t sh_ms = 64 - a.log2_2n - 1;
    const uint32_t mask = (1u << a.log2_2n) - 1u;
    const uint64_t body = gate_in(a.gv, f, a.ct_words)[a.D];
#pragma unroll
    for (int c = 0; c < COLS; c++) {
        const uint32_t col = col0 + c;
        if (col > a.n) break;
        uint64_t lo = ((uint64_t)acc1[c] << 32) | acc0[c];
        uint32_t hi = acc2[c];
#pragma unroll
        for (int w = 0; w < 3; w++) {
            const uint32_t *slot = tile + ((w * COLS + c) * 3) * 64 + lane;
            const uint64_t plo = ((uint64_t)slot[64] << 32) | slot[0];
            const uint64_t s = lo + plo;
            hi += slot[128] + (s < lo ? 1u : 0u);
            lo = s;
        }
        const uint64_t sum = gl_canon(gl_reduce128(lo, hi));
        const uint64_t r = gl_sub(col == a.n ? body : 0, sum);
        a.ms[f * (a.n + 1) + col] = (uint32_t)(((r >> sh_ms) + 1) >> 1) & mask;
    }
}

// ---------------------------------------------------------------------------------------------
// blind rotation + sample extraction
// ---------------------------------------------------------------------------------------------
struct BrArgs {
    GateView gv;
    const uint32_t *ms;        // [count][n+1], values in [0, 2N)
    const uint64_t *bsk_hat;   // [n][rows][2][N]
    const uint64_t *tw_fwd, *tw_inv;
    const uint64_t *tvs;       // [tables][N]
    const uint64_t *post;      // [tables]
    uint32_t n, l, beta, ct_words;
};

template <int LOGN>
__global__ __launch_bounds__(128) void k_blind_rotate(BrArgs a) {
    using W = WaveNtt<LOGN>;
    constexpr int N = W::N, E = W::E;
    __shared__ uint64_t lds[2 * N];
    const uint32_t wave = threadIdx.x >> 6;   // GLWE component owned by this wave: 0 = mask, 1 = body
    const uint32_t lane = threadIdx.x & 63;
    uint64_t *mine = lds + wave * N;
    uint64_t *theirs = lds + (wave ^ 1u) * N;

    const size_t f = blockIdx.x;
    const size_t gate = f / a.gv.s_count;
    const uint32_t table = a.gv.table_ids ? a.gv.table_ids[gate] : 0;
    const uint32_t *ms = a.ms + f * (a.n + 1);
    const uint64_t *tv = a.tvs + (size_t)table * N;
    const uint32_t rows = 2 * a.l;

    // ACC = (0, X^{-b~} * TV); register m of a lane is coefficient lane + 64 m
    uint64_t acc[E];
    {
        const uint32_t r = (2u * N - ms[a.n]) & (2u * N - 1u);
#pragma unroll
        for (int m = 0; m < E; m++) {
            const uint32_t idx = (lane + 64u * m - r) & (2u * N - 1u);
            const uint64_t v = tv[idx & (N - 1)];
            acc[m] = wave ? ((idx & N) ? gl_neg(v) : v) : 0;
        }
    }

    for (uint32_t i = 0; i < a.n; i++) {
        const uint32_t r = __builtin_amdgcn_readfirstlane(ms[i]);
        if (r == 0) continue;   // X^0 * ACC - ACC = 0: nothing to add (uniform over the workgroup)

        // ---- (X^r - 1) * ACC_c, rounded to the closest multiple of q / B^l ----------------------
        uint32_t abar[E];
        W::wave_sync();
        W::template store_group<0>(mine, lane, acc);
        W::wave_sync();
#pragma unroll
        for (int m = 0; m < E; m++) {
            const uint32_t idx = (lane + 64u * m - r) & (2u * N - 1u);
            const uint64_t v = mine[W::phys(idx & (N - 1))];
            const uint64_t rot = (idx & N) ? gl_neg(v) : v;
            const uint64_t d = gl_sub(rot, acc[m]);
            abar[m] = (uint32_t)(((d >> (63 - a.l * a.beta)) + 1) >> 1);
        }

        // ---- digits, least significant level first; NTT; multiply-accumulate with the key row ---
        uint64_t own[E], other[E];   // contributions to component `wave` and to the partner's
#pragma unroll
        for (int m = 0; m < E; m++) own[m] = other[m] = 0;
        const uint32_t bmask = (1u << a.beta) - 1u, bhalf = 1u << (a.beta - 1);
        for (int lv = (int)a.l - 1; lv >= 0; lv--) {
            uint64_t x[E];
#pragma unroll
            for (int m = 0; m < E; m++) {
                uint32_t dg = abar[m] & bmask;
                const uint32_t carry = dg >= bhalf ? 1u : 0u;
                abar[m] = (abar[m] >> a.beta) + carry;
                // balanced digit dg - carry*B as a field element
                x[m] = carry ? GQ - (uint64_t)((1u << a.beta) - dg) : (uint64_t)dg;
            }
            W::forward(x, mine, lane, a.tw_fwd);
            const uint64_t *krow = a.bsk_hat + (((size_t)i * rows + wave * a.l + lv) * 2) * N;
            const ulonglong2 *k_own = reinterpret_cast<const ulonglong2 *>(krow + (size_t)wave * N);
            const ulonglong2 *k_oth = reinterpret_cast<const ulonglong2 *>(krow + (size_t)(wave ^ 1u) * N);
#pragma unroll
            for (int m = 0; m < E; m += 2) {
                const ulonglong2 ko = k_own[(m >> 1) * 64 + lane];
                const ulonglong2 kt = k_oth[(m >> 1) * 64 + lane];
                own[m] = gl_add_lc(own[m], gl_mul(x[m], ko.x));
                own[m + 1] = gl_add_lc(own[m + 1], gl_mul(x[m + 1], ko.y));
                other[m] = gl_add_lc(other[m], gl_mul(x[m], kt.x));
                other[m + 1] = gl_add_lc(other[m + 1], gl_mul(x[m + 1], kt.y));
            }
        }

        // ---- hand the partner its half of the external product --------------------------------
        __syncthreads();   // partner is done with its buffer
#pragma unroll
        for (int m = 0; m < E; m++) theirs[m * 64 + lane] = gl_canon(other[m]);
        __syncthreads();
#pragma unroll
        for (int m = 0; m < E; m++) own[m] = gl_canon(gl_add_lc(own[m], mine[m * 64 + lane]));

        // ---- back to coefficients (the 1/N is folded into the key) and accumulate ---------------
        W::wave_sync();
        W::inverse(own, mine, lane, a.tw_inv);
#pragma unroll
        for (int m = 0; m < E; m++) acc[m] = gl_add(acc[m], own[m]);
    }

    // ---- sample extraction of coefficient 0, plus the table's constant -----------------------------
    uint64_t *out = gate_out(a.gv, f, a.ct_words);
    if (wave == 0) {
#pragma unroll
        for (int m = 0; m < E; m++) {
            const uint32_t j = lane + 64u * m;
            if (j == 0) out[0] = acc[m];
            else out[N - j] = gl_neg(acc[m]);
        }
    } else if (lane == 0) {
        out[N] = gl_add(acc[0], a.post[table]);
    }
}

// ---------------------------------------------------------------------------------------------
// linear combination over wire slots
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lincomb(uint64_t *wires, size_t T, uint32_t ct_words, const uint32_t *dst,
                                                 const uint32_t *term_off, const uint32_t *srcs,
                                                 const uint64_t *coefs, const uint64_t *consts) {
    const uint32_t g = blockIdx.y;
    const size_t s = blockIdx.x;
    const uint32_t t0 = term_off[g], t1 = term_off[g + 1];
    uint64_t *out = wires + ((size_t)dst[g] * T + s) * ct_words;
    for (uint32_t j = threadIdx.x; j < ct_words; j += 256) {
        uint64_t accv = (j == ct_words - 1) ? consts[g] : 0;
        for (uint32_t t = t0; t < t1; t++) {
            const uint64_t v = wires[((size_t)srcs[t] * T + s) * ct_words + j];
            accv = gl_add(accv, gl_mul(v, coefs[t]));
        }
        out[j] = accv;
    }
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
    ctx->prof.pending[which].push_back({e0, e1});
}

int dev_supported(const fbs_ctx *ctx) {
    const fbs_params &p = ctx->p;
    if (p.k != 1) return set_error(ctx, FBS_E_INVALID, "this build supports GLWE dimension k = 1 only");
    if (p.log_n_poly < 8 || p.log_n_poly > 11)
        return set_error(ctx, FBS_E_INVALID, "supported polynomial sizes are N = 256, 512, 1024, 2048");
    if (p.l_bsk * p.beta_bsk > 31 || p.l_bsk < 1 || p.beta_bsk < 1)
        return set_error(ctx, FBS_E_INVALID, "need 1 <= l*beta <= 31");
    if (p.t_ksk * p.gamma_ksk > 31 || p.t_ksk < 1 || p.gamma_ksk < 1)
        return set_error(ctx, FBS_E_INVALID, "need 1 <= t*gamma <= 31");
    if (p.n < 1 || p.n > 4096) return set_error(ctx, FBS_E_INVALID, "need 1 <= n <= 4096");
    // 96-bit key-switch accumulators: D*t digits < 2^gamma times words < 2^64
    double bits = 64.0 + p.gamma_ksk + std::log2((double)p.t_ksk * ctx->D);
    if (bits > 95.0) return set_error(ctx, FBS_E_INVALID, "key-switch accumulator would overflow 96 bits");
    return FBS_OK;
}

template <int LOGN>
static int upload_keys_t(fbs_ctx *ctx) {
    const fbs_params &p = ctx->p;
    const uint32_t N = ctx->N;
    const size_t polys = (size_t)p.n * ctx->rows * (p.k + 1);
    uint64_t *d_src = nullptr;
    FBS_HIP(ctx, hipMalloc(&d_src, polys * N * 8));
    hipError_t e = hipMemcpyAsync(d_src, ctx->bsk.data(), polys * N * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        const uint64_t n_inv = gl_inv(N);
        unsigned grid = (unsigned)std::min<size_t>(polys, 4096);
        hipLaunchKernelGGL(k_bsk_transform<LOGN>, dim3(grid), dim3(64), 0, ctx->stream, d_src, ctx->d_bsk_hat, ctx->d_tw_fwd,
                           n_inv, polys);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_src);
    if (e != hipSuccess) return set_error(ctx, FBS_E_DEVICE, std::string("bootstrapping-key transform: ") + hipGetErrorString(e));
    return FBS_OK;
}

int dev_upload_keys(fbs_ctx *ctx) {
    const fbs_params &p = ctx->p;
    const uint32_t N = ctx->N;
    std::vector<uint64_t> fwd, inv;
    host_twiddles(p.log_n_poly, fwd, inv);
    const size_t bsk_words = (size_t)p.n * ctx->rows * (p.k + 1) * N;
    const size_t ksk_rows = (size_t)ctx->D * p.t_ksk;
    if (!ctx->d_tw_fwd) {
        FBS_HIP(ctx, hipMalloc(&ctx->d_tw_fwd, N * 8));
        FBS_HIP(ctx, hipMalloc(&ctx->d_tw_inv, N * 8));
        FBS_HIP(ctx, hipMalloc(&ctx->d_bsk_hat, bsk_words * 8));
        FBS_HIP(ctx, hipMalloc(&ctx->d_ksk, ksk_rows * ctx->ksk_stride * 8));
    }
    FBS_HIP(ctx, hipMemcpyAsync(ctx->d_tw_fwd, fwd.data(), N * 8, hipMemcpyHostToDevice, ctx->stream));
    FBS_HIP(ctx, hipMemcpyAsync(ctx->d_tw_inv, inv.data(), N * 8, hipMemcpyHostToDevice, ctx->stream));
    FBS_HIP(ctx, hipMemsetAsync(ctx->d_ksk, 0, ksk_rows * ctx->ksk_stride * 8, ctx->stream));
    FBS_HIP(ctx, hipMemcpy2DAsync(ctx->d_ksk, (size_t)ctx->ksk_stride * 8, ctx->ksk.data(), (size_t)(p.n + 1) * 8,
                                  (size_t)(p.n + 1) * 8, ksk_rows, hipMemcpyHostToDevice, ctx->stream));
    FBS_HIP(ctx, hipStreamSynchronize(ctx->stream));
    switch (p.log_n_poly) {
        case 8: return upload_keys_t<8>(ctx);
        case 9: return upload_keys_t<9>(ctx);
        case 10: return upload_keys_t<10>(ctx);
        case 11: return upload_keys_t<11>(ctx);
    }
    return set_error(ctx, FBS_E_INVALID, "unsupported N");
}

int dev_keyswitch(fbs_ctx *ctx, const GateView &gv, uint32_t *d_ms, hipStream_t stream) {
    const fbs_params &p = ctx->p;
    KsArgs a{};
    a.gv = gv;
    a.ksk = ctx->d_ksk;
    a.ms = d_ms;
    a.n = p.n;
    a.D = ctx->D;
    a.t = p.t_ksk;
    a.gamma = p.gamma_ksk;
    a.stride = ctx->ksk_stride;
    a.ct_words = ctx->D + 1;
    a.log2_2n = p.log_n_poly + 1;
    a.count = (size_t)gv.n_gates * gv.s_count;
    if (a.count == 0) return FBS_OK;
    hipEvent_t e0, e1;
    prof_begin(ctx, 0, stream, &e0, &e1);
    if (a.count >= 32) {
        // lanes = ciphertexts: pays once a wave is at least half full
        constexpr int COLS = 8;
        dim3 grid((unsigned)((a.count + 63) / 64), (p.n + 1 + COLS - 1) / COLS);
        hipLaunchKernelGGL(k_keyswitch_lanes<COLS>, grid, dim3(256), 0, stream, a);
    } else {
        constexpr int FB = 8;
        dim3 grid((unsigned)((a.count + FB - 1) / FB), ctx->ksk_stride / 256);
        size_t shmem = (size_t)FB * ctx->D * sizeof(uint32_t);
        hipLaunchKernelGGL(k_keyswitch<FB>, grid, dim3(256), shmem, stream, a);
    }
    prof_end(ctx, 0, stream, e0, e1);
    FBS_HIP(ctx, hipGetLastError());
    return FBS_OK;
}

int dev_blind_rotate(fbs_ctx *ctx, const fbs_tvset *tv, const GateView &gv, const uint32_t *d_ms, hipStream_t stream) {
    const fbs_params &p = ctx->p;
    BrArgs a{};
    a.gv = gv;
    a.ms = d_ms;
    a.bsk_hat = ctx->d_bsk_hat;
    a.tw_fwd = ctx->d_tw_fwd;
    a.tw_inv = ctx->d_tw_inv;
    a.tvs = tv->d_tvs;
    a.post = tv->d_post;
    a.n = p.n;
    a.l = p.l_bsk;
    a.beta = p.beta_bsk;
    a.ct_words = ctx->D + 1;
    const size_t count = (size_t)gv.n_gates * gv.s_count;
    if (count == 0) return FBS_OK;
    if (count > 0x7FFFFFFFull) return set_error(ctx, FBS_E_INVALID, "batch too large for one launch");
    dim3 grid((unsigned)count);
    hipEvent_t e0, e1;
    prof_begin(ctx, 1, stream, &e0, &e1);
    switch (p.log_n_poly) {
        case 8: hipLaunchKernelGGL(k_blind_rotate<8>, grid, dim3(128), 0, stream, a); break;
        case 9: hipLaunchKernelGGL(k_blind_rotate<9>, grid, dim3(128), 0, stream, a); break;
        case 10: hipLaunchKernelGGL(k_blind_rotate<10>, grid, dim3(128), 0, stream, a); break;
        case 11: hipLaunchKernelGGL(k_blind_rotate<11>, grid, dim3(128), 0, stream, a); break;
        default: return set_error(ctx, FBS_E_INVALID, "unsupported N");
    }
    prof_end(ctx, 1, stream, e0, e1);
    FBS_HIP(ctx, hipGetLastError());
    return FBS_OK;
}

int dev_lincomb(fbs_ctx *ctx, uint64_t *d_wires, size_t T, uint32_t n_out, const uint32_t *d_dst,
                const uint32_t *d_term_off, const uint32_t *d_srcs, const uint64_t *d_coefs, const uint64_t *d_consts,
                hipStream_t stream) {
    if (n_out == 0 || T == 0) return FBS_OK;
    if (n_out > 65535) return set_error(ctx, FBS_E_INVALID, "more than 65535 linear combinations in one launch");
    hipEvent_t e0, e1;
    prof_begin(ctx, 2, stream, &e0, &e1);
    hipLaunchKernelGGL(k_lincomb, dim3((unsigned)T, n_out), dim3(256), 0, stream, d_wires, T, ctx->D + 1, d_dst, d_term_off,
                       d_srcs, d_coefs, d_consts);
    prof_end(ctx, 2, stream, e0, e1);
    FBS_HIP(ctx, hipGetLastError());
    return FBS_OK;
}

int dev_polymul(fbs_ctx *ctx, const uint64_t *d_a, const uint64_t *d_b, uint64_t *d_c, hipStream_t stream) {
    const uint64_t n_inv = gl_inv(ctx->N);
    switch (ctx->p.log_n_poly) {
        case 8: hipLaunchKernelGGL(k_polymul<8>, dim3(1), dim3(64), 0, stream, d_a, d_b, d_c, ctx->d_tw_fwd, ctx->d_tw_inv, n_inv); break;
        case 9: hipLaunchKernelGGL(k_polymul<9>, dim3(1), dim3(64), 0, stream, d_a, d_b, d_c, ctx->d_tw_fwd, ctx->d_tw_inv, n_inv); break;
        case 10: hipLaunchKernelGGL(k_polymul<10>, dim3(1), dim3(64), 0, stream, d_a, d_b, d_c, ctx->d_tw_fwd, ctx->d_tw_inv, n_inv); break;
        case 11: hipLaunchKernelGGL(k_polymul<11>, dim3(1), dim3(64), 0, stream, d_a, d_b, d_c, ctx->d_tw_fwd, ctx->d_tw_inv, n_inv); break;
        default: return set_error(ctx, FBS_E_INVALID, "unsupported N");
    }
    FBS_HIP(ctx, hipGetLastError());
    return FBS_OK;
}

}  // namespace fbs

/* A TUNED CPU baseline for the functional bootstrap -- test/bench infrastructure, like everything under oracle/, and never
 * the checker: the parity witness stays oracle/tfhe_oracle.c (scalar, 128-bit products, textbook transforms), which this file
 * is itself checked against word for word (tests/test_oracle_tfhe.py::test_tuned_baseline_equals_the_oracle).
 *
 * Why it exists: bench.py prints a `cpu_baseline` beside the GPU figure.  The scalar oracle does ~7 bootstraps/s per thread;
 * a CPU TFHE library does one to two orders of magnitude better, and a baseline that slow says nothing.  This is what a CPU
 * does with the same scheme when it is written for the machine:
 *   * the 46-bit modulus fits the 52-bit multiplier of AVX-512 IFMA (vpmadd52luq / vpmadd52huq) exactly: a modular product
 *     by a fixed operand (twiddle, key word) is Shoup's three multiplies -- qhat = hi52(a w'), r = lo52(a w) - lo52(qhat q),
 *     w' = floor(w 2^52 / q) -- plus one conditional subtraction;
 *   * EIGHT bootstraps per vector: lane b of every vector belongs to ciphertext b of a group, so all eight share twiddles
 *     and key words (broadcast loads) and the transforms need no shuffles at all;
 *   * the key switch of a group is 8 192 x 631 IFMA multiply-adds on unsigned digits (balanced digits by the B/2 offset and
 *     one correction vector, as in the GPU kernels), reduced once at the end;
 *   * OpenMP over groups.
 * Same conventions as the oracle, hence the same ciphertexts.  One key bit per step, k = 1 only (the
 * benchmark shape); anything else is refused and bench.py reports the scalar figure alone.
 * Build: oracle/Makefile (gcc -O3 -mavx512f -mavx512dq -mavx512ifma -fopenmp); tuned_supported() says whether this CPU has IFMA. */
#include <immintrin.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "tfhe_oracle.h"

#define Q ORC_Q
#define QBITS ORC_QBITS
typedef unsigned __int128 u128;

typedef struct tuned_ctx {
    orc_params p;
    uint32_t N, rows;
    uint64_t *tw, *tw_s, *itw, *itw_s; /* psi^bitrev(i), inverse, and their Shoup companions */
    uint64_t *bsk_hat, *bsk_hat_s;     /* [n][rows][2][N], NTT domain, times 1/N; Shoup companions */
    const uint64_t *ksk;               /* borrowed: [N*t][n+1] */
    uint64_t *ks_corr;                 /* [n+1]: (B/2) * sum of all key-switching-key rows, mod q */
} tuned_ctx;

int tuned_supported(void) {
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512ifma");
}

static uint64_t mulq(uint64_t a, uint64_t b) { return (uint64_t)(((u128)a * b) % Q); }
static uint64_t powq(uint64_t a, uint64_t e) {
    uint64_t r = 1;
    for (; e; e >>= 1, a = mulq(a, a))
        if (e & 1) r = mulq(r, a);
    return r;
}
static uint64_t shoup(uint64_t w) { return (uint64_t)((((u128)w) << 52) / Q); }

/* ---- scalar transform, used once per key polynomial at set-up (same tables, same order as the vector code) ---------- */
static void ntt_fwd_scalar(const tuned_ctx *c, uint64_t *a) {
    uint32_t N = c->N;
    for (uint32_t m = 1, t = N / 2; m < N; m <<= 1, t >>= 1)
        for (uint32_t i = 0; i < m; i++) {
            uint64_t w = c->tw[m + i];
            for (uint32_t j = 2 * i * t; j < 2 * i * t + t; j++) {
                uint64_t u = a[j], v = mulq(a[j + t], w);
                a[j] = u + v >= Q ? u + v - Q : u + v;
                a[j + t] = u >= v ? u - v : u + Q - v;
            }
        }
}

tuned_ctx *tuned_create(const orc_params *p, const uint64_t *bsk, const uint64_t *ksk) {
    if (!tuned_supported() || p->k != 1 || p->bsk_group == 2 || p->log_n_poly < 4 || p->log_n_poly > 13) return NULL;
    if (p->t_ksk * p->gamma_ksk > 45 || p->gamma_ksk > 6 || p->l_bsk * p->beta_bsk > 44) return NULL;
    {   /* 64-bit sums of the key switch: N t fields below 2^gamma times words below 2^46 */
        uint64_t terms = ((uint64_t)1 << p->log_n_poly) * p->t_ksk;
        int bits = QBITS + (int)p->gamma_ksk;
        while (terms > 1) terms = (terms + 1) >> 1, bits++;
        if (bits > 63) return NULL;
    }
    tuned_ctx *c = calloc(1, sizeof *c);
    c->p = *p;
    c->N = 1u << p->log_n_poly;
    c->rows = 2 * p->l_bsk;
    uint32_t N = c->N, logn = p->log_n_poly;
    c->tw = malloc(N * 8), c->tw_s = malloc(N * 8), c->itw = malloc(N * 8), c->itw_s = malloc(N * 8);
    uint64_t psi = powq(7, (Q - 1) / (2ull * N)), ipsi = powq(psi, Q - 2);
    for (uint32_t i = 0; i < N; i++) {
        uint32_t r = 0;
        for (uint32_t b = 0; b < logn; b++) r |= ((i >> b) & 1u) << (logn - 1 - b);
        c->tw[i] = powq(psi, r);
        c->itw[i] = powq(ipsi, r);
        c->tw_s[i] = shoup(c->tw[i]);
        c->itw_s[i] = shoup(c->itw[i]);
    }
    size_t polys = (size_t)p->n * c->rows * 2;
    c->bsk_hat = aligned_alloc(64, polys * N * 8);
    c->bsk_hat_s = aligned_alloc(64, polys * N * 8);
    uint64_t ninv = powq(N, Q - 2);
#pragma omp parallel for schedule(static)
    for (long q = 0; q < (long)polys; q++) {
        uint64_t *h = c->bsk_hat + (size_t)q * N, *hs = c->bsk_hat_s + (size_t)q * N;
        memcpy(h, bsk + (size_t)q * N, N * 8);
        ntt_fwd_scalar(c, h);
        for (uint32_t j = 0; j < N; j++) {
            h[j] = mulq(h[j], ninv);
            hs[j] = shoup(h[j]);
        }
    }
    c->ksk = ksk;
    c->ks_corr = calloc(p->n + 1, 8);
    uint32_t rows = N * p->t_ksk;
    for (uint32_t i = 0; i <= p->n; i++) {
        u128 s = 0;
        for (uint32_t r = 0; r < rows; r++) s += ksk[(size_t)r * (p->n + 1) + i];
        c->ks_corr[i] = mulq((uint64_t)(s % Q), 1ull << (p->gamma_ksk - 1));
    }
    return c;
}

void tuned_destroy(tuned_ctx *c) {
    if (!c) return;
    free(c->tw), free(c->tw_s), free(c->itw), free(c->itw_s), free(c->bsk_hat), free(c->bsk_hat_s), free(c->ks_corr);
    free(c);
}

/* ---- vector arithmetic: 8 residues in [0, q) per register ------------------------------------------------------------ */
typedef __m512i v8;
static inline v8 vq(void) { return _mm512_set1_epi64((long long)Q); }
static inline v8 vred(v8 r) { return _mm512_min_epu64(r, _mm512_sub_epi64(r, vq())); } /* [0, 2q) -> [0, q) */
static inline v8 vaddq(v8 a, v8 b) { return vred(_mm512_add_epi64(a, b)); }
static inline v8 vsubq(v8 a, v8 b) { return vred(_mm512_add_epi64(_mm512_sub_epi64(a, b), vq())); }
/* a * w mod q, a < 2^52, w < q fixed with companion ws = floor(w 2^52 / q) */
static inline v8 vmul_shoup(v8 a, v8 w, v8 ws) {
    const v8 zero = _mm512_setzero_si512(), m52 = _mm512_set1_epi64((1ll << 52) - 1);
    v8 qhat = _mm512_madd52hi_epu64(zero, a, ws);
    v8 lo = _mm512_madd52lo_epu64(zero, a, w);
    v8 r = _mm512_and_si512(_mm512_sub_epi64(lo, _mm512_madd52lo_epu64(zero, qhat, vq())), m52);
    return vred(r);
}

static void ntt_fwd_v(const tuned_ctx *c, v8 *a) {
    uint32_t N = c->N;
    for (uint32_t m = 1, t = N / 2; m < N; m <<= 1, t >>= 1)
        for (uint32_t i = 0; i < m; i++) {
            v8 w = _mm512_set1_epi64((long long)c->tw[m + i]), ws = _mm512_set1_epi64((long long)c->tw_s[m + i]);
            for (uint32_t j = 2 * i * t; j < 2 * i * t + t; j++) {
                v8 u = a[j], v = vmul_shoup(a[j + t], w, ws);
                a[j] = vaddq(u, v);
                a[j + t] = vsubq(u, v);
            }
        }
}
static void ntt_inv_v(const tuned_ctx *c, v8 *a) { /* Gentleman-Sande, the mirror image; 1/N is in the key */
    uint32_t N = c->N;
    for (uint32_t m = N / 2, t = 1; m >= 1; m >>= 1, t <<= 1)
        for (uint32_t i = 0; i < m; i++) {
            v8 w = _mm512_set1_epi64((long long)c->itw[m + i]), ws = _mm512_set1_epi64((long long)c->itw_s[m + i]);
            for (uint32_t j = 2 * i * t; j < 2 * i * t + t; j++) {
                v8 u = a[j], v = a[j + t];
                a[j] = vaddq(u, v);
                a[j + t] = vmul_shoup(vsubq(u, v), w, ws);
            }
        }
}

/* ---- one group of eight ciphertexts ------------------------------------------------------------------------------------ */
typedef struct {
    v8 *acc, *sum, *dig, *pk;  /* [2][N], [2][N], [N], [N] packed digit words */
    uint64_t *small;      /* [n+1][8] key-switched ciphertexts, lane-interleaved */
    uint32_t *ms;         /* [8][n+1] */
} scratch;

static void keyswitch_group(const tuned_ctx *c, const uint64_t *const cts[8], scratch *s) {
    const uint32_t n = c->p.n, t = c->p.t_ksk, gam = c->p.gamma_ksk, D = c->N, tg = t * gam, cols = n + 1;
    v8 *out = (v8 *)s->small;
    for (uint32_t i = 0; i < cols; i++) out[i] = _mm512_setzero_si512();
    uint64_t offs = 0;
    for (uint32_t v = 0; v < t; v++) offs |= (1ull << (gam - 1)) << (v * gam);
    const uint64_t mask = (1ull << tg) - 1, bmask = (1ull << gam) - 1;
    for (uint32_t j = 0; j < D; j++) {
        uint64_t ab[8];
        for (int b = 0; b < 8; b++) ab[b] = (((((cts[b][j] >> (QBITS - 1 - tg)) + 1) >> 1) & mask) + offs) & mask;
        for (uint32_t f = 0; f < t; f++) { /* field f from the least significant end is level v = t - 1 - f */
            uint64_t u[8];
            int any = 0;
            for (int b = 0; b < 8; b++) any |= (u[b] = (ab[b] >> (f * gam)) & bmask) != 0;
            if (!any) continue;
            const v8 uv = _mm512_loadu_si512(u);
            const uint64_t *row = c->ksk + ((size_t)j * t + (t - 1 - f)) * cols;
            for (uint32_t i = 0; i < cols; i++) out[i] = _mm512_madd52lo_epu64(out[i], uv, _mm512_set1_epi64((long long)row[i]));
        }
    }
    /* small = (0, body) - (S - corr): unsigned fields are digit + B/2 */
    for (uint32_t i = 0; i < cols; i++)
        for (int b = 0; b < 8; b++) {
            uint64_t S = s->small[(size_t)i * 8 + b] % Q;
            uint64_t neg = (c->ks_corr[i] + Q - S) % Q;
            s->small[(size_t)i * 8 + b] = i == n ? (cts[b][D] + neg) % Q : neg;
        }
}

static void modswitch_group(const tuned_ctx *c, scratch *s) {
    const uint32_t sh = QBITS - c->p.log_n_poly - 1, mask = 2 * c->N - 1, n = c->p.n;
    for (int b = 0; b < 8; b++) {
        uint32_t *ms = s->ms + (size_t)b * (n + 1);
        int64_t eps = 0;
        for (uint32_t i = 0; i < n; i++) {
            uint64_t w = s->small[(size_t)i * 8 + b], m = ((w >> (sh - 1)) + 1) >> 1;
            eps += (int64_t)w - (int64_t)(m << sh);
            ms[i] = (uint32_t)m & mask;
        }
        int64_t half = eps >> 1;
        uint64_t e = half >= 0 ? (uint64_t)half % Q : Q - ((uint64_t)(-half) % Q);
        uint64_t body = (s->small[(size_t)n * 8 + b] + Q - (e % Q)) % Q;
        ms[n] = (uint32_t)(((body >> (sh - 1)) + 1) >> 1) & mask;
    }
}

static void blind_rotate_group(const tuned_ctx *c, const uint64_t *const tvs[8], scratch *s) {
    const uint32_t N = c->N, n = c->p.n, l = c->p.l_bsk, beta = c->p.beta_bsk, twoN = 2 * N;
    uint64_t *acc64 = (uint64_t *)s->acc;
    memset(s->acc, 0, (size_t)N * 64);
    for (int b = 0; b < 8; b++) { /* ACC = (0, X^{-b~} TV) */
        uint32_t r = (twoN - s->ms[(size_t)b * (n + 1) + n]) & (twoN - 1);
        for (uint32_t j = 0; j < N; j++) {
            uint32_t idx = (j + twoN - r) & (twoN - 1);
            uint64_t v = tvs[b][idx & (N - 1)];
            acc64[((size_t)N + j) * 8 + b] = idx < N ? v : (v ? Q - v : 0);
        }
    }
    const int sft = QBITS - (int)(l * beta);
    const v8 vhalfq = _mm512_set1_epi64((long long)(Q / 2)), rnd = _mm512_set1_epi64(1ll << (sft - 1));
    const v8 lbmask = _mm512_set1_epi64((1ll << (l * beta)) - 1), bm = _mm512_set1_epi64((1ll << beta) - 1);
    const v8 vhalf = _mm512_set1_epi64(1ll << (beta - 1));
    uint64_t offs = 0;
    for (uint32_t f = 0; f < l; f++) offs |= (1ull << (beta - 1)) << (f * beta);
    const v8 voffs = _mm512_set1_epi64((long long)offs);
    const v8 lane = _mm512_set_epi64(7, 6, 5, 4, 3, 2, 1, 0);
    for (uint32_t i = 0; i < n; i++) {
        uint32_t rr[8];
        int any = 0;
        for (int b = 0; b < 8; b++) any |= (rr[b] = s->ms[(size_t)b * (n + 1) + i]) != 0;
        if (!any) continue;
        /* lanes whose rotation is zero contribute a zero difference: same result as skipping the step for them */
        const v8 rv = _mm512_set_epi64(rr[7], rr[6], rr[5], rr[4], rr[3], rr[2], rr[1], rr[0]);
        memset(s->sum, 0, (size_t)2 * N * 64);
        for (uint32_t cc = 0; cc < 2; cc++) {
            const v8 *a = s->acc + (size_t)cc * N;
            const uint64_t *base = (const uint64_t *)a;
            /* (X^r - 1) ACC_cc over centred representatives, rounded: d = centred(rot) - centred(acc); abar = floor((d + 2^(s-1)) /
             * 2^s) mod B^l; B/2 added at every digit position, so that the balanced digits (carries included) are plain bit
             * fields minus B/2 -- the same digits as the oracle's carry loop, which are unique */
            for (uint32_t j = 0; j < N; j++) {
                v8 idx = _mm512_and_si512(_mm512_sub_epi64(_mm512_set1_epi64((long long)(j + twoN)), rv), _mm512_set1_epi64(twoN - 1));
                __mmask8 negm = _mm512_cmpge_epu64_mask(idx, _mm512_set1_epi64(N));
                v8 pos = _mm512_and_si512(idx, _mm512_set1_epi64(N - 1));
                v8 g = _mm512_i64gather_epi64(_mm512_add_epi64(_mm512_slli_epi64(pos, 3), lane), base, 8);
                v8 cg = _mm512_mask_sub_epi64(g, _mm512_cmpgt_epu64_mask(g, vhalfq), g, vq());
                cg = _mm512_mask_sub_epi64(cg, negm, _mm512_setzero_si512(), cg);
                v8 ca = _mm512_mask_sub_epi64(a[j], _mm512_cmpgt_epu64_mask(a[j], vhalfq), a[j], vq());
                v8 tt = _mm512_add_epi64(_mm512_sub_epi64(cg, ca), rnd);
                s->pk[j] = _mm512_and_si512(_mm512_add_epi64(_mm512_and_si512(_mm512_srai_epi64(tt, sft), lbmask), voffs), lbmask);
            }
            for (int lv = (int)l - 1; lv >= 0; lv--) {
                const uint32_t shift = ((uint32_t)l - 1u - (uint32_t)lv) * beta;   /* level l-1 is the least significant digit */
                for (uint32_t j = 0; j < N; j++) {
                    v8 dg = _mm512_sub_epi64(_mm512_and_si512(_mm512_srli_epi64(s->pk[j], shift), bm), vhalf);
                    s->dig[j] = _mm512_mask_add_epi64(dg, _mm512_cmplt_epi64_mask(dg, _mm512_setzero_si512()), dg, vq());
                }
                ntt_fwd_v(c, s->dig);
                const size_t row = (((size_t)i * c->rows) + cc * l + lv) * 2 * N;
                for (uint32_t oc = 0; oc < 2; oc++) {
                    const uint64_t *kh = c->bsk_hat + row + (size_t)oc * N, *ks = c->bsk_hat_s + row + (size_t)oc * N;
                    v8 *sm = s->sum + (size_t)oc * N;
                    for (uint32_t j = 0; j < N; j++)
                        sm[j] = vaddq(sm[j], vmul_shoup(s->dig[j], _mm512_set1_epi64((long long)kh[j]), _mm512_set1_epi64((long long)ks[j])));
                }
            }
        }
        for (uint32_t oc = 0; oc < 2; oc++) {
            v8 *sm = s->sum + (size_t)oc * N, *a = s->acc + (size_t)oc * N;
            ntt_inv_v(c, sm);
            for (uint32_t j = 0; j < N; j++) a[j] = vaddq(a[j], sm[j]);
        }
    }
}

int tuned_bootstrap_batch(const tuned_ctx *c, const uint64_t *cts_in, const uint32_t *tv_idx, const uint64_t *tvs,
                          const uint64_t *post_adds, size_t count, uint64_t *cts_out, int threads) {
    const uint32_t N = c->N, n = c->p.n;
    const size_t w = (size_t)N + 1, groups = (count + 7) / 8;
    int used = 1;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
    used = threads;
#pragma omp parallel num_threads(threads)
#endif
    {
        scratch s;
        s.acc = aligned_alloc(64, (size_t)2 * N * 64);
        s.sum = aligned_alloc(64, (size_t)2 * N * 64);
        s.dig = aligned_alloc(64, (size_t)N * 64);
        s.pk = aligned_alloc(64, (size_t)N * 64);
        s.small = aligned_alloc(64, (size_t)(n + 1) * 64);
        s.ms = malloc((size_t)8 * (n + 1) * 4);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (long g = 0; g < (long)groups; g++) {
            const uint64_t *in[8], *tv[8];
            size_t id[8];
            for (int b = 0; b < 8; b++) { /* a short last group repeats its last ciphertext */
                id[b] = (size_t)g * 8 + b < count ? (size_t)g * 8 + b : count - 1;
                in[b] = cts_in + id[b] * w;
                tv[b] = tvs + (size_t)(tv_idx ? tv_idx[id[b]] : 0) * N;
            }
            keyswitch_group(c, in, &s);
            modswitch_group(c, &s);
            blind_rotate_group(c, tv, &s);
            const uint64_t *acc64 = (const uint64_t *)s.acc;
            for (int b = 0; b < 8; b++) {
                if ((size_t)g * 8 + b >= count) break;
                uint64_t *out = cts_out + id[b] * w;
                out[0] = acc64[b];
                for (uint32_t j = 1; j < N; j++) {
                    uint64_t v = acc64[(size_t)(N - j) * 8 + b];
                    out[j] = v ? Q - v : 0;
                }
                uint64_t post = post_adds ? post_adds[tv_idx ? tv_idx[id[b]] : 0] : 0;
                out[N] = (acc64[(size_t)N * 8 + b] + post) % Q;
            }
        }
        free(s.acc), free(s.sum), free(s.dig), free(s.pk), free(s.small), free(s.ms);
    }
    return used;
}

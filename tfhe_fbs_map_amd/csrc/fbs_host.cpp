// Host side of libfbsexec.so: key generation, encryption/decryption and test-vector construction.
// None of this is on the timed path; it exists so that the drop-in `LutExecEnv.eval`
// (reference fbs_mapper/fbs_exec_env.py:208-229) can take cleartext bits in and hand cleartext
// values back, as the reference's harness (fbs_mapper/map_circuit.py:137-180) expects.
#include <algorithm>
#include <functional>
#include <cstring>
#include <thread>

#include "fbs_internal.hpp"

namespace fbs {

// ---------------------------------------------------------------------------------------------
// ChaCha20, original 64-bit-counter layout.  key = seed || fixed tail, nonce = stream id.
// ---------------------------------------------------------------------------------------------
namespace {
struct ChaCha {
    uint32_t in[16];
    ChaCha(const RandKey &key, uint64_t stream) {
        static const uint32_t sigma[4] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
        for (int i = 0; i < 4; i++) in[i] = sigma[i];
        for (int i = 0; i < 8; i++) in[4 + i] = key.w[i];
        in[12] = in[13] = 0;
        in[14] = (uint32_t)stream;
        in[15] = (uint32_t)(stream >> 32);
    }
    static uint32_t rol(uint32_t v, int s) { return (v << s) | (v >> (32 - s)); }
    static void quarter(uint32_t *x, int a, int b, int c, int d) {
        x[a] += x[b]; x[d] = rol(x[d] ^ x[a], 16);
        x[c] += x[d]; x[b] = rol(x[b] ^ x[c], 12);
        x[a] += x[b]; x[d] = rol(x[d] ^ x[a], 8);
        x[c] += x[d]; x[b] = rol(x[b] ^ x[c], 7);
    }
    void block(uint64_t counter, uint64_t out[8]) {
        uint32_t x[16];
        in[12] = (uint32_t)counter;
        in[13] = (uint32_t)(counter >> 32);
        std::memcpy(x, in, sizeof x);
        for (int round = 0; round < 20; round += 2) {
            quarter(x, 0, 4, 8, 12); quarter(x, 1, 5, 9, 13); quarter(x, 2, 6, 10, 14); quarter(x, 3, 7, 11, 15);
            quarter(x, 0, 5, 10, 15); quarter(x, 1, 6, 11, 12); quarter(x, 2, 7, 8, 13); quarter(x, 3, 4, 9, 14);
        }
        for (int i = 0; i < 8; i++)
            out[i] = (uint64_t)(x[2 * i] + in[2 * i]) | ((uint64_t)(x[2 * i + 1] + in[2 * i + 1]) << 32);
    }
};
}  // namespace

RandKey rand_key_from_seed64(uint64_t seed) {
    // key words 2..7 spell "fbs-exec-amd-gfx950-key1"
    static const uint32_t tail[6] = {0x2d736266u, 0x63657865u, 0x646d612du, 0x7866672du, 0x2d303539u, 0x3179656bu};
    RandKey k;
    k.w[0] = (uint32_t)seed;
    k.w[1] = (uint32_t)(seed >> 32);
    for (int i = 0; i < 6; i++) k.w[2 + i] = tail[i];
    return k;
}

// 32 bytes of caller entropy -> the context's key: one ChaCha block under the caller's bytes, on a stream named by the
// parameter set, so that contexts with different parameters under one seed share no key material (their secret keys would
// otherwise be prefixes of each other).
RandKey rand_key_derive(const uint8_t seed[32], const fbs_params &p) {
    RandKey master;
    std::memcpy(master.w, seed, 32);
    uint64_t h = 0xcbf29ce484222325ull;   // FNV-1a over the fields that define the key material
    const uint64_t fields[] = {p.n, p.log_n_poly, p.k, p.l_bsk, p.beta_bsk, p.t_ksk, p.gamma_ksk, p.p_msg, p.sigma_lwe, p.sigma_glwe,
                               p.bsk_group == 2 ? 2u : 1u};
    for (uint64_t f : fields)
        for (int b = 0; b < 8; b++) {
            h ^= (f >> (8 * b)) & 0xff;
            h *= 0x100000001b3ull;
        }
    ChaCha c(master, (0xFFull << 56) | (h & 0x00FFFFFFFFFFFFFFull));
    uint64_t blk[8];
    c.block(0, blk);
    RandKey k;
    std::memcpy(k.w, blk, 32);
    return k;
}

void rand_words(const RandKey &seed, uint64_t stream, uint64_t idx0, uint64_t *dst, size_t count) {
    ChaCha c(seed, stream);
    uint64_t blk[8];
    uint64_t have = ~0ull;
    for (size_t i = 0; i < count; i++) {
        uint64_t idx = idx0 + i;
        if ((idx >> 3) != have) {
            have = idx >> 3;
            c.block(have, blk);
        }
        dst[i] = blk[idx & 7];
    }
}

// Integer-only Gaussian stand-in (Irwin-Hall, 12 uniform 32-bit terms, variance 2^64), scaled by
// sigma / 2^32 and rounded half-up.  Bounded at 6 sigma; fine for tests, not a production sampler.
int64_t noise_sample(const RandKey &seed, uint64_t stream, uint64_t idx, uint64_t sigma) {
    if (!sigma) return 0;
    uint64_t w[6];
    rand_words(seed, stream, idx * 6, w, 6);
    __int128 s = -(__int128)6 * 0xFFFFFFFFll;
    for (uint64_t v : w) s += (__int128)(uint32_t)v + (__int128)(v >> 32);
    return (int64_t)((s * (__int128)sigma + ((__int128)1 << 31)) >> 32);
}

static void parallel_for(size_t n, const std::function<void(size_t, size_t)> &body) {
    unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    size_t workers = std::min<size_t>(hw, std::max<size_t>(1, n));
    if (workers <= 1) {
        body(0, n);
        return;
    }
    std::vector<std::thread> pool;
    size_t chunk = (n + workers - 1) / workers;
    for (size_t w = 0; w < workers; w++) {
        size_t a = w * chunk, b = std::min(n, a + chunk);
        if (a >= b) break;
        pool.emplace_back([=, &body] { body(a, b); });
    }
    for (auto &t : pool) t.join();
}

// ---------------------------------------------------------------------------------------------
// keys
// ---------------------------------------------------------------------------------------------
// What a context derives from its parameter set before any device is involved: range checks, sizes, Delta = 2 round(q / 4p),
// the gadget factors g_l = round(q / 2^(beta (l+1))), h_v = round(q / 2^(gamma (v+1))), and the key all randomness is expanded from.
int host_ctx_init(fbs_ctx *ctx, const fbs_params *params, uint64_t seed, const uint8_t *seed32) {
    ctx->p = *params;
    ctx->seed = seed;
    ctx->rkey = seed32 ? rand_key_derive(seed32, *params) : rand_key_from_seed64(seed);
    const fbs_params &p = ctx->p;
    if (p.log_n_poly < 2 || p.log_n_poly > 14 || p.k < 1 || p.k > 4 || p.p_msg < 1 || p.p_msg > 4096 || p.l_bsk > 16 ||
        p.t_ksk > 64)
        return set_error(ctx, FBS_E_INVALID, "parameter out of range");
    if (p.bsk_group > 2 || (p.bsk_group == 2 && (p.n & 1)))
        return set_error(ctx, FBS_E_INVALID, "bsk_group is 0, 1 or 2, and 2 needs an even n");
    ctx->N = 1u << p.log_n_poly;
    ctx->D = p.k * ctx->N;
    ctx->rows = (p.k + 1) * p.l_bsk;
    ctx->group = p.bsk_group == 2 ? 2 : 1;
    ctx->n_ggsw = ctx->group == 2 ? (size_t)p.n / 2 * 3 : p.n;
    ctx->ksk_stride = ((p.n + 1 + 255) / 256) * 256;
    ctx->delta_half = (uint64_t)(((unsigned __int128)FQ + 2ull * p.p_msg) / (4ull * p.p_msg));
    auto round_div = [](uint32_t e) {
        unsigned __int128 d = (unsigned __int128)1 << e;
        return (uint64_t)(((unsigned __int128)FQ + d / 2) / d);
    };
    for (uint32_t lv = 0; lv < p.l_bsk; lv++) ctx->g[lv] = round_div(p.beta_bsk * (lv + 1));
    for (uint32_t v = 0; v < p.t_ksk; v++) ctx->h[v] = round_div(p.gamma_ksk * (v + 1));
    return FBS_OK;
}

void host_keygen(fbs_ctx *ctx) {
    const fbs_params &p = ctx->p;
    const uint32_t N = ctx->N, D = ctx->D, n = p.n, k = p.k, l = p.l_bsk, t = p.t_ksk, rows = ctx->rows;
    ctx->sk_lwe.assign(n, 0);
    ctx->sk_glwe.assign(D, 0);
    {
        std::vector<uint64_t> w(std::max(n, D));
        rand_words(ctx->rkey, stream_id(DOM_SK_LWE, 0), 0, w.data(), n);
        for (uint32_t i = 0; i < n; i++) ctx->sk_lwe[i] = w[i] & 1;
        rand_words(ctx->rkey, stream_id(DOM_SK_GLWE, 0), 0, w.data(), D);
        for (uint32_t i = 0; i < D; i++) ctx->sk_glwe[i] = w[i] & 1;
    }
    // support of each GLWE key polynomial (binary key => A*S is a signed sum of shifted copies of A)
    std::vector<std::vector<uint32_t>> support(k);
    for (uint32_t c = 0; c < k; c++)
        for (uint32_t i = 0; i < N; i++)
            if (ctx->sk_glwe[(size_t)c * N + i]) support[c].push_back(i);

    // the bit GGSW sample g encrypts: key bit g, or for pairs (s0, s1) of key bits the products s0(1-s1), (1-s0)s1, s0 s1
    auto ggsw_bit = [&](size_t g) -> uint64_t {
        if (ctx->group != 2) return ctx->sk_lwe[g];
        const uint64_t s0 = ctx->sk_lwe[2 * (g / 3)], s1 = ctx->sk_lwe[2 * (g / 3) + 1];
        return g % 3 == 0 ? (s0 & (1 - s1)) : g % 3 == 1 ? ((1 - s0) & s1) : (s0 & s1);
    };
    const size_t row_words = (size_t)(k + 1) * N;
    ctx->bsk.assign(ctx->n_ggsw * rows * row_words, 0);
    parallel_for(ctx->n_ggsw * rows, [&](size_t r0, size_t r1) {
        std::vector<uint64_t> prod(N);
        for (size_t r = r0; r < r1; r++) {
            size_t i = r / rows;
            uint32_t rr = (uint32_t)(r % rows), comp = rr / l, lv = rr % l;
            uint64_t *row = ctx->bsk.data() + r * row_words;
            uint64_t *body = row + (size_t)k * N;
            for (uint32_t j = 0; j < N; j++)
                body[j] = fq_from_i64(noise_sample(ctx->rkey, stream_id(DOM_BSK_NOISE, r), j, p.sigma_glwe));
            for (uint32_t c = 0; c < k; c++) {
                uint64_t *a = row + (size_t)c * N;
                rand_words(ctx->rkey, stream_id(DOM_BSK_MASK, r), (uint64_t)c * N, a, N);
                for (uint32_t j = 0; j < N; j++) a[j] = fq_fold(a[j]);
                std::fill(prod.begin(), prod.end(), 0);
                for (uint32_t sh : support[c]) {
                    // prod += X^sh * a
                    for (uint32_t j = 0; j < N - sh; j++) prod[j + sh] = fq_add(prod[j + sh], a[j]);
                    for (uint32_t j = N - sh; j < N; j++) prod[j + sh - N] = fq_sub(prod[j + sh - N], a[j]);
                }
                for (uint32_t j = 0; j < N; j++) body[j] = fq_add(body[j], prod[j]);
            }
            if (ggsw_bit(i)) row[(size_t)comp * N] = fq_add(row[(size_t)comp * N], ctx->g[lv]);
        }
    });

    ctx->ksk.assign((size_t)D * t * (n + 1), 0);
    parallel_for((size_t)D * t, [&](size_t r0, size_t r1) {
        for (size_t r = r0; r < r1; r++) {
            uint32_t j = (uint32_t)(r / t), v = (uint32_t)(r % t);
            uint64_t *row = ctx->ksk.data() + r * (n + 1);
            rand_words(ctx->rkey, stream_id(DOM_KSK_MASK, r), 0, row, n);
            uint64_t b = fq_from_i64(noise_sample(ctx->rkey, stream_id(DOM_KSK_NOISE, r), 0, p.sigma_lwe));
            for (uint32_t i = 0; i < n; i++) {
                row[i] = fq_fold(row[i]);
                if (ctx->sk_lwe[i]) b = fq_add(b, row[i]);
            }
            if (ctx->sk_glwe[j]) b = fq_add(b, ctx->h[v]);
            row[n] = b;
        }
    });
}

// ---------------------------------------------------------------------------------------------
// encrypt / decrypt under the big key
// ---------------------------------------------------------------------------------------------
void host_encrypt(const fbs_ctx *ctx, const int64_t *msgs, size_t count, uint64_t nonce0, uint64_t *cts) {
    const uint32_t D = ctx->D;
    const uint64_t delta = 2 * ctx->delta_half;
    parallel_for(count, [&](size_t a, size_t b) {
        for (size_t i = a; i < b; i++) {
            uint64_t *ct = cts + i * (D + 1);
            rand_words(ctx->rkey, stream_id(DOM_ENC_MASK, nonce0 + i), 0, ct, D);
            uint64_t body = fq_from_i64(noise_sample(ctx->rkey, stream_id(DOM_ENC_NOISE, nonce0 + i), 0, ctx->p.sigma_glwe));
            for (uint32_t j = 0; j < D; j++) {
                ct[j] = fq_fold(ct[j]);
                if (ctx->sk_glwe[j]) body = fq_add(body, ct[j]);
            }
            ct[D] = fq_add(body, fq_mul(fq_from_i64(msgs[i]), delta));
        }
    });
}

void host_decrypt(const fbs_ctx *ctx, const uint64_t *cts, size_t count, int64_t *msgs) {
    const uint32_t D = ctx->D;
    const uint64_t two_p = 2ull * ctx->p.p_msg;
    parallel_for(count, [&](size_t a, size_t b) {
        for (size_t i = a; i < b; i++) {
            const uint64_t *ct = cts + i * (D + 1);
            uint64_t phase = ct[D];
            for (uint32_t j = 0; j < D; j++)
                if (ctx->sk_glwe[j]) phase = fq_sub(phase, ct[j]);
            unsigned __int128 v = (unsigned __int128)phase * two_p + FQ / 2;
            msgs[i] = (int64_t)((uint64_t)(v / FQ) % two_p);
        }
    });
}

// ---------------------------------------------------------------------------------------------
// test vector for one table.  A table longer than p is only computable when the values met at
// x and x+p add up to one constant c (the reference's three "modes", map_to_fbs.py:81-98 are
// c = 1, 0, 2): then f - c/2 is negacyclic, the polynomial carries (2f - c) * Delta/2 and c*Delta/2
// is added back after sample extraction.
// ---------------------------------------------------------------------------------------------
int host_build_tv(const fbs_ctx *ctx, const int32_t *table, uint32_t len, uint64_t *tv, uint64_t *post_add) {
    const uint32_t p = ctx->p.p_msg, N = ctx->N;
    if (len == 0 || len > 2 * p) return FBS_E_TABLE;
    int64_t c = 0;
    if (len > p) {
        c = (int64_t)table[0] + table[p];
        for (uint32_t i = 0; i + p < len; i++)
            if ((int64_t)table[i] + table[i + p] != c) return FBS_E_TABLE;
    }
    std::vector<uint64_t> enc(p);
    for (uint32_t x = 0; x < p; x++) {
        int64_t f = x < len ? table[x] : 0;   // unreachable slots
        enc[x] = fq_mul(fq_from_i64(2 * f - c), ctx->delta_half);
    }
    for (uint32_t j = 0; j < N; j++) {
        uint64_t x = ((uint64_t)j * 2 * p + N) / (2ull * N);   // nearest multiple of N/p
        tv[j] = x < p ? enc[x] : fq_neg(enc[0]);               // the half box below X^N wraps to -f(0)
    }
    *post_add = fq_mul(fq_from_i64(c), ctx->delta_half);
    return FBS_OK;
}

// ---------------------------------------------------------------------------------------------
// Several tables on one blind rotation (multi-value bootstrap; Carpov, Izabachene, Mollimard, CT-RSA 2019).  The test
// vector above is delta_half * G(X) with G_j = +-(2 f - c), and (1 + X + .. + X^(N-1)) (1 - X) = 2 in Z[X]/(X^N + 1), so
//     TV_F = TV_0 * D_F,   TV_0 = delta_half (1 + X + .. + X^(N-1)),   D_F = G (1 - X) / 2
// -- an integer polynomial (every G_j has the parity of c) that is non-zero only where the table changes value: at box
// boundaries j with f(x_j) != f(x_j - 1), and where the last half box flips to -(2 f(0) - c).  The coefficient at j = 0,
// G_0 + G_(N-1), is zero by that very flip.
// ---------------------------------------------------------------------------------------------
int host_build_tv_diff(const fbs_ctx *ctx, const int32_t *table, uint32_t len, uint32_t *pos, int32_t *val, uint32_t *count,
                       uint64_t *norm2, uint64_t *g_norm2, uint64_t *abs_sum) {
    const uint32_t p = ctx->p.p_msg, N = ctx->N;
    if (len == 0 || len > 2 * p) return FBS_E_TABLE;
    int64_t c = 0;
    if (len > p) {
        c = (int64_t)table[0] + table[p];
        for (uint32_t i = 0; i + p < len; i++)
            if ((int64_t)table[i] + table[i + p] != c) return FBS_E_TABLE;
    }
    auto g_of = [&](uint32_t j) -> int64_t {
        const uint64_t x = ((uint64_t)j * 2 * p + N) / (2ull * N);
        const int64_t f = x < p ? (x < len ? table[x] : 0) : table[0];
        return x < p ? 2 * f - c : -(2 * f - c);
    };
    uint32_t n = 0;
    uint64_t n2 = 0, g2 = 0, sum = 0;
    int64_t prev = -g_of(N - 1);   // G_(-1) = -G_(N-1)
    for (uint32_t j = 0; j < N; j++) {
        const int64_t g = g_of(j);
        g2 += (uint64_t)(g * g);
        const int64_t d = (g - prev) / 2;
        prev = g;
        if (d == 0) continue;
        if (n > p || d > INT32_MAX || d < INT32_MIN) return FBS_E_TABLE;   // (cannot happen: p boxes, p + 1 boundaries)
        pos[n] = j;
        val[n] = (int32_t)d;
        n++;
        n2 += (uint64_t)(d * d);
        sum += (uint64_t)(d < 0 ? -d : d);
    }
    *count = n;
    *norm2 = n2;
    *g_norm2 = g2;
    *abs_sum = sum;
    return FBS_OK;
}

// ---------------------------------------------------------------------------------------------
// twiddles for the merged negacyclic NTT: fwd[i] = psi^bitrev(i), inv[i] = psi^-bitrev(i), psi = 7^((q-1)/2N)
// (any primitive 2N-th root gives the same ciphertexts: the transform is an internal representation).
// Entries [N, 2N) repeat the two half-size subtrees (nodes 2 and 3 of the twiddle tree) as tables of their own, N/2
// entries each: entry i of half h = entry ((2 + h) << d) + (i - 2^d), d = floor(log2 i); entries [2N, 3N) the four
// quarter-size subtrees (nodes 4 .. 7) likewise -- what the multi-wave transforms (WavesNtt, fbs_ntt_split.hpp) gather from.
// ---------------------------------------------------------------------------------------------
void host_twiddles(uint32_t log_n, std::vector<uint64_t> &fwd, std::vector<uint64_t> &inv) {
    const uint32_t N = 1u << log_n;
    const uint64_t psi = fq_pow(FQ_GENERATOR, (FQ - 1) / (2ull * N));
    const uint64_t psi_inv = fq_inv(psi);
    fwd.assign(N, 1);
    inv.assign(N, 1);
    for (uint32_t i = 1; i < N; i++) {
        uint32_t r = 0;
        for (uint32_t b = 0; b < log_n; b++) r |= ((i >> b) & 1u) << (log_n - 1 - b);
        fwd[i] = fq_pow(psi, r);
        inv[i] = fq_pow(psi_inv, r);
    }
    fwd.resize(3 * (size_t)N, 1);
    inv.resize(3 * (size_t)N, 1);
    for (uint32_t parts = 2, base = N; parts <= 4; parts *= 2, base += N)   // [N, 2N): the two halves; [2N, 3N): the four quarters
        for (uint32_t h = 0; h < parts; h++)
            for (uint32_t i = 1; i < N / parts; i++) {
                uint32_t d = 31 - (uint32_t)__builtin_clz(i);
                uint32_t big = ((parts + h) << d) + (i - (1u << d));
                fwd[base + h * (N / parts) + i] = fwd[big];
                inv[base + h * (N / parts) + i] = inv[big];
            }
}

}  // namespace fbs

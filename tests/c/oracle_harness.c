/* Sanitizer harness for the CPU oracle and the tuned CPU baseline (test infrastructure, both): oracle/tfhe_oracle.c and
 * oracle/tfhe_tuned.c compiled with -fsanitize=address,undefined by tests/c/Makefile and run by tests/test_sanitizers.py at toy
 * parameter sets -- key generation, encryption, every stage of a bootstrap, several tables on one rotation, a batch, and the tuned
 * baseline against the scalar oracle word for word (where the CPU has AVX-512 IFMA). */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../oracle/tfhe_oracle.h"

typedef struct tuned_ctx tuned_ctx;
int tuned_supported(void);
tuned_ctx *tuned_create(const orc_params *p, const uint64_t *bsk, const uint64_t *ksk);
void tuned_destroy(tuned_ctx *c);
int tuned_bootstrap_batch(const tuned_ctx *c, const uint64_t *cts_in, const uint32_t *tv_idx, const uint64_t *tvs, const uint64_t *post_adds,
                          size_t count, uint64_t *cts_out, int threads);

static int run(orc_params p) {
    const uint32_t N = 1u << p.log_n_poly, D = p.k * N, ctw = D + 1;
    orc_ctx *c = orc_create(&p, 5);
    if (!c) return 1;
    orc_keygen(c);
    const int32_t tables[3][14] = {{0, 1, 1, 0, 1, 0, 0}, {0, 1, 1, 0, 1, 0, 0, 1, 0, 0, 1, 0, 1, 1}, {0, 1, 2, 3, 2, 1, 0}};
    const uint32_t lens[3] = {7, 14, 7};
    uint64_t *tvs = malloc(3 * (size_t)N * 8), posts[3];
    for (int t = 0; t < 3; t++)
        if (orc_build_tv(c, tables[t], lens[t], tvs + (size_t)t * N, &posts[t]) != 0) return 2;
    enum { B = 11 };
    int64_t msgs[B], back[B];
    uint32_t ids[B];
    for (int i = 0; i < B; i++) ids[i] = (uint32_t)(i % 3), msgs[i] = i % (int)lens[ids[i]];
    uint64_t *in = malloc((size_t)B * ctw * 8), *out = malloc((size_t)B * ctw * 8), *out2 = malloc((size_t)B * ctw * 8);
    orc_encrypt(c, msgs, B, 3, in);
    orc_decrypt(c, in, B, back);
    for (int i = 0; i < B; i++)
        if (back[i] != msgs[i]) return 3;
    memset(in + (size_t)(B - 1) * ctw, 0, (size_t)D * 8);                 /* mask zero: every step of its rotation is skipped (its body no longer means msgs[B - 1]) */
    /* one bootstrap stage by stage == orc_bootstrap */
    uint64_t *small = malloc((size_t)(p.n + 1) * 8), *acc = malloc((size_t)(p.k + 1) * N * 8), *one = malloc((size_t)ctw * 8), *two = malloc((size_t)ctw * 8);
    uint32_t *ms = malloc((size_t)(p.n + 1) * 4);
    orc_keyswitch(c, in, small);
    orc_modswitch(c, small, ms);
    orc_blind_rotate(c, ms, tvs, acc);
    orc_sample_extract(c, acc, posts[0], one);
    orc_bootstrap(c, in, tvs, posts[0], two);
    if (memcmp(one, two, (size_t)ctw * 8)) return 4;
    if (orc_bootstrap_batch(c, in, ids, tvs, posts, B, out, 2) <= 0) return 5;
    orc_decrypt(c, out, B, back);
    for (int i = 0; i < B - 1; i++)
        if (back[i] != tables[ids[i]][msgs[i]]) return 6;
    if (p.k == 1) {                                                        /* several tables cut out of one rotation of TV_0 */
        uint64_t *tv0 = malloc((size_t)N * 8), post;
        int32_t *diff = malloc((size_t)N * 4);
        orc_tv0(c, tv0);
        orc_blind_rotate(c, ms, tv0, acc);
        for (int t = 0; t < 3; t += 2) {                                   /* (message 0 is below both tables' lengths) */
            int64_t m;
            if (orc_build_tv_diff(c, tables[t], lens[t], diff, &post) != 0) return 7;
            orc_multi_extract(c, acc, diff, post, one);
            orc_decrypt(c, one, 1, &m);
            if (m != tables[t][msgs[0]]) return 8;
        }
        free(tv0), free(diff);
    }
    if (tuned_supported()) {
        tuned_ctx *t = tuned_create(&p, orc_bsk(c), orc_ksk(c));
        if (t) {                                                           /* (NULL: a shape the tuned baseline does not cover) */
            if (tuned_bootstrap_batch(t, in, ids, tvs, posts, B, out2, 2) <= 0) return 9;
            if (memcmp(out, out2, (size_t)B * ctw * 8)) return 10;
            tuned_destroy(t);
        }
    }
    free(tvs), free(in), free(out), free(out2), free(small), free(acc), free(one), free(two), free(ms);
    orc_destroy(c);
    printf("set n=%u N=%u k=%u l=%u group=%u ok\n", p.n, N, p.k, p.l_bsk, p.bsk_group);
    return 0;
}

int main(void) {
    const orc_params sets[] = {
        {.n = 12, .log_n_poly = 8, .k = 1, .l_bsk = 3, .beta_bsk = 7, .t_ksk = 8, .gamma_ksk = 2, .p_msg = 7, .sigma_lwe = 256, .sigma_glwe = 16, .bsk_group = 1},
        {.n = 12, .log_n_poly = 8, .k = 1, .l_bsk = 1, .beta_bsk = 20, .t_ksk = 8, .gamma_ksk = 2, .p_msg = 7, .sigma_lwe = 256, .sigma_glwe = 4, .bsk_group = 2},
        {.n = 8, .log_n_poly = 8, .k = 2, .l_bsk = 1, .beta_bsk = 21, .t_ksk = 8, .gamma_ksk = 2, .p_msg = 7, .sigma_lwe = 256, .sigma_glwe = 4, .bsk_group = 2},
    };
    for (size_t i = 0; i < sizeof sets / sizeof sets[0]; i++) {
        const int rc = run(sets[i]);
        if (rc) {
            printf("set %zu failed at check %d\n", i, rc);
            return 1;
        }
    }
    return 0;
}

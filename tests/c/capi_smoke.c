/* Plain-C client of include/fbs_exec.h: proves the header is C (not C++), that every entry point links, and -- on a
 * machine with an MI355X -- runs one bootstrap through the raw C ABI.  Exit code 0 = all good, 3 = no GPU (expected
 * in the build container), anything else = failure. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fbs_exec.h"

int main(void) {
    fbs_params p = {16, 10, 1, 3, 7, 8, 2, 7, 256, 256};
    fbs_ctx *ctx = NULL;
    int rc = fbs_ctx_create(&p, 1, 0, &ctx);
    if (rc == FBS_E_DEVICE) {
        printf("no device: %s\n", fbs_last_error(NULL));
        return 3;
    }
    if (rc != FBS_OK) { printf("create failed: %s\n", fbs_last_error(NULL)); return 1; }
    if (fbs_keygen(ctx) != FBS_OK) { printf("keygen: %s\n", fbs_last_error(ctx)); return 1; }
    const int32_t table[7] = {0, 1, 1, 0, 1, 0, 0};
    const uint32_t off[2] = {0, 7};
    fbs_tvset *tv = NULL;
    if (fbs_tvset_create(ctx, table, off, 1, &tv) != FBS_OK) { printf("tv: %s\n", fbs_last_error(ctx)); return 1; }
    enum { B = 7, W = 1025 };
    int64_t msgs[B], got[B];
    uint64_t *in = malloc(sizeof(uint64_t) * B * W), *out = malloc(sizeof(uint64_t) * B * W);
    for (int i = 0; i < B; i++) msgs[i] = i;
    if (fbs_encrypt(ctx, msgs, B, 0, in) != FBS_OK) return 1;
    if (fbs_bootstrap_batch(ctx, tv, in, NULL, B, out) != FBS_OK) { printf("bootstrap: %s\n", fbs_last_error(ctx)); return 1; }
    if (fbs_decrypt(ctx, out, B, got) != FBS_OK) return 1;
    for (int i = 0; i < B; i++)
        if (got[i] != table[i]) { printf("mismatch at %d: %lld\n", i, (long long)got[i]); return 2; }
    printf("ok on %s\n", fbs_device_info(ctx));
    fbs_tvset_destroy(tv);
    fbs_ctx_destroy(ctx);
    free(in);
    free(out);
    return 0;
}

/* A batch of NAND gates over a device group, in plain C: one context, stream and host thread per listed GPU, keys
 * uploaded once and replicated device to device, the batch split into contiguous shards; then a two-level netlist over
 * replicated ciphertext pools (tfhe_hip_group_gate_level).
 *   gcc -O2 -Iinclude examples/group_nand.c -Ltfhe_amd -ltfhe_hip -ltfhe_client -Wl,-rpath,$PWD/tfhe_amd -o group_nand
 *   ./group_nand 20000 0 1 2 3        (gate count, then the devices; a device may be listed twice)
 * The reference has no multi-device code; its equivalent is the loop of src/test/test-gate-bootstrapping.cpp:60-80. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "tfhe_hip.h"

void tfhe_client_keygen(uint64_t seed, int32_t *lwe_key, int32_t *tlwe_key, int32_t *bk_coef, int32_t *ks);
void tfhe_client_encrypt(uint64_t seed, const int32_t *lwe_key, const uint8_t *bits, int64_t count, int32_t *out);
void tfhe_client_decrypt(const int32_t *key, int32_t n, const int32_t *ct, int64_t count, uint8_t *bits, int32_t *phases);

#define CHECK(call)                                                            \
    do {                                                                       \
        if ((call) != TFHE_HIP_OK) {                                           \
            fprintf(stderr, "%s: %s\n", #call, tfhe_hip_last_error());         \
            return 1;                                                          \
        }                                                                      \
    } while (0)

int main(int argc, char **argv) {
    const int64_t count = argc > 1 ? atoll(argv[1]) : 5000;
    int devices[64], ndev = 0;
    for (int i = 2; i < argc && ndev < 64; i++) devices[ndev++] = atoi(argv[i]);
    if (ndev == 0) devices[ndev++] = 0;
    const int n = TFHE_HIP_n, N = TFHE_HIP_N;
    int32_t *lwe_key = calloc(n, 4), *tlwe_key = calloc(N, 4);
    int32_t *bk = malloc((size_t) n * TFHE_HIP_KPL * 2 * N * 4), *ks = malloc((size_t) N * TFHE_HIP_KS_T * TFHE_HIP_KS_BASE * (n + 1) * 4);
    tfhe_client_keygen(42, lwe_key, tlwe_key, bk, ks);

    tfhe_hip_group *grp;
    CHECK(tfhe_hip_group_create(&grp, devices, ndev));
    CHECK(tfhe_hip_group_load_keys_coef(grp, bk, ks));       /* one upload, ndev - 1 device-to-device copies */

    uint8_t *a = malloc(count), *b = malloc(count), *got = malloc(count);
    for (int64_t i = 0; i < count; i++) { a[i] = (uint8_t) (rand() & 1); b[i] = (uint8_t) (rand() & 1); }
    size_t bytes = (size_t) count * (n + 1) * 4;
    int32_t *ca = malloc(bytes), *cb = malloc(bytes), *co = malloc(bytes), *single = malloc(bytes);
    tfhe_client_encrypt(1, lwe_key, a, count, ca);
    tfhe_client_encrypt(2, lwe_key, b, count, cb);
    CHECK(tfhe_hip_group_gate_batch(grp, TFHE_HIP_NAND, co, ca, cb, NULL, count));
    /* the same batch on the first member alone: identical ciphertexts */
    CHECK(tfhe_hip_gate_batch(tfhe_hip_group_ctx(grp, 0), TFHE_HIP_NAND, single, ca, cb, NULL, count));
    tfhe_client_decrypt(lwe_key, n, co, count, got, NULL);
    int64_t bad = 0;
    for (int64_t i = 0; i < count; i++) bad += got[i] != (uint8_t) (1 - (a[i] & b[i]));
    const int same = memcmp(co, single, bytes) == 0;
    printf("%lld NAND gates on %d member(s), %lld wrong, group == single context: %s\n", (long long) count,
           tfhe_hip_group_size(grp), (long long) bad, same ? "yes" : "NO");

    /* a two-level netlist over the group: rows [0, m) = a, [m, 2m) = b, level 1: x = a NAND b into [2m, 3m), level 2:
     * y = x XOR a into [3m, 4m).  Every member holds a replica of the pool; each evaluates its block of a level and the
     * rows it wrote travel device to device into the other replicas. */
    const int64_t m = count < 600 ? count : 600, rows = 4 * m;
    const size_t pool_bytes = (size_t) rows * (n + 1) * 4, part = (size_t) m * (n + 1) * 4;
    int32_t *pool = calloc(1, pool_bytes), *back = malloc(pool_bytes), *gates = malloc((size_t) m * 5 * 4);
    memcpy(pool, ca, part);
    memcpy((char *) pool + part, cb, part);
    int32_t *replica[64];
    for (int d = 0; d < ndev; d++) {
        CHECK(tfhe_hip_dev_alloc(tfhe_hip_group_ctx(grp, d), pool_bytes, (void **) &replica[d]));
        CHECK(tfhe_hip_copy_to_dev(tfhe_hip_group_ctx(grp, d), replica[d], pool, pool_bytes));
    }
    for (int level = 0; level < 2; level++) {
        for (int64_t i = 0; i < m; i++) {
            int32_t *g = gates + 5 * i;
            g[0] = level == 0 ? TFHE_HIP_NAND : TFHE_HIP_XOR;
            g[1] = (int32_t) ((2 + level) * m + i);
            g[2] = (int32_t) (level == 0 ? i : 2 * m + i);
            g[3] = (int32_t) (level == 0 ? m + i : i);
            g[4] = -1;
        }
        CHECK(tfhe_hip_group_gate_level(grp, replica, rows, gates, m));
    }
    int64_t bad2 = 0, differ = 0;
    for (int d = 0; d < ndev; d++) {
        CHECK(tfhe_hip_copy_from_dev(tfhe_hip_group_ctx(grp, d), d == 0 ? pool : back, replica[d], pool_bytes));
        if (d > 0) differ += memcmp(pool, back, pool_bytes) != 0;
        CHECK(tfhe_hip_dev_free(tfhe_hip_group_ctx(grp, d), replica[d]));
    }
    tfhe_client_decrypt(lwe_key, n, pool + 3 * m * (n + 1), m, got, NULL);
    for (int64_t i = 0; i < m; i++) bad2 += got[i] != (uint8_t) ((1 - (a[i] & b[i])) ^ a[i]);
    printf("%lld two-level circuits over the group, %lld wrong, replicas equal: %s\n", (long long) m, (long long) bad2,
           differ ? "NO" : "yes");
    tfhe_hip_group_destroy(grp);
    return bad != 0 || !same || bad2 != 0 || differ != 0;
}

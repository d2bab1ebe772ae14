/* A batch of NAND gates over a device group, in plain C: one context, stream and host thread per listed GPU, keys
 * uploaded once and replicated device to device, the batch split into contiguous shards.
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
    tfhe_hip_group_destroy(grp);
    return bad != 0 || !same;
}

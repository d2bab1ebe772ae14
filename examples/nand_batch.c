/* Flat C-ABI end to end, in plain C: client keys from a seed, the whole cloud keyset drawn on the GPU, a batch of
 * encrypted NAND gates, decryption.  Build (after `python -m tfhe_amd.build`):
 *   gcc -O2 -Iinclude examples/nand_batch.c -Ltfhe_amd -ltfhe_hip -ltfhe_client -Wl,-rpath,$PWD/tfhe_amd -o nand_batch
 * The reference's equivalent is the loop of src/test/test-gate-bootstrapping.cpp:60-80 over bootsNAND. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "tfhe_hip.h"

/* tfhe_amd/csrc/client_keys.c (portable client library) */
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
    const int64_t count = argc > 1 ? atoll(argv[1]) : 1000;
    const int n = TFHE_HIP_n, N = TFHE_HIP_N;
    int32_t *lwe_key = calloc(n, 4), *tlwe_key = calloc(N, 4);
    int32_t *bk = malloc((size_t) n * TFHE_HIP_KPL * 2 * N * 4), *ks = malloc((size_t) N * TFHE_HIP_KS_T * TFHE_HIP_KS_BASE * (n + 1) * 4);
    tfhe_client_keygen(42, lwe_key, tlwe_key, bk, ks);   /* secret keys; its cloud keys are only compared below */

    tfhe_hip_ctx *ctx;
    CHECK(tfhe_hip_create(&ctx, 0));
    CHECK(tfhe_hip_keygen_cloud(ctx, 42, lwe_key, tlwe_key, NULL, NULL)); /* same cloud keys, drawn on the GPU */

    uint8_t *a = malloc(count), *b = malloc(count), *got = malloc(count);
    for (int64_t i = 0; i < count; i++) { a[i] = (uint8_t) (rand() & 1); b[i] = (uint8_t) (rand() & 1); }
    int32_t *ca = malloc((size_t) count * (n + 1) * 4), *cb = malloc((size_t) count * (n + 1) * 4), *co = malloc((size_t) count * (n + 1) * 4);
    CHECK(tfhe_hip_encrypt_batch(ctx, ca, 1, 0, lwe_key, a, count));      /* bootsSymEncrypt on the GPU */
    tfhe_client_encrypt(2, lwe_key, b, count, cb);                        /* ... or on the CPU */
    CHECK(tfhe_hip_gate_batch(ctx, TFHE_HIP_NAND, co, ca, cb, NULL, count));
    tfhe_client_decrypt(lwe_key, n, co, count, got, NULL);
    int64_t bad = 0;
    for (int64_t i = 0; i < count; i++) bad += got[i] != (uint8_t) (1 - (a[i] & b[i]));
    printf("%lld NAND gates, %lld wrong\n", (long long) count, (long long) bad);
    tfhe_hip_destroy(ctx);
    return bad != 0;
}

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "tfhe_oracle.h"
void tfhe_client_keygen(uint64_t seed, int32_t *lwe_key, int32_t *tlwe_key, int32_t *bk_coef, int32_t *ks);
void tfhe_client_encrypt(uint64_t seed, const int32_t *lwe_key, const uint8_t *bits, int64_t count, int32_t *out);
void tfhe_client_decrypt(const int32_t *key, int32_t n, const int32_t *ct, int64_t count, uint8_t *bits, int32_t *phases);
int main(void) {
    enum { n = ORC_n, N = ORC_N, kpl = ORC_kpl };
    int32_t *lk = calloc(n, 4), *tk = calloc(N, 4), *bk = calloc((size_t) n * kpl * 2 * N, 4), *ks = calloc((size_t) N * 8 * 4 * (n + 1), 4);
    tfhe_client_keygen(42, lk, tk, bk, ks);
    uint8_t bits[8] = {0, 1, 0, 1, 0, 0, 1, 1}, got[4];
    int32_t *ct = calloc(8 * (n + 1), 4), *out = calloc(4 * (n + 1), 4), ph[4];
    tfhe_client_encrypt(7, lk, bits, 8, ct);
    oracle_keys *k = oracle_keys_from_coef(bk, ks);
    for (int gate = 0; gate <= 10; gate += 5) {
        oracle_gate_batch(k, gate, out, ct, ct + 4 * (n + 1), ct + 2 * (n + 1), 4, 2);
        tfhe_client_decrypt(lk, n, out, 4, got, ph);
        printf("gate %d: %d%d%d%d\n", gate, got[0], got[1], got[2], got[3]);
    }
    return 0;
}

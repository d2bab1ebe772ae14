// TEST INFRASTRUCTURE (see oracle/README.md): a small application written ONLY against the reference's public API
// (src/include/tfhe.h, tfhe_gate_bootstrapping_functions.h) and linked ONLY against the reference build
// (oracle/_ref/libtfhe_ref.so): reference key generation, reference encryption, a gate-by-gate loop in the style of
// src/test/test-gate-bootstrapping.cpp:60-88, reference decryption.  The tests run it twice - as it is (gates on the
// CPU) and with LD_PRELOAD=tfhe_amd/libtfhe_hip_gates.so (gates on the GPU, nothing else changed) - and compare the
// printed fingerprint of all output ciphertexts: the reference's generator is deterministic from its default seed,
// so both runs see the same keys and inputs.
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <ctime>
#include "tfhe.h"
#include "tfhe_io.h"

static uint64_t fnv(uint64_t h, const void *p, size_t n) {
    const unsigned char *b = (const unsigned char *) p;
    for (size_t i = 0; i < n; i++) h = (h ^ b[i]) * 1099511628211ull;
    return h;
}

int main(int argc, char **argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 4;
    TFheGateBootstrappingParameterSet *params = new_default_gate_bootstrapping_parameters(110);
    TFheGateBootstrappingSecretKeySet *key = new_random_gate_bootstrapping_secret_keyset(params);
    const TFheGateBootstrappingCloudKeySet *bk = &key->cloud;
    const int n = params->in_out_params->n;
    const int per = 6;  // gates per round
    LweSample *in = new_gate_bootstrapping_ciphertext_array(3 * rounds, params);
    LweSample *out = new_gate_bootstrapping_ciphertext_array(per * rounds, params);
    int bits[3 * 64];
    for (int i = 0; i < 3 * rounds; i++) {
        bits[i] = (i * 7 + 3) % 5 < 2;
        bootsSymEncrypt(&in[i], bits[i], key);
    }
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int r = 0; r < rounds; r++) {
        const LweSample *a = &in[3 * r], *b = &in[3 * r + 1], *c = &in[3 * r + 2];
        LweSample *o = &out[per * r];
        bootsNAND(&o[0], a, b, bk);
        bootsXOR(&o[1], a, c, bk);
        bootsMUX(&o[2], a, b, c, bk);
        bootsORYN(&o[3], b, c, bk);
        bootsNOT(&o[4], &o[0], bk);          // depends on an earlier output
        bootsAND(&o[5], &o[4], &o[1], bk);   // a small circuit: (a AND b) AND (a XOR c)
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    int good = 0;
    uint64_t h = 14695981039346656037ull;
    for (int r = 0; r < rounds; r++) {
        const int a = bits[3 * r], b = bits[3 * r + 1], c = bits[3 * r + 2];
        const int want[per] = {1 - (a & b), a ^ c, a ? b : c, b | (1 - c), a & b, (a & b) & (a ^ c)};
        for (int g = 0; g < per; g++) {
            const LweSample *o = &out[per * r + g];
            good += bootsSymDecrypt(o, key) == want[g];
            h = fnv(h, o->a, sizeof(int32_t) * n);
            h = fnv(h, &o->b, sizeof(int32_t));
        }
    }
    const double ms = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
    printf("gates=%d correct=%d fingerprint=%016llx ms_per_gate=%.2f\n", per * rounds, good, (unsigned long long) h,
           ms / (per * rounds));
    delete_gate_bootstrapping_ciphertext_array(per * rounds, out);
    delete_gate_bootstrapping_ciphertext_array(3 * rounds, in);
    delete_gate_bootstrapping_secret_keyset(key);
    delete_gate_bootstrapping_parameters(params);
    return good == per * rounds ? 0 : 1;
}

/* Client-side key material and ciphertexts for the gate-bootstrapping path (host CPU code, plain C).
 *
 * NOT CRYPTOGRAPHICALLY SECURE: a reproducible fixture generator for tests, benchmarks and demos.  Keys, masks and
 * noise come from xoshiro256** streams of one 64-bit seed; the generator is linear and invertible, so a ciphertext's
 * public mask words give away its noise.  Real deployments generate keys and ciphertexts with the reference's own
 * client functions over a cryptographic generator (include/tfhe_hip.h, SECURITY NOTE).
 *
 * The accelerated path only consumes keys and ciphertexts; producing them (key generation, encryption, decryption)
 * is client work that the reference does once on the CPU:
 *   new_random_gate_bootstrapping_secret_keyset   reference src/libtfhe/tfhe_gate_bootstrapping.cpp:97-108
 *   tfhe_createLweBootstrappingKey                reference src/libtfhe/lwe-bootstrapping-functions.cpp:185-217
 *   lweCreateKeySwitchKey                         reference src/libtfhe/lwe-keyswitch-functions.cpp:163-215
 *   tGswSymEncryptInt / tLweSymEncryptZero        reference src/libtfhe/tgsw-functions.cpp:191-194, tlwe-functions.cpp:26-39
 *   bootsSymEncrypt / bootsSymDecrypt             reference src/libtfhe/tfhe_gate_bootstrapping.cpp:154-165
 * This file produces the same objects in the flat layouts of include/tfhe_hip.h, from a PORTABLE integer-only
 * generator (the reference draws from std::default_random_engine + std::normal_distribution, whose streams are
 * not reproducible across standard libraries), so that tests, smoke() and bench.py regenerate identical keys on
 * the GPU box from a seed instead of shipping 145 MB of key files.  Distributions: key bits and masks uniform;
 * noise = the reference's law, a centred Gaussian truncated toward zero to a Torus32 (gaussian32,
 * numeric-functions.cpp:22-34), sampled by integer table inversion (gauss_cdt.h; tail to 9.16 sigma) with the
 * reference's standard deviations (2^-15 for LWE_n / keyswitch rows, 2^-25 for the bootstrapping key;
 * tfhe_gate_bootstrapping.cpp:62-63; for the 80-bit set 2.44e-5 and 7.18e-9, :35-36).
 */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "gauss_cdt.h"

/* one library per parameter set, as for libtfhe_hip (include/tfhe_hip.h): -DTFHE_HIP_LAMBDA=80 selects the 80-bit set.
 * Noise multipliers: standard deviation x 2^40, rounded (tfhe_gate_bootstrapping.cpp:35-36,62-63). */
#ifndef TFHE_HIP_LAMBDA
#define TFHE_HIP_LAMBDA 128
#endif
#if TFHE_HIP_LAMBDA == 128
enum { CN = 630, CL = 3, CBGBIT = 7 };
#define KS_NOISE_M ((int64_t) 1 << 25) /* 2^-15 */
#define BK_NOISE_M ((int64_t) 1 << 15) /* 2^-25 */
#elif TFHE_HIP_LAMBDA == 80
enum { CN = 500, CL = 2, CBGBIT = 10 };
#define KS_NOISE_M ((int64_t) 26828165) /* 2.44e-5 */
#define BK_NOISE_M ((int64_t) 7894)     /* 7.18e-9 */
#else
#error "TFHE_HIP_LAMBDA must be 128 or 80"
#endif
enum { CBIGN = 1024, CKPL = 2 * CL, CT = 8, CBASEBIT = 2, CBASE = 4 };

typedef struct { uint64_t s[4]; } rng_t;

static uint64_t splitmix(uint64_t *x) {
    uint64_t z = (*x += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
static void rng_seed(rng_t *r, uint64_t seed, uint64_t stream) {
    uint64_t x = seed ^ (stream * 0xD1342543DE82EF95ULL + 0x632BE59BD9B4E019ULL);
    for (int i = 0; i < 4; i++) r->s[i] = splitmix(&x);
}
static inline uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
static uint64_t rng_next(rng_t *r) { /* xoshiro256** */
    uint64_t *s = r->s, res = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
    return res;
}
static inline int32_t rng_torus(rng_t *r) { return (int32_t) (uint32_t) (rng_next(r) >> 32); }
/* centred Gaussian noise with standard deviation m / 2^40 on the torus (m = 2^25 gives 2^-15): two words of the stream */
static uint64_t g_cdt[GAUSS_CDT_BINS];
static int g_cdt_ok = 0;
static pthread_once_t g_cdt_once = PTHREAD_ONCE_INIT;
static void cdt_init(void) { g_cdt_ok = gauss_cdt_build(g_cdt, NULL); }
/* 1 if the noise table has the pinned fingerprint on this host (a libm that rounds differently changes keys) */
int tfhe_client_selftest(void) {
    pthread_once(&g_cdt_once, cdt_init);
    return g_cdt_ok;
}
static int32_t rng_gauss(rng_t *r, int64_t m) {
    const uint64_t w_mag = rng_next(r), w_pos = rng_next(r);
    return gauss_cdt_value(g_cdt, w_mag, w_pos, m);
}
/* one centred noise sample of standard deviation m / 2^40 from stream `stream` of `seed`, draw `index` of it (tests: the
 * law itself - sigma, symmetry, tail - is checked on a few million of these) */
int32_t tfhe_client_noise_from_words(uint64_t w_mag, uint64_t w_pos, int64_t m) { /* the inversion itself (tests: tail) */
    pthread_once(&g_cdt_once, cdt_init);
    return gauss_cdt_value(g_cdt, w_mag, w_pos, m);
}
void tfhe_client_noise(uint64_t seed, int64_t m, int64_t count, int32_t *out) {
    rng_t r;
    pthread_once(&g_cdt_once, cdt_init);
    rng_seed(&r, seed, 7);
    for (int64_t i = 0; i < count; i++) out[i] = rng_gauss(&r, m);
}

/* b += key * a  in Z[X]/(X^N+1), key binary (reference torusPolynomialAddMulR, used at tlwe-functions.cpp:35) */
static void addmul_binkey(int32_t *b, const int32_t *key, const int32_t *a) {
    uint32_t acc[CBIGN];
    for (int i = 0; i < CBIGN; i++) acc[i] = (uint32_t) b[i];
    for (int j = 0; j < CBIGN; j++) {
        if (!key[j]) continue;
        for (int i = 0; i < CBIGN - j; i++) acc[i + j] += (uint32_t) a[i];
        for (int i = CBIGN - j; i < CBIGN; i++) acc[i + j - CBIGN] -= (uint32_t) a[i];
    }
    for (int i = 0; i < CBIGN; i++) b[i] = (int32_t) acc[i];
}

typedef struct {
    uint64_t seed; const int32_t *lwe_key, *tlwe_key; int32_t *bk; int i0, i1;
} bk_job;
static void *bk_worker(void *arg) {
    bk_job *j = (bk_job *) arg;
    for (int i = j->i0; i < j->i1; i++) {
        rng_t r;
        rng_seed(&r, j->seed, 0x10000 + (uint64_t) i);
        for (int p = 0; p < CKPL; p++) {
            int32_t *a = j->bk + (((size_t) i * CKPL + p) * 2 + 0) * CBIGN, *b = a + CBIGN;
            for (int q = 0; q < CBIGN; q++) b[q] = rng_gauss(&r, BK_NOISE_M);
            for (int q = 0; q < CBIGN; q++) a[q] = rng_torus(&r);
            addmul_binkey(b, j->tlwe_key, a);
            /* + message * H: row p belongs to block p / l and adds s_i / Bg^(p%l+1) to that block's polynomial
             * (reference tGswAddMuIntH, tgsw-functions.cpp:115-124) */
            int32_t h = (int32_t) (1u << (32 - (p % CL + 1) * CBGBIT));
            int32_t *target = (p / CL == 0) ? a : b;
            target[0] = (int32_t) ((uint32_t) target[0] + (uint32_t) (j->lwe_key[i] * h));
        }
    }
    return NULL;
}

void tfhe_client_keygen(uint64_t seed, int32_t *lwe_key /*n*/, int32_t *tlwe_key /*N*/,
                        int32_t *bk_coef /*[n][kpl][2][N]*/, int32_t *ks /*[N][t][base][n+1]*/) {
    rng_t r;
    pthread_once(&g_cdt_once, cdt_init);
    rng_seed(&r, seed, 1);
    for (int i = 0; i < CN; i++) lwe_key[i] = (int32_t) (rng_next(&r) >> 63);
    rng_seed(&r, seed, 2);
    for (int i = 0; i < CBIGN; i++) tlwe_key[i] = (int32_t) (rng_next(&r) >> 63);

    /* keyswitch key: ks[i][j][h] = LWE_n(h * s'_i / base^(j+1)); h = 0 is the trivial zero sample */
    rng_seed(&r, seed, 3);
    for (int i = 0; i < CBIGN; i++)
        for (int j = 0; j < CT; j++)
            for (int h = 0; h < CBASE; h++) {
                int32_t *row = ks + (((size_t) i * CT + j) * CBASE + h) * (CN + 1);
                if (h == 0) { memset(row, 0, sizeof(int32_t) * (CN + 1)); continue; }
                uint32_t b = (uint32_t) (tlwe_key[i] * h) << (32 - (j + 1) * CBASEBIT);
                b += (uint32_t) rng_gauss(&r, KS_NOISE_M);
                for (int q = 0; q < CN; q++) {
                    row[q] = rng_torus(&r);
                    b += (uint32_t) row[q] * (uint32_t) lwe_key[q];
                }
                row[CN] = (int32_t) b;
            }

    /* bootstrapping key: bk[i] = TGSW(s_i) under the TLWE key */
    enum { T = 8 };
    pthread_t th[T];
    bk_job jobs[T];
    for (int t = 0; t < T; t++) {
        bk_job j = {seed, lwe_key, tlwe_key, bk_coef, CN * t / T, CN * (t + 1) / T};
        jobs[t] = j;
        pthread_create(&th[t], NULL, bk_worker, &jobs[t]);
    }
    for (int t = 0; t < T; t++) pthread_join(th[t], NULL);
}

/* bootsSymEncrypt: mu = +-1/8, noise 2^-15 */
void tfhe_client_encrypt(uint64_t seed, const int32_t *lwe_key, const uint8_t *bits, int64_t count,
                         int32_t *out /*[count][n+1]*/) {
    pthread_once(&g_cdt_once, cdt_init);
    for (int64_t c = 0; c < count; c++) {
        rng_t r;
        rng_seed(&r, seed, 0x100000000ULL + (uint64_t) c);
        int32_t *ct = out + c * (CN + 1);
        uint32_t b = bits[c] ? (1u << 29) : (0u - (1u << 29));
        b += (uint32_t) rng_gauss(&r, KS_NOISE_M);
        for (int q = 0; q < CN; q++) {
            ct[q] = rng_torus(&r);
            b += (uint32_t) ct[q] * (uint32_t) lwe_key[q];
        }
        ct[CN] = (int32_t) b;
    }
}

/* lwePhase + bootsSymDecrypt (phase > 0) */
void tfhe_client_decrypt(const int32_t *key, int32_t n, const int32_t *ct, int64_t count, uint8_t *bits,
                         int32_t *phases /*may be NULL*/) {
    for (int64_t c = 0; c < count; c++) {
        const int32_t *s = ct + c * (n + 1);
        uint32_t axs = 0;
        for (int q = 0; q < n; q++) axs += (uint32_t) s[q] * (uint32_t) key[q];
        int32_t ph = (int32_t) ((uint32_t) s[n] - axs);
        if (bits) bits[c] = ph > 0;
        if (phases) phases[c] = ph;
    }
}

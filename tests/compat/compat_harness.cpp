// Test harness for the struct-level shim (include/tfhe_hip_compat.h): plays the role of an application written
// against libtfhe.  Builds the reference's pointer-rich structs (each LweSample with its own heap `a`, the 3-level
// keyswitch pointer table of lwekeyswitch.cpp:3-18, TGswSampleFFT -> TLweSampleFFT -> LagrangeHalfCPolynomial) from
// flat arrays, calls bootsXXX_batch / tfhe_hip_blindRotate_FFT, flattens the results for the Python test.
#include <cstddef>
#include <cstdlib>
#include <cstring>
#include <new>
#include <time.h>
#include "tfhe_hip_compat.h"

static const int n = TFHE_HIP_n, N = TFHE_HIP_N, KPL = TFHE_HIP_KPL, T = TFHE_HIP_KS_T, BASE = TFHE_HIP_KS_BASE;

struct Holder {
    LweParams in_out;
    TLweParams tlwe;
    TGswParams tgsw;
    TFheGateBootstrappingParameterSet params;
    LweKeySwitchKey ks;
    LweBootstrappingKeyFFT bkfft;
    TFheGateBootstrappingCloudKeySet cloud;
    TGswSampleFFT *bk;
};

static LweSample *new_samples(int count, int len) {
    LweSample *s = (LweSample *) calloc(count, sizeof(LweSample));
    for (int i = 0; i < count; i++) s[i].a = (int32_t *) calloc(len, sizeof(int32_t));
    return s;
}
static void free_samples(LweSample *s, int count) {
    for (int i = 0; i < count; i++) free(s[i].a);
    free(s);
}

extern "C" {

void harness_layout(int64_t *o) {
    int k = 0;
    o[k++] = sizeof(LweSample); o[k++] = offsetof(LweSample, b); o[k++] = offsetof(LweSample, current_variance);
    o[k++] = sizeof(LweParams); o[k++] = sizeof(TLweParams); o[k++] = offsetof(TLweParams, extracted_lweparams);
    o[k++] = sizeof(TGswParams); o[k++] = offsetof(TGswParams, tlwe_params); o[k++] = offsetof(TGswParams, offset);
    o[k++] = sizeof(LweKeySwitchKey); o[k++] = offsetof(LweKeySwitchKey, ks);
    o[k++] = sizeof(TLweSample); o[k++] = sizeof(TLweSampleFFT); o[k++] = sizeof(TGswSampleFFT);
    o[k++] = sizeof(LweBootstrappingKeyFFT); o[k++] = offsetof(LweBootstrappingKeyFFT, bkFFT);
    o[k++] = offsetof(LweBootstrappingKeyFFT, ks);
    o[k++] = sizeof(TFheGateBootstrappingParameterSet); o[k++] = sizeof(TFheGateBootstrappingCloudKeySet);
    o[k++] = offsetof(TFheGateBootstrappingCloudKeySet, bkFFT); o[k++] = sizeof(LagrangeHalfCPolynomial);
    o[k++] = sizeof(TorusPolynomial);
}

void *harness_make_keyset(const double *bkfft, const int32_t *ksflat) {
    Holder *h = (Holder *) calloc(1, sizeof(Holder));
    h->in_out = LweParams{n, 0., 0.};
    h->tlwe = TLweParams{N, 1, 0., 0., LweParams{N, 0., 0.}};
    h->tgsw.l = 3; h->tgsw.Bgbit = 7; h->tgsw.Bg = 128; h->tgsw.halfBg = 64; h->tgsw.maskMod = 127;
    h->tgsw.tlwe_params = &h->tlwe; h->tgsw.kpl = KPL;
    h->params.ks_t = T; h->params.ks_basebit = 2; h->params.in_out_params = &h->in_out; h->params.tgsw_params = &h->tgsw;
    // keyswitch key: ks0_raw[n*t*base] + two pointer levels
    LweSample *raw = new_samples(N * T * BASE, n);
    for (long r = 0; r < (long) N * T * BASE; r++) {
        memcpy(raw[r].a, ksflat + r * (n + 1), sizeof(int32_t) * n);
        raw[r].b = ksflat[r * (n + 1) + n];
    }
    h->ks.n = N; h->ks.t = T; h->ks.basebit = 2; h->ks.base = BASE; h->ks.out_params = &h->in_out;
    h->ks.ks0_raw = raw;
    h->ks.ks1_raw = (LweSample **) calloc(N * T, sizeof(LweSample *));
    h->ks.ks = (LweSample ***) calloc(N, sizeof(LweSample **));
    for (int p = 0; p < N * T; p++) h->ks.ks1_raw[p] = raw + BASE * p;
    for (int p = 0; p < N; p++) h->ks.ks[p] = h->ks.ks1_raw + T * p;
    // bootstrapping key
    h->bk = (TGswSampleFFT *) calloc(n, sizeof(TGswSampleFFT));
    for (int i = 0; i < n; i++) {
        h->bk[i].k = 1; h->bk[i].l = 3;
        h->bk[i].all_samples = (TLweSampleFFT *) calloc(KPL, sizeof(TLweSampleFFT));
        for (int p = 0; p < KPL; p++) {
            TLweSampleFFT *s = &h->bk[i].all_samples[p];
            s->a = (LagrangeHalfCPolynomial *) malloc(2 * sizeof(LagrangeHalfCPolynomial));
            s->b = s->a + 1; s->k = 1;
            memcpy(s->a, bkfft + ((size_t) i * KPL + p) * 2 * N, 2 * sizeof(LagrangeHalfCPolynomial));
        }
    }
    h->bkfft.in_out_params = &h->in_out; h->bkfft.bk_params = &h->tgsw; h->bkfft.accum_params = &h->tlwe;
    h->bkfft.extract_params = &h->tlwe.extracted_lweparams; h->bkfft.bkFFT = h->bk; h->bkfft.ks = &h->ks;
    h->cloud.params = &h->params; h->cloud.bk = nullptr; h->cloud.bkFFT = &h->bkfft;
    return h;
}

// rewrites the key material of an existing keyset IN PLACE (every pointer keeps its address): what the shim's
// fingerprint must notice
void harness_overwrite_keyset(void *p, const double *bkfft, const int32_t *ksflat) {
    Holder *h = (Holder *) p;
    for (long r = 0; r < (long) N * T * BASE; r++) {
        memcpy(h->ks.ks0_raw[r].a, ksflat + r * (n + 1), sizeof(int32_t) * n);
        h->ks.ks0_raw[r].b = ksflat[r * (n + 1) + n];
    }
    for (int i = 0; i < n; i++)
        for (int q = 0; q < KPL; q++)
            memcpy(h->bk[i].all_samples[q].a, bkfft + ((size_t) i * KPL + q) * 2 * N, 2 * sizeof(LagrangeHalfCPolynomial));
}

void harness_free_keyset(void *p) {
    Holder *h = (Holder *) p;
    for (int i = 0; i < n; i++) {
        for (int q = 0; q < KPL; q++) free(h->bk[i].all_samples[q].a);
        free(h->bk[i].all_samples);
    }
    free(h->bk);
    free_samples(h->ks.ks0_raw, N * T * BASE);
    free(h->ks.ks1_raw); free(h->ks.ks);
    free(h);
}

// gate ids of tfhe_hip.h; dispatches to the libtfhe-style bootsXXX_batch functions
void harness_gate(void *p, int gate, int32_t *out, const int32_t *a, const int32_t *b, const int32_t *c, int count) {
    Holder *h = (Holder *) p;
    LweSample *r = new_samples(count, n), *sa = new_samples(count, n), *sb = new_samples(count, n),
              *sc = new_samples(count, n);
    for (int i = 0; i < count; i++) {
        if (a) { memcpy(sa[i].a, a + (size_t) i * (n + 1), 4 * n); sa[i].b = a[(size_t) i * (n + 1) + n]; }
        if (b) { memcpy(sb[i].a, b + (size_t) i * (n + 1), 4 * n); sb[i].b = b[(size_t) i * (n + 1) + n]; }
        if (c) { memcpy(sc[i].a, c + (size_t) i * (n + 1), 4 * n); sc[i].b = c[(size_t) i * (n + 1) + n]; }
    }
    const TFheGateBootstrappingCloudKeySet *bk = &h->cloud;
    switch (gate) {
        case 0: bootsNAND_batch(r, sa, sb, count, bk); break;
        case 1: bootsOR_batch(r, sa, sb, count, bk); break;
        case 2: bootsAND_batch(r, sa, sb, count, bk); break;
        case 3: bootsXOR_batch(r, sa, sb, count, bk); break;
        case 4: bootsXNOR_batch(r, sa, sb, count, bk); break;
        case 5: bootsNOR_batch(r, sa, sb, count, bk); break;
        case 6: bootsANDNY_batch(r, sa, sb, count, bk); break;
        case 7: bootsANDYN_batch(r, sa, sb, count, bk); break;
        case 8: bootsORNY_batch(r, sa, sb, count, bk); break;
        case 9: bootsORYN_batch(r, sa, sb, count, bk); break;
        case 10: bootsMUX_batch(r, sa, sb, sc, count, bk); break;
        case 11: bootsNOT_batch(r, sa, count, bk); break;
        case 12: bootsCOPY_batch(r, sa, count, bk); break;
        case 13: bootsCONSTANT_batch(r, 1, count, bk); break;
        default: bootsCONSTANT_batch(r, 0, count, bk); break;
    }
    for (int i = 0; i < count; i++) {
        memcpy(out + (size_t) i * (n + 1), r[i].a, 4 * n);
        out[(size_t) i * (n + 1) + n] = r[i].b;
    }
    free_samples(sc, count); free_samples(sb, count); free_samples(sa, count); free_samples(r, count);
}

// throughput of the struct-level path alone: `count` NAND gates on LweSample arrays that already exist (what an
// application written against libtfhe holds); returns milliseconds per call
double harness_time_nand(void *p, const int32_t *a, const int32_t *b, int32_t *out, int count, int reps) {
    Holder *h = (Holder *) p;
    LweSample *r = new_samples(count, n), *sa = new_samples(count, n), *sb = new_samples(count, n);
    for (int i = 0; i < count; i++) {
        memcpy(sa[i].a, a + (size_t) i * (n + 1), 4 * n); sa[i].b = a[(size_t) i * (n + 1) + n];
        memcpy(sb[i].a, b + (size_t) i * (n + 1), 4 * n); sb[i].b = b[(size_t) i * (n + 1) + n];
    }
    bootsNAND_batch(r, sa, sb, count, &h->cloud);   // warm-up (key upload on first use)
    timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int k = 0; k < reps; k++) bootsNAND_batch(r, sa, sb, count, &h->cloud);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    for (int i = 0; i < count; i++) {
        memcpy(out + (size_t) i * (n + 1), r[i].a, 4 * n);
        out[(size_t) i * (n + 1) + n] = r[i].b;
    }
    free_samples(sb, count); free_samples(sa, count); free_samples(r, count);
    return ((t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6) / reps;
}

// the reference seam: tfhe_blindRotate_FFT(accum, bkFFT, bara, n, bk_params)
void harness_blind_rotate(void *p, int32_t *acc, const int32_t *bara) {
    Holder *h = (Holder *) p;
    TLweSample s;
    s.a = (TorusPolynomial *) malloc(2 * sizeof(TorusPolynomial));
    s.b = s.a + 1; s.current_variance = 0; s.k = 1;
    memcpy(s.a, acc, 2 * sizeof(TorusPolynomial));
    tfhe_hip_blindRotate_FFT(&s, h->bk, bara, n, &h->tgsw);
    memcpy(acc, s.a, 2 * sizeof(TorusPolynomial));
    free(s.a);
}

// an application that calls the reference's SINGLE-gate functions, one gate per call (the loop of
// src/test/test-gate-bootstrapping.cpp:69-75): `fn` is the address of bootsXXX in whichever library provides it
// (libtfhe_hip_gates.so in the tests); kind 0: (result, ca, cb, bk), 1: MUX (result, a, b, c, bk), 2: NOT / COPY
// (result, ca, bk), 3: CONSTANT (result, value, bk) with value = count_or_value
void harness_single_gates(void *p, void *fn, int kind, int32_t *out, const int32_t *a, const int32_t *b, const int32_t *c,
                          int count, int value) {
    Holder *h = (Holder *) p;
    LweSample *r = new_samples(count, n), *sa = new_samples(count, n), *sb = new_samples(count, n),
              *sc = new_samples(count, n);
    for (int i = 0; i < count; i++) {
        if (a) { memcpy(sa[i].a, a + (size_t) i * (n + 1), 4 * n); sa[i].b = a[(size_t) i * (n + 1) + n]; }
        if (b) { memcpy(sb[i].a, b + (size_t) i * (n + 1), 4 * n); sb[i].b = b[(size_t) i * (n + 1) + n]; }
        if (c) { memcpy(sc[i].a, c + (size_t) i * (n + 1), 4 * n); sc[i].b = c[(size_t) i * (n + 1) + n]; }
    }
    const TFheGateBootstrappingCloudKeySet *bk = &h->cloud;
    typedef void (*bin_t)(LweSample *, const LweSample *, const LweSample *, const TFheGateBootstrappingCloudKeySet *);
    typedef void (*mux_t)(LweSample *, const LweSample *, const LweSample *, const LweSample *, const TFheGateBootstrappingCloudKeySet *);
    typedef void (*un_t)(LweSample *, const LweSample *, const TFheGateBootstrappingCloudKeySet *);
    typedef void (*cst_t)(LweSample *, int32_t, const TFheGateBootstrappingCloudKeySet *);
    for (int i = 0; i < count; i++) {
        if (kind == 0) ((bin_t) fn)(&r[i], &sa[i], &sb[i], bk);
        else if (kind == 1) ((mux_t) fn)(&r[i], &sa[i], &sb[i], &sc[i], bk);
        else if (kind == 2) ((un_t) fn)(&r[i], &sa[i], bk);
        else ((cst_t) fn)(&r[i], value, bk);
    }
    for (int i = 0; i < count; i++) {
        memcpy(out + (size_t) i * (n + 1), r[i].a, 4 * n);
        out[(size_t) i * (n + 1) + n] = r[i].b;
    }
    free_samples(sc, count); free_samples(sb, count); free_samples(sa, count); free_samples(r, count);
}
// the reference's seam under its own name (tfhe_blindRotate_FFT of libtfhe_hip_gates.so), through its address
void harness_blind_rotate_fn(void *p, void *fn, int32_t *acc, const int32_t *bara) {
    Holder *h = (Holder *) p;
    typedef void (*seam_t)(TLweSample *, const TGswSampleFFT *, const int32_t *, const int32_t, const TGswParams *);
    TLweSample s;
    s.a = (TorusPolynomial *) malloc(2 * sizeof(TorusPolynomial));
    s.b = s.a + 1; s.current_variance = 0; s.k = 1;
    memcpy(s.a, acc, 2 * sizeof(TorusPolynomial));
    ((seam_t) fn)(&s, h->bk, bara, n, &h->tgsw);
    memcpy(acc, s.a, 2 * sizeof(TorusPolynomial));
    free(s.a);
}

// tfhe_bootstrap_FFT / tfhe_bootstrap_woKS_FFT under their own names (src/include/tfhe.h:44-45), through their address:
// `count` samples one call at a time; out rows have n + 1 (with keyswitch) or N + 1 entries
void harness_bootstrap_fn(void *p, void *fn, int with_ks, int32_t *out, const int32_t *in, int32_t mu, int count) {
    Holder *h = (Holder *) p;
    typedef void (*boot_t)(LweSample *, const LweBootstrappingKeyFFT *, int32_t, const LweSample *);
    const int dim = with_ks ? n : N;
    LweSample *r = new_samples(count, dim), *x = new_samples(count, n);
    for (int i = 0; i < count; i++) {
        memcpy(x[i].a, in + (size_t) i * (n + 1), 4 * n);
        x[i].b = in[(size_t) i * (n + 1) + n];
        ((boot_t) fn)(&r[i], &h->bkfft, mu, &x[i]);
        memcpy(out + (size_t) i * (dim + 1), r[i].a, 4 * dim);
        out[(size_t) i * (dim + 1) + dim] = r[i].b;
    }
    free_samples(x, count); free_samples(r, count);
}

}  // extern "C"

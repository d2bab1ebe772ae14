// TEST INFRASTRUCTURE ONLY - flat-array (ctypes-friendly) driver over the *reference's own* structs and functions.
// Built together with oracle/ref_shim.cpp and the reference sources into oracle/_ref/libtfhe_ref.so, in the
// authoring container only (the reference does not travel to the GPU box).  Used by oracle/gen_golden.py to
// produce tests/golden/*.npz and by tests/test_oracle_vs_ref.py to pin oracle/tfhe_oracle.c bit-for-bit.
//
// Flat layouts (all little-endian, C order) - identical to include/tfhe_hip.h:
//   LWE_n sample      int32[n+1]            = a[0..n-1], b               (n = 630)
//   LWE_N sample      int32[N+1]            = a[0..N-1], b               (N = 1024, extracted)
//   TLWE sample       int32[2][N]           = a, b
//   bk (coef domain)  int32[n][kpl][2][N]   = bk[i].all_sample[p].a[c]   (reference tfhe_io.cpp:899-905 order)
//   bkFFT             double[n][kpl][2][N/2][2]  = bkFFT[i].all_samples[p].a[c].coefsC[f] (re,im)
//   ks                int32[N][t][base][n+1]     = ks[i][j][d] as a[0..n-1], b (reference tfhe_io.cpp:777-784)
#include <cstring>
#include <cstdlib>
#include "tfhe.h"
#include "fpga.h"

extern "C" void ref_shim_MuxRotate_FFT(TLweSample *, const TLweSample *, const TGswSampleFFT *, const int32_t,
                                       const TGswParams *);

static TFheGateBootstrappingParameterSet *g_params = nullptr;
static TFheGateBootstrappingSecretKeySet *g_keys = nullptr;

static const int32_t n_ = Value_n, N_ = Value_N, kpl_ = Value_kpl, t_ = 8, base_ = 4;

static void ensure_params() {
    if (!g_params) g_params = new_default_gate_bootstrapping_parameters(110);
}

static void to_sample(LweSample *s, const int32_t *flat, int32_t n) {
    memcpy(s->a, flat, sizeof(int32_t) * n);
    s->b = flat[n];
    s->current_variance = 0.;
}
static void from_sample(int32_t *flat, const LweSample *s, int32_t n) {
    memcpy(flat, s->a, sizeof(int32_t) * n);
    flat[n] = s->b;
}
static void to_tlwe(TLweSample *s, const int32_t *flat) {
    for (int c = 0; c < 2; c++) memcpy(s->a[c].coefsT, flat + c * N_, sizeof(int32_t) * N_);
    s->current_variance = 0.;
}
static void from_tlwe(int32_t *flat, const TLweSample *s) {
    for (int c = 0; c < 2; c++) memcpy(flat + c * N_, s->a[c].coefsT, sizeof(int32_t) * N_);
}

extern "C" {

void ref_get_sizes(int32_t *out) {  // n, N, k, l, Bgbit, kpl, t, basebit
    ensure_params();
    out[0] = g_params->in_out_params->n;
    out[1] = g_params->tgsw_params->tlwe_params->N;
    out[2] = g_params->tgsw_params->tlwe_params->k;
    out[3] = g_params->tgsw_params->l;
    out[4] = g_params->tgsw_params->Bgbit;
    out[5] = g_params->tgsw_params->kpl;
    out[6] = g_params->ks_t;
    out[7] = g_params->ks_basebit;
}

void ref_get_trig_tables(double *rev, double *dir) {  // 4088 doubles each (tfhe_core.h:26,93)
    memcpy(rev, trig_table_reverse, sizeof(trig_table_reverse));
    memcpy(dir, trig_table_direct, sizeof(trig_table_direct));
}
int32_t ref_trig_table_len() { return (int32_t) (sizeof(trig_table_reverse) / sizeof(double)); }
void ref_get_bit_reversed(int32_t *out) {
    for (int i = 0; i < Value_2N; i++) out[i] = (int32_t) bit_reversed[i];
}

// reference key generation (tfhe_gate_bootstrapping.cpp:97-108) from the reference's own RNG
void ref_keygen(uint32_t seed) {
    ensure_params();
    if (g_keys) delete_gate_bootstrapping_secret_keyset(g_keys);
    tfhe_random_generator_setSeed(&seed, 1);
    g_keys = new_random_gate_bootstrapping_secret_keyset(g_params);
}

void ref_seed(uint32_t seed) { tfhe_random_generator_setSeed(&seed, 1); }

// install keys produced elsewhere (the build's portable client keygen); bkFFT is computed by the reference
// (init_LweBootstrappingKeyFFT, lwe-bootstrapping-functions-fft.cpp:24-53)
void ref_set_keys(const int32_t *lwe_key, const int32_t *tlwe_key, const int32_t *bk_coef, const int32_t *ks) {
    ensure_params();
    if (g_keys) delete_gate_bootstrapping_secret_keyset(g_keys);
    LweKey *lk = new_LweKey(g_params->in_out_params);
    memcpy(lk->key, lwe_key, sizeof(int32_t) * n_);
    TGswKey *gk = new_TGswKey(g_params->tgsw_params);
    memcpy(gk->tlwe_key.key[0].coefs, tlwe_key, sizeof(int32_t) * N_);
    LweBootstrappingKey *bk = new_LweBootstrappingKey(g_params->ks_t, g_params->ks_basebit, g_params->in_out_params,
                                                      g_params->tgsw_params);
    for (int i = 0; i < n_; i++)
        for (int p = 0; p < kpl_; p++) {
            for (int c = 0; c < 2; c++)
                memcpy(bk->bk[i].all_sample[p].a[c].coefsT, bk_coef + (((size_t) i * kpl_ + p) * 2 + c) * N_,
                       sizeof(int32_t) * N_);
            bk->bk[i].all_sample[p].current_variance = 0.;
        }
    for (int i = 0; i < N_; i++)
        for (int j = 0; j < t_; j++)
            for (int d = 0; d < base_; d++)
                to_sample(&bk->ks->ks[i][j][d], ks + (((size_t) i * t_ + j) * base_ + d) * (n_ + 1), n_);
    LweBootstrappingKeyFFT *bkFFT = new_LweBootstrappingKeyFFT(bk);
    g_keys = new TFheGateBootstrappingSecretKeySet(g_params, bk, bkFFT, lk, gk);
}

void ref_get_keys(int32_t *lwe_key, int32_t *tlwe_key, int32_t *bk_coef, int32_t *ks) {
    memcpy(lwe_key, g_keys->lwe_key->key, sizeof(int32_t) * n_);
    memcpy(tlwe_key, g_keys->tgsw_key->tlwe_key.key[0].coefs, sizeof(int32_t) * N_);
    const LweBootstrappingKey *bk = g_keys->cloud.bk;
    for (int i = 0; i < n_; i++)
        for (int p = 0; p < kpl_; p++)
            for (int c = 0; c < 2; c++)
                memcpy(bk_coef + (((size_t) i * kpl_ + p) * 2 + c) * N_, bk->bk[i].all_sample[p].a[c].coefsT,
                       sizeof(int32_t) * N_);
    for (int i = 0; i < N_; i++)
        for (int j = 0; j < t_; j++)
            for (int d = 0; d < base_; d++)
                from_sample(ks + (((size_t) i * t_ + j) * base_ + d) * (n_ + 1), &bk->ks->ks[i][j][d], n_);
}

void ref_get_bkfft(double *out) {
    const TGswSampleFFT *b = g_keys->cloud.bkFFT->bkFFT;
    for (int i = 0; i < n_; i++)
        for (int p = 0; p < kpl_; p++)
            for (int c = 0; c < 2; c++)
                memcpy(out + (((size_t) i * kpl_ + p) * 2 + c) * N_, b[i].all_samples[p].a[c].coefsC,
                       sizeof(double) * N_);
}

// ---- client side: bootsSymEncrypt / bootsSymDecrypt / lwePhase (tfhe_gate_bootstrapping.cpp:154-165) ----
void ref_encrypt(int32_t *out, int32_t bit) {
    LweSample *s = new_LweSample(g_params->in_out_params);
    bootsSymEncrypt(s, bit, g_keys);
    from_sample(out, s, n_);
    delete_LweSample(s);
}
int32_t ref_phase(const int32_t *ct) {
    LweSample *s = new_LweSample(g_params->in_out_params);
    to_sample(s, ct, n_);
    int32_t r = lwePhase(s, g_keys->lwe_key);
    delete_LweSample(s);
    return r;
}
int32_t ref_decrypt(const int32_t *ct) { return ref_phase(ct) > 0 ? 1 : 0; }

// ---- gates (boot-gates.cpp).  gate ids follow include/tfhe_hip.h ----
void ref_gate(int32_t gate, int32_t *out, const int32_t *a, const int32_t *b, const int32_t *c) {
    const LweParams *p = g_params->in_out_params;
    LweSample *r = new_LweSample(p), *sa = new_LweSample(p), *sb = new_LweSample(p), *sc = new_LweSample(p);
    if (a) to_sample(sa, a, n_);
    if (b) to_sample(sb, b, n_);
    if (c) to_sample(sc, c, n_);
    const TFheGateBootstrappingCloudKeySet *bk = &g_keys->cloud;
    switch (gate) {
        case 0: bootsNAND(r, sa, sb, bk); break;
        case 1: bootsOR(r, sa, sb, bk); break;
        case 2: bootsAND(r, sa, sb, bk); break;
        case 3: bootsXOR(r, sa, sb, bk); break;
        case 4: bootsXNOR(r, sa, sb, bk); break;
        case 5: bootsNOR(r, sa, sb, bk); break;
        case 6: bootsANDNY(r, sa, sb, bk); break;
        case 7: bootsANDYN(r, sa, sb, bk); break;
        case 8: bootsORNY(r, sa, sb, bk); break;
        case 9: bootsORYN(r, sa, sb, bk); break;
        case 10: bootsMUX(r, sa, sb, sc, bk); break;
        case 11: bootsNOT(r, sa, bk); break;
        case 12: bootsCOPY(r, sa, bk); break;
        case 13: bootsCONSTANT(r, 1, bk); break;
        case 14: bootsCONSTANT(r, 0, bk); break;
        default: abort();
    }
    from_sample(out, r, n_);
    delete_LweSample(sc); delete_LweSample(sb); delete_LweSample(sa); delete_LweSample(r);
}

// ---- bootstrapping stages (lwe-bootstrapping-functions-fft.cpp, lwe-keyswitch-functions.cpp) ----
void ref_bootstrap_woKS(int32_t *out /*N+1*/, const int32_t *in /*n+1*/, int32_t mu) {
    LweSample *x = new_LweSample(g_params->in_out_params);
    LweSample *u = new_LweSample(&g_params->tgsw_params->tlwe_params->extracted_lweparams);
    to_sample(x, in, n_);
    tfhe_bootstrap_woKS_FFT(u, g_keys->cloud.bkFFT, mu, x);
    from_sample(out, u, N_);
    delete_LweSample(u); delete_LweSample(x);
}
void ref_bootstrap(int32_t *out /*n+1*/, const int32_t *in /*n+1*/, int32_t mu) {
    LweSample *x = new_LweSample(g_params->in_out_params);
    LweSample *r = new_LweSample(g_params->in_out_params);
    to_sample(x, in, n_);
    tfhe_bootstrap_FFT(r, g_keys->cloud.bkFFT, mu, x);
    from_sample(out, r, n_);
    delete_LweSample(r); delete_LweSample(x);
}
void ref_keyswitch(int32_t *out /*n+1*/, const int32_t *in /*N+1*/) {
    LweSample *u = new_LweSample(&g_params->tgsw_params->tlwe_params->extracted_lweparams);
    LweSample *r = new_LweSample(g_params->in_out_params);
    to_sample(u, in, N_);
    lweKeySwitch(r, g_keys->cloud.bkFFT->ks, u);
    from_sample(out, r, n_);
    delete_LweSample(r); delete_LweSample(u);
}
void ref_blind_rotate(int32_t *acc /*[2][N] in/out*/, const int32_t *bara /*n*/) {
    TLweSample *a = new_TLweSample(g_params->tgsw_params->tlwe_params);
    to_tlwe(a, acc);
    tfhe_blindRotate_FFT(a, g_keys->cloud.bkFFT->bkFFT, bara, n_, g_params->tgsw_params);
    from_tlwe(acc, a);
    delete_TLweSample(a);
}
void ref_mux_rotate(int32_t *out /*[2][N]*/, const int32_t *acc /*[2][N]*/, int32_t i, int32_t barai) {
    TLweSample *a = new_TLweSample(g_params->tgsw_params->tlwe_params);
    TLweSample *r = new_TLweSample(g_params->tgsw_params->tlwe_params);
    to_tlwe(a, acc);
    ref_shim_MuxRotate_FFT(r, a, g_keys->cloud.bkFFT->bkFFT + i, barai, g_params->tgsw_params);
    from_tlwe(out, r);
    delete_TLweSample(r); delete_TLweSample(a);
}
void ref_extern_mul(int32_t *acc /*[2][N] in/out*/, int32_t i) {
    TLweSample *a = new_TLweSample(g_params->tgsw_params->tlwe_params);
    to_tlwe(a, acc);
    tGswFFTExternMulToTLwe(a, g_keys->cloud.bkFFT->bkFFT + i, g_params->tgsw_params);
    from_tlwe(acc, a);
    delete_TLweSample(a);
}
void ref_extract(int32_t *out /*N+1*/, const int32_t *acc /*[2][N]*/) {
    const TLweParams *tp = g_params->tgsw_params->tlwe_params;
    TLweSample *a = new_TLweSample(tp);
    LweSample *u = new_LweSample(&tp->extracted_lweparams);
    to_tlwe(a, acc);
    tLweExtractLweSample(u, a, &tp->extracted_lweparams, tp);
    from_sample(out, u, N_);
    delete_LweSample(u); delete_TLweSample(a);
}

// ---- polynomial primitives ----
void ref_decomp(int32_t *out /*[l][N]*/, const int32_t *poly /*N*/) {
    ensure_params();
    IntPolynomial *r = new_IntPolynomial_array(Value_l);
    TorusPolynomial *p = new_TorusPolynomial();
    memcpy(p->coefsT, poly, sizeof(int32_t) * N_);
    tGswTorus32PolynomialDecompH(r, p);
    for (int q = 0; q < Value_l; q++) memcpy(out + q * N_, r[q].coefs, sizeof(int32_t) * N_);
    delete_TorusPolynomial(p); delete_IntPolynomial_array(Value_l, r);
}
void ref_int_ifft(double *out /*[N/2][2]*/, const int32_t *poly) {
    IntPolynomial *p = new_IntPolynomial();
    LagrangeHalfCPolynomial *r = new_LagrangeHalfCPolynomial();
    memcpy(p->coefs, poly, sizeof(int32_t) * N_);
    IntPolynomial_ifft(r, p);
    memcpy(out, r->coefsC, sizeof(double) * N_);
    delete_LagrangeHalfCPolynomial(r); delete_IntPolynomial(p);
}
void ref_torus_ifft(double *out, const int32_t *poly) {
    TorusPolynomial *p = new_TorusPolynomial();
    LagrangeHalfCPolynomial *r = new_LagrangeHalfCPolynomial();
    memcpy(p->coefsT, poly, sizeof(int32_t) * N_);
    TorusPolynomial_ifft(r, p);
    memcpy(out, r->coefsC, sizeof(double) * N_);
    delete_LagrangeHalfCPolynomial(r); delete_TorusPolynomial(p);
}
void ref_torus_fft(int32_t *out, const double *spec /*[N/2][2]*/) {
    TorusPolynomial *p = new_TorusPolynomial();
    LagrangeHalfCPolynomial *r = new_LagrangeHalfCPolynomial();
    memcpy(r->coefsC, spec, sizeof(double) * N_);
    TorusPolynomial_fft(p, r);
    memcpy(out, p->coefsT, sizeof(int32_t) * N_);
    delete_LagrangeHalfCPolynomial(r); delete_TorusPolynomial(p);
}
void ref_mul_fft(int32_t *out, const int32_t *ipoly, const int32_t *tpoly) {
    IntPolynomial *a = new_IntPolynomial();
    TorusPolynomial *b = new_TorusPolynomial(), *r = new_TorusPolynomial();
    memcpy(a->coefs, ipoly, sizeof(int32_t) * N_);
    memcpy(b->coefsT, tpoly, sizeof(int32_t) * N_);
    torusPolynomialMultFFT(r, a, b);
    memcpy(out, r->coefsT, sizeof(int32_t) * N_);
    delete_TorusPolynomial(r); delete_TorusPolynomial(b); delete_IntPolynomial(a);
}
void ref_mul_naive(int32_t *out, const int32_t *ipoly, const int32_t *tpoly) {
    IntPolynomial *a = new_IntPolynomial();
    TorusPolynomial *b = new_TorusPolynomial(), *r = new_TorusPolynomial();
    memcpy(a->coefs, ipoly, sizeof(int32_t) * N_);
    memcpy(b->coefsT, tpoly, sizeof(int32_t) * N_);
    torusPolynomialMultNaive(r, a, b);
    memcpy(out, r->coefsT, sizeof(int32_t) * N_);
    delete_TorusPolynomial(r); delete_TorusPolynomial(b); delete_IntPolynomial(a);
}
void ref_mul_by_xai(int32_t *out, int32_t a, const int32_t *in) {
    TorusPolynomial *s = new_TorusPolynomial(), *r = new_TorusPolynomial();
    memcpy(s->coefsT, in, sizeof(int32_t) * N_);
    torusPolynomialMulByXai(r, a, s);
    memcpy(out, r->coefsT, sizeof(int32_t) * N_);
    delete_TorusPolynomial(r); delete_TorusPolynomial(s);
}
void ref_mul_by_xai_minus_one(int32_t *out, int32_t a, const int32_t *in) {
    TorusPolynomial *s = new_TorusPolynomial(), *r = new_TorusPolynomial();
    memcpy(s->coefsT, in, sizeof(int32_t) * N_);
    torusPolynomialMulByXaiMinusOne(r, a, s);
    memcpy(out, r->coefsT, sizeof(int32_t) * N_);
    delete_TorusPolynomial(r); delete_TorusPolynomial(s);
}
int32_t ref_modswitch_from(int32_t phase, int32_t Msize) { return modSwitchFromTorus32(phase, Msize); }
int32_t ref_modswitch_to(int32_t mu, int32_t Msize) { return modSwitchToTorus32(mu, Msize); }

}  // extern "C"

// ---- multi-threaded batch driver over the reference's bootsNAND... (cpu_baseline "reference" leg of bench.py) ----
#include <pthread.h>
namespace {
struct RefJob { int32_t gate; int32_t *out; const int32_t *a, *b, *c; long long i0, i1; };
void *ref_worker(void *arg) {
    RefJob *j = (RefJob *) arg;
    const long long s = n_ + 1;
    for (long long i = j->i0; i < j->i1; i++)
        ref_gate(j->gate, j->out + i * s, j->a ? j->a + i * s : nullptr, j->b ? j->b + i * s : nullptr,
                 j->c ? j->c + i * s : nullptr);
    return nullptr;
}
}
extern "C" void ref_gate_batch(int32_t gate, int32_t *out, const int32_t *a, const int32_t *b, const int32_t *c,
                               long long count, int32_t nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    RefJob jobs[256];
    for (int t = 0; t < nthreads; t++) {
        jobs[t] = RefJob{gate, out, a, b, c, count * t / nthreads, count * (t + 1) / nthreads};
        pthread_create(&th[t], nullptr, ref_worker, &jobs[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], nullptr);
}

// ---- the reference's own wire formats (tfhe_io.cpp), used to pin tfhe_amd/csrc/wire_io.cpp -----------------------
#include <cstdio>
extern "C" {
// export_tfheGateBootstrappingCloudKeySet_toFile (tfhe_io.cpp:1109) of the installed keys
int32_t ref_export_cloud_keyset(const char *path) {
    FILE *F = fopen(path, "wb");
    if (!F) return -1;
    export_tfheGateBootstrappingCloudKeySet_toFile(F, &g_keys->cloud);
    fclose(F);
    return 0;
}
// new_tfheGateBootstrappingCloudKeySet_fromFile (tfhe_io.cpp:1117), flattened; the key lives until the next call
static TFheGateBootstrappingCloudKeySet *g_cloud_in = nullptr;
int32_t ref_import_cloud_keyset(const char *path, int32_t *bk_coef, int32_t *ks) {
    FILE *F = fopen(path, "rb");
    if (!F) return -1;
    if (g_cloud_in) delete_gate_bootstrapping_cloud_keyset(g_cloud_in);
    g_cloud_in = new_tfheGateBootstrappingCloudKeySet_fromFile(F);
    fclose(F);
    const LweBootstrappingKey *bk = g_cloud_in->bk;
    for (int i = 0; i < n_; i++)
        for (int p = 0; p < kpl_; p++)
            for (int c = 0; c < 2; c++)
                memcpy(bk_coef + (((size_t) i * kpl_ + p) * 2 + c) * N_, bk->bk[i].all_sample[p].a[c].coefsT,
                       sizeof(int32_t) * N_);
    for (int i = 0; i < N_; i++)
        for (int j = 0; j < t_; j++)
            for (int d = 0; d < base_; d++)
                from_sample(ks + (((size_t) i * t_ + j) * base_ + d) * (n_ + 1), &bk->ks->ks[i][j][d], n_);
    return 0;
}
// a NAND on the imported cloud key (shows that a file written by the build is a working key for the reference)
void ref_imported_nand(int32_t *out, const int32_t *a, const int32_t *b) {
    LweSample *sa = new_LweSample(g_params->in_out_params), *sb = new_LweSample(g_params->in_out_params),
              *so = new_LweSample(g_params->in_out_params);
    to_sample(sa, a, n_);
    to_sample(sb, b, n_);
    bootsNAND(so, sa, sb, g_cloud_in);
    from_sample(out, so, n_);
    delete_LweSample(sa); delete_LweSample(sb); delete_LweSample(so);
}
// export_gate_bootstrapping_ciphertext_toFile (tfhe_io.cpp:1214) in a loop, variance[i] stored with each sample
int32_t ref_export_ciphertexts(const char *path, const int32_t *flat, const double *variance, int32_t count) {
    ensure_params();
    FILE *F = fopen(path, "wb");
    if (!F) return -1;
    LweSample *s = new_LweSample(g_params->in_out_params);
    for (int i = 0; i < count; i++) {
        to_sample(s, flat + (size_t) i * (n_ + 1), n_);
        s->current_variance = variance ? variance[i] : 0.;
        export_gate_bootstrapping_ciphertext_toFile(F, s, g_params);
    }
    delete_LweSample(s);
    fclose(F);
    return 0;
}
int32_t ref_import_ciphertexts(const char *path, int32_t *flat, double *variance, int32_t count) {
    ensure_params();
    FILE *F = fopen(path, "rb");
    if (!F) return -1;
    LweSample *s = new_LweSample(g_params->in_out_params);
    for (int i = 0; i < count; i++) {
        import_gate_bootstrapping_ciphertext_fromFile(F, s, g_params);
        from_sample(flat + (size_t) i * (n_ + 1), s, n_);
        if (variance) variance[i] = s->current_variance;
    }
    delete_LweSample(s);
    fclose(F);
    return 0;
}
int32_t ref_export_params(const char *path) {
    ensure_params();
    FILE *F = fopen(path, "wb");
    if (!F) return -1;
    export_tfheGateBootstrappingParameterSet_toFile(F, g_params);
    fclose(F);
    return 0;
}
}  // extern "C"

// ---- the 80-bit parameter set (n = 500, l = 2, Bgbit = 10; tfhe_gate_bootstrapping.cpp:23-48), stage by stage ------
// The fork's composed hot path hard-codes the default set (Value_n, Value_l, Value_Bgbit: tfhe_core.h:13-24), but these
// stages of the reference take their parameters at run time: lweKeySwitch (lwe-keyswitch-functions.cpp:114-129,228-238),
// the generic gadget decomposition Torus32PolynomialDecompH_old (tgsw-functions.cpp:277-293), tGswToFFTConvert
// (tgsw-fft-operations.cpp:52-57) and the transform-domain pieces of the external product (IntPolynomial_ifft,
// tLweFFTClear / tLweFFTAddMulRTo / tLweFromFFTConvert: k = 1 and N = 1024 in both sets).  The external product below is
// tGswFFTExternMulToTLwe (tgsw-fft-operations.cpp:92-120) line for line with the generic decomposition in place of the
// hard-wired one - every arithmetic operation is the reference's.
extern "C" void Torus32PolynomialDecompH_old(IntPolynomial *result, const TorusPolynomial *sample, const TGswParams *params);
static TFheGateBootstrappingParameterSet *g_p80 = nullptr;
static void ensure80() {
    if (!g_p80) g_p80 = new_default_gate_bootstrapping_parameters(80);
}
extern "C" {
void ref80_get_sizes(int32_t *out) {  // n, N, k, l, Bgbit, kpl, t, basebit
    ensure80();
    out[0] = g_p80->in_out_params->n;
    out[1] = g_p80->tgsw_params->tlwe_params->N;
    out[2] = g_p80->tgsw_params->tlwe_params->k;
    out[3] = g_p80->tgsw_params->l;
    out[4] = g_p80->tgsw_params->Bgbit;
    out[5] = g_p80->tgsw_params->kpl;
    out[6] = g_p80->ks_t;
    out[7] = g_p80->ks_basebit;
}
void ref80_decomp(int32_t *out /*[l][N]*/, const int32_t *poly /*N*/) {
    ensure80();
    const TGswParams *gp = g_p80->tgsw_params;
    IntPolynomial *r = new_IntPolynomial_array(gp->l);
    TorusPolynomial *p = new_TorusPolynomial();
    memcpy(p->coefsT, poly, sizeof(int32_t) * N_);
    Torus32PolynomialDecompH_old(r, p, gp);
    for (int q = 0; q < gp->l; q++) memcpy(out + q * N_, r[q].coefs, sizeof(int32_t) * N_);
    delete_TorusPolynomial(p); delete_IntPolynomial_array(gp->l, r);
}
// count keyswitches LWE_N -> LWE_500 with the key ks[N][t][base][n80+1]
void ref80_keyswitch(int32_t *out /*[count][n80+1]*/, const int32_t *in /*[count][N+1]*/, const int32_t *ks, int32_t count) {
    ensure80();
    const LweParams *op = g_p80->in_out_params;
    const int32_t n80 = op->n;
    LweKeySwitchKey *k = new_LweKeySwitchKey(N_, g_p80->ks_t, g_p80->ks_basebit, op);
    for (int i = 0; i < N_; i++)
        for (int j = 0; j < t_; j++)
            for (int d = 0; d < base_; d++)
                to_sample(&k->ks[i][j][d], ks + (((size_t) i * t_ + j) * base_ + d) * (n80 + 1), n80);
    LweSample *u = new_LweSample(&g_p80->tgsw_params->tlwe_params->extracted_lweparams);
    LweSample *r = new_LweSample(op);
    for (int c = 0; c < count; c++) {
        to_sample(u, in + (size_t) c * (N_ + 1), N_);
        lweKeySwitch(r, k, u);
        from_sample(out + (size_t) c * (n80 + 1), r, n80);
    }
    delete_LweSample(r); delete_LweSample(u); delete_LweKeySwitchKey(k);
}
// one TGSW sample, coefficient domain tgsw[kpl][2][N] -> bkfft_out[kpl][2][N/2][2] (tGswToFFTConvert), and
// acc[2][N] <- tgsw (*) acc
void ref80_extern_mul(int32_t *acc /*[2][N] in/out*/, double *bkfft_out /*may be null*/, const int32_t *tgsw) {
    ensure80();
    const TGswParams *gp = g_p80->tgsw_params;
    const TLweParams *tp = gp->tlwe_params;
    const int32_t k = tp->k, l = gp->l, kpl = gp->kpl;
    TGswSample *s = new_TGswSample(gp);
    for (int p = 0; p < kpl; p++) {
        for (int c = 0; c < 2; c++) memcpy(s->all_sample[p].a[c].coefsT, tgsw + ((size_t) p * 2 + c) * N_, sizeof(int32_t) * N_);
        s->all_sample[p].current_variance = 0.;
    }
    TGswSampleFFT *gsw = new_TGswSampleFFT(gp);
    tGswToFFTConvert(gsw, s, gp);
    if (bkfft_out)
        for (int p = 0; p < kpl; p++)
            for (int c = 0; c < 2; c++)
                memcpy(bkfft_out + ((size_t) p * 2 + c) * N_, gsw->all_samples[p].a[c].coefsC, sizeof(double) * N_);
    TLweSample *accum = new_TLweSample(tp);
    to_tlwe(accum, acc);
    // tgsw-fft-operations.cpp:92-120
    IntPolynomial *deca = new_IntPolynomial_array(kpl);
    LagrangeHalfCPolynomial *decaFFT = new_LagrangeHalfCPolynomial_array(kpl);
    TLweSampleFFT *tmpa = new_TLweSampleFFT(tp);
    for (int32_t i = 0; i <= k; i++) Torus32PolynomialDecompH_old(deca + i * l, accum->a + i, gp);   // :103-105, generic form
    for (int32_t p = 0; p < kpl; p++) IntPolynomial_ifft(&decaFFT[p], &deca[p]);                     // :106-108
    tLweFFTClear(tmpa);                                                                              // :110
    for (int32_t p = 0; p < kpl; p++) tLweFFTAddMulRTo(tmpa, decaFFT + p, gsw->all_samples + p);     // :112-114
    tLweFromFFTConvert(accum, tmpa);                                                                 // :115
    delete_TLweSampleFFT(tmpa);
    delete_LagrangeHalfCPolynomial_array(kpl, decaFFT);
    delete_IntPolynomial_array(kpl, deca);
    from_tlwe(acc, accum);
    delete_TLweSample(accum); delete_TGswSampleFFT(gsw); delete_TGswSample(s);
}
}  // extern "C"

// ---- leveled operations on arbitrary operands (default parameter set): tGswExternMulToTLwe (tgsw-functions.cpp:157-170),
// tGswFFTExternMulToTLwe with a caller-built TGswSampleFFT (tgsw-fft-operations.cpp:52-57,92-120), tLweAddMulRTo
// (tlwe-functions.cpp:191-197)
extern "C" {
static TGswSample *tgsw_from_flat(const int32_t *tgsw) {
    ensure_params();
    TGswSample *s = new_TGswSample(g_params->tgsw_params);
    for (int p = 0; p < kpl_; p++) {
        for (int c = 0; c < 2; c++) memcpy(s->all_sample[p].a[c].coefsT, tgsw + ((size_t) p * 2 + c) * N_, sizeof(int32_t) * N_);
        s->all_sample[p].current_variance = 0.;
    }
    return s;
}
void ref_tgsw_extern_mul(int32_t *acc /*[2][N] in/out*/, const int32_t *tgsw /*[kpl][2][N]*/) {
    TGswSample *s = tgsw_from_flat(tgsw);
    TLweSample *a = new_TLweSample(g_params->tgsw_params->tlwe_params);
    to_tlwe(a, acc);
    tGswExternMulToTLwe(a, s, g_params->tgsw_params);
    from_tlwe(acc, a);
    delete_TLweSample(a); delete_TGswSample(s);
}
void ref_tgsw_fft_extern_mul(int32_t *acc /*[2][N] in/out*/, double *gsw_fft_out /*[kpl][2][N/2][2]*/, const int32_t *tgsw) {
    TGswSample *s = tgsw_from_flat(tgsw);
    TGswSampleFFT *f = new_TGswSampleFFT(g_params->tgsw_params);
    tGswToFFTConvert(f, s, g_params->tgsw_params);
    for (int p = 0; p < kpl_; p++)
        for (int c = 0; c < 2; c++)
            memcpy(gsw_fft_out + ((size_t) p * 2 + c) * N_, f->all_samples[p].a[c].coefsC, sizeof(double) * N_);
    TLweSample *a = new_TLweSample(g_params->tgsw_params->tlwe_params);
    to_tlwe(a, acc);
    tGswFFTExternMulToTLwe(a, f, g_params->tgsw_params);
    from_tlwe(acc, a);
    delete_TLweSample(a); delete_TGswSampleFFT(f); delete_TGswSample(s);
}
void ref_tlwe_addmul_r(int32_t *result /*[2][N] in/out*/, const int32_t *poly /*N*/, const int32_t *sample /*[2][N]*/) {
    ensure_params();
    const TLweParams *tp = g_params->tgsw_params->tlwe_params;
    TLweSample *r = new_TLweSample(tp), *s = new_TLweSample(tp);
    IntPolynomial *p = new_IntPolynomial();
    to_tlwe(r, result);
    to_tlwe(s, sample);
    memcpy(p->coefs, poly, sizeof(int32_t) * N_);
    tLweAddMulRTo(r, p, s, tp);
    from_tlwe(result, r);
    delete_IntPolynomial(p); delete_TLweSample(s); delete_TLweSample(r);
}
}  // extern "C"

// ---- run-time N, l, Bgbit ----------------------------------------------------------------------------
// What the reference can run at other parameters than its compiled-in set: the generic gadget decomposition
// (Torus32PolynomialDecompH_old, tgsw-functions.cpp:277-293; its structs hold Value_N = 1024 coefficients, so N <= 1024) with
// a TGswParams of the caller's l and Bgbit (tgsw.cpp:7-27), and the exact schoolbook product torusPolynomialMultNaive_aux
// (multiplication.cpp:31-43) on raw arrays of any N.  Its composed external product goes through the hard-wired
// decomposition and the N = 1024 transform and cannot.
void torusPolynomialMultNaive_aux(Torus32 *__restrict result, const int32_t *__restrict poly1, const Torus32 *__restrict poly2,
                                  const int32_t N);
extern "C" {
int refn_decomp(int32_t *out /*[l][n]*/, const int32_t *poly /*n*/, int32_t n, int32_t l, int32_t Bgbit) {
    if (n > N_) return -1;
    TLweParams *tp = new_TLweParams(n, 1, 0., 1.);
    TGswParams *gp = new_TGswParams(l, Bgbit, tp);
    IntPolynomial *r = new_IntPolynomial_array(l);
    TorusPolynomial *p = new_TorusPolynomial();
    memset(p->coefsT, 0, sizeof(p->coefsT));
    memcpy(p->coefsT, poly, sizeof(int32_t) * n);
    Torus32PolynomialDecompH_old(r, p, gp);
    for (int q = 0; q < l; q++) memcpy(out + (size_t) q * n, r[q].coefs, sizeof(int32_t) * n);
    delete_TorusPolynomial(p); delete_IntPolynomial_array(l, r);
    delete_TGswParams(gp); delete_TLweParams(tp);
    return 0;
}
void refn_mul_naive(int32_t *out, const int32_t *ipoly, const int32_t *tpoly, int32_t n) {
    torusPolynomialMultNaive_aux(out, ipoly, tpoly, n);
}
}  // extern "C"

/* TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement, in plain C, of the gate-bootstrapping hot path of emwjacobson/tfhe (reference under
 * /root/reference, paths below are relative to it).  It is the checker for the HIP kernels in tfhe_amd/csrc:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it, and the product never does.
 *
 * Pinned: oracle/_ref/libtfhe_ref.so is the reference itself compiled in the authoring container
 * (oracle/Makefile `ref`); tests/test_oracle_vs_ref.py checks every function below against it BIT FOR BIT
 * (spectra as IEEE doubles, ciphertexts as int32), and tests/golden/*.npz holds reference outputs that
 * tests/test_oracle_golden.py replays where the reference is absent (the GPU box).
 *
 * Floating point contract: every double operation below is an individually rounded IEEE-754 binary64 operation
 * in exactly the order the reference's source performs it (build with -ffp-contract=off; the reference's own
 * flags, src/CMakeLists.txt:24, have no FMA either).  That is what makes bit-for-bit agreement possible.
 */
#include "tfhe_oracle.h"
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define N ORC_N
#define N2 ORC_2N
#define NH ORC_Ns2
#define LOG2N2 11

/* ---------------------------------------------------------------------------------------------------------
 * Twiddle tables.  The reference hard-codes them (src/include/tfhe_core.h:26 `trig_table_reverse`, :93
 * `trig_table_direct`, :160 `bit_reversed`).  Their values are double-precision libm cos/sin of the angle
 * folded into the first octant (so cos(pi/2 - x) and sin(x) are the same double); tests/test_oracle_vs_ref.py
 * checks the tables generated here against the reference header entry by entry.
 * cs[lv][j] = cos(2 pi j / size), sn[lv][j] = sin(2 pi j / size), size = 8 << lv, j < size/2.
 * --------------------------------------------------------------------------------------------------------- */
#define NLEVELS 9 /* sizes 8,16,...,2048 */
static double *g_cs[NLEVELS], *g_sn[NLEVELS];
static int32_t g_bitrev[N2];
static pthread_once_t g_once = PTHREAD_ONCE_INIT;

static double oct_cos(int j, int n) {
    const double w = 2.0 * M_PI;
    if (4 * j == n) return -0.0; /* the reference table stores -0.0 at the quarter turn */
    if (8 * j <= n) return cos(w * j / n);
    if (4 * j <= n) return sin(w * (n / 4 - j) / n);
    if (8 * j < 3 * n) return -sin(w * (j - n / 4) / n);
    return -cos(w * (n / 2 - j) / n);
}
static double oct_sin(int j, int n) {
    const double w = 2.0 * M_PI;
    if (8 * j < n) return sin(w * j / n);
    if (4 * j <= n) return cos(w * (n / 4 - j) / n);
    if (8 * j <= 3 * n) return cos(w * (j - n / 4) / n);
    return sin(w * (n / 2 - j) / n);
}
static void init_tables(void) {
    for (int lv = 0; lv < NLEVELS; lv++) {
        int size = 8 << lv;
        g_cs[lv] = (double *) malloc(sizeof(double) * (size / 2));
        g_sn[lv] = (double *) malloc(sizeof(double) * (size / 2));
        for (int j = 0; j < size / 2; j++) {
            g_cs[lv][j] = oct_cos(j, size);
            g_sn[lv][j] = oct_sin(j, size);
        }
    }
    for (int i = 0; i < N2; i++) {
        int r = 0;
        for (int b = 0; b < LOG2N2; b++)
            if (i & (1 << b)) r |= 1 << (LOG2N2 - 1 - b);
        g_bitrev[i] = r;
    }
}
static void ensure_tables(void) { pthread_once(&g_once, init_tables); }

int32_t oracle_trig_table_len(void) { return 2 * N2 - 8; }
void oracle_get_trig_tables(double *rev, double *dir) {
    /* reference layout: per size, groups of 4 cos then 4 sin; reverse table holds -sin */
    ensure_tables();
    int64_t off = 0;
    for (int lv = 0; lv < NLEVELS; lv++) {
        int size = 8 << lv;
        for (int j = 0; j < size / 2; j += 4)
            for (int q = 0; q < 4; q++) {
                dir[off + 2 * j + q] = g_cs[lv][j + q];
                dir[off + 2 * j + 4 + q] = g_sn[lv][j + q];
                rev[off + 2 * j + q] = g_cs[lv][j + q];
                rev[off + 2 * j + 4 + q] = -g_sn[lv][j + q];
            }
        off += size;
    }
}
void oracle_get_bit_reversed(int32_t *out) {
    ensure_tables();
    memcpy(out, g_bitrev, sizeof(g_bitrev));
}
uint64_t oracle_trig_table_hash(void) { /* FNV-1a over the bit patterns of cs/sn, level by level */
    ensure_tables();
    uint64_t h = 1469598103934665603ULL;
    for (int lv = 0; lv < NLEVELS; lv++)
        for (int pass = 0; pass < 2; pass++) {
            const double *t = pass ? g_sn[lv] : g_cs[lv];
            for (int j = 0; j < (4 << lv); j++) {
                uint64_t u;
                double v = t[j] == 0.0 ? 0.0 : t[j]; /* ignore the sign of zero */
                memcpy(&u, &v, 8);
                for (int b = 0; b < 8; b++) {
                    h ^= (u >> (8 * b)) & 0xff;
                    h *= 1099511628211ULL;
                }
            }
        }
    return h;
}

/* ---------------------------------------------------------------------------------------------------------
 * 2048-point radix-2 decimation-in-time FFT on split re/im arrays.
 * reference: src/libtfhe/polynomials.cpp:13-93 (fft_transform_reverse, kernel e^{+i theta}) and :95-175
 * (fft_transform, kernel e^{-i theta}).  dir = +1 reproduces the "reverse" routine, dir = -1 the "direct" one.
 * Stage order, and inside a butterfly the order  t = (re*c + im*s, im*c - re*s); lo += t; hi = lo_old - t,
 * are the reference's (:78-85).
 * --------------------------------------------------------------------------------------------------------- */
static void fft2048(double *re, double *im, int dir) {
    ensure_tables();
    for (int i = 0; i < N2; i++) { /* :16-28 */
        int j = g_bitrev[i];
        if (i < j) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (int i = 0; i < N2; i += 2) { /* size-2 merge :31-40 */
        double ar = re[i], ai = im[i];
        re[i] = ar + re[i + 1];
        im[i] = ai + im[i + 1];
        re[i + 1] = ar - re[i + 1];
        im[i + 1] = ai - im[i + 1];
    }
    for (int i = 0; i < N2; i += 4) { /* size-4 merge :43-63 / :125-145 */
        double ar = re[i], ai = im[i];
        re[i] = ar + re[i + 2];
        im[i] = ai + im[i + 2];
        re[i + 2] = ar - re[i + 2];
        im[i + 2] = ai - im[i + 2];
        double br = re[i + 1], bi = im[i + 1], cr = re[i + 3], ci = im[i + 3];
        if (dir > 0) { /* twiddle +i */
            re[i + 1] = br - ci; im[i + 1] = bi + cr;
            re[i + 3] = br + ci; im[i + 3] = bi - cr;
        } else { /* twiddle -i */
            re[i + 1] = br + ci; im[i + 1] = bi - cr;
            re[i + 3] = br - ci; im[i + 3] = bi + cr;
        }
    }
    for (int lv = 0; lv < NLEVELS; lv++) { /* general merges :68-92 */
        int size = 8 << lv, half = size >> 1;
        const double *cs = g_cs[lv], *sn = g_sn[lv];
        for (int base = 0; base < N2; base += size)
            for (int j = 0; j < half; j++) {
                int lo = base + j, hi = lo + half;
                double c = cs[j], s = dir > 0 ? -sn[j] : sn[j]; /* table entry as stored by the reference */
                double xr = re[hi], xi = im[hi];
                double tr = xr * c + xi * s;
                double ti = xi * c - xr * s;
                re[hi] = re[lo] - tr;
                im[hi] = im[lo] - ti;
                re[lo] += tr;
                im[lo] += ti;
            }
    }
}
void oracle_fft_reverse(double *re, double *im) { fft2048(re, im, +1); }
void oracle_fft_direct(double *re, double *im) { fft2048(re, im, -1); }

/* polynomials.cpp:347-362 */
void oracle_IntPolynomial_ifft(double *out, const int32_t *poly) {
    double re[N2], im[N2];
    for (int i = 0; i < N; i++) re[i] = poly[i] / 2.;
    for (int i = 0; i < N; i++) re[N + i] = -re[i];
    for (int i = 0; i < N2; i++) im[i] = 0;
    fft2048(re, im, +1);
    for (int i = 0; i < NH; i++) {
        out[2 * i] = re[2 * i + 1];
        out[2 * i + 1] = im[2 * i + 1];
    }
}
/* polynomials.cpp:364-376 */
void oracle_TorusPolynomial_ifft(double *out, const int32_t *poly) {
    double re[N2], im[N2];
    const double two_m33 = 1. / (double) (INT64_C(1) << 33);
    for (int i = 0; i < N; i++) re[i] = poly[i] * two_m33;
    for (int i = 0; i < N; i++) re[N + i] = -re[i];
    for (int i = 0; i < N2; i++) im[i] = 0;
    fft2048(re, im, +1);
    for (int i = 0; i < NH; i++) {
        out[2 * i] = re[2 * i + 1];
        out[2 * i + 1] = im[2 * i + 1];
    }
}
/* polynomials.cpp:378-393; note the truncation toward zero of the double->int64 conversion at :392 */
void oracle_TorusPolynomial_fft(int32_t *out, const double *spec) {
    double re[N2], im[N2];
    const double two_32 = (double) (INT64_C(1) << 32);
    const double inv_N = (double) 1 / (double) N;
    for (int i = 0; i < N; i++) re[2 * i] = 0, im[2 * i] = 0;
    for (int i = 0; i < NH; i++) {
        re[2 * i + 1] = spec[2 * i];
        im[2 * i + 1] = spec[2 * i + 1];
        re[N2 - 1 - 2 * i] = spec[2 * i];
        im[N2 - 1 - 2 * i] = -spec[2 * i + 1];
    }
    fft2048(re, im, -1);
    for (int i = 0; i < N; i++) out[i] = (int32_t) (uint32_t) (uint64_t) (int64_t) (re[i] * inv_N * two_32);
}

/* src/libtfhe/tgsw-functions.cpp:297-331 with the constants of src/include/tfhe_core.h:20-24 */
void oracle_decompH(int32_t *out, const int32_t *poly) {
    uint32_t offset = 0; /* sum of Bg/2 * 2^(32 - p*Bgbit), tgsw.cpp:21-27; 2164391936 for the default set */
    for (int p = 1; p <= ORC_l; p++) offset += (uint32_t) (1 << (ORC_Bgbit - 1)) << (32 - p * ORC_Bgbit);
    for (int p = 0; p < ORC_l; p++) {
        int shift = 32 - (p + 1) * ORC_Bgbit;
        for (int j = 0; j < N; j++) {
            uint32_t u = (uint32_t) poly[j] + offset;
            out[p * N + j] = (int32_t) ((u >> shift) & ((1u << ORC_Bgbit) - 1)) - (1 << (ORC_Bgbit - 1));
        }
    }
}

/* src/libtfhe/toruspolynomial-functions.cpp:128-147 ; all int32 arithmetic wraps mod 2^32 */
static inline int32_t neg32(int32_t x) { return (int32_t) (0u - (uint32_t) x); }
static inline int32_t sub32(int32_t x, int32_t y) { return (int32_t) ((uint32_t) x - (uint32_t) y); }
static inline int32_t add32(int32_t x, int32_t y) { return (int32_t) ((uint32_t) x + (uint32_t) y); }

void oracle_mulByXai(int32_t *out, int32_t a, const int32_t *in) {
    if (a < N) {
        for (int i = 0; i < a; i++) out[i] = neg32(in[i - a + N]);
        for (int i = a; i < N; i++) out[i] = in[i - a];
    } else {
        int aa = a - N;
        for (int i = 0; i < aa; i++) out[i] = in[i - aa + N];
        for (int i = aa; i < N; i++) out[i] = neg32(in[i - aa]);
    }
}
/* src/libtfhe/toruspolynomial-functions.cpp:106-124 */
void oracle_mulByXaiMinusOne(int32_t *out, int32_t a, const int32_t *in) {
    if (a < N) {
        for (int i = 0; i < a; i++) out[i] = sub32(neg32(in[i - a + N]), in[i]);
        for (int i = a; i < N; i++) out[i] = sub32(in[i - a], in[i]);
    } else {
        int aa = a - N;
        for (int i = 0; i < aa; i++) out[i] = sub32(in[i - aa + N], in[i]);
        for (int i = aa; i < N; i++) out[i] = sub32(neg32(in[i - aa]), in[i]);
    }
}

/* complex multiply as GCC evaluates std::complex<double> operator* for finite operands (C99 Annex G):
 * (ac - bd, ad + bc), four rounded products, two rounded sums.  polynomials.cpp:301-323 */
static inline void cmul(double *xr, double *xi, double ar, double ai, double br, double bi) {
    double ac = ar * br, bd = ai * bi, ad = ar * bi, bc = ai * br;
    *xr = ac - bd;
    *xi = ad + bc;
}

/* polynomials.cpp:241-248 */
void oracle_mulFFT(int32_t *out, const int32_t *ipoly, const int32_t *tpoly) {
    double a[N], b[N], c[N];
    oracle_IntPolynomial_ifft(a, ipoly);
    oracle_TorusPolynomial_ifft(b, tpoly);
    for (int i = 0; i < NH; i++) cmul(&c[2 * i], &c[2 * i + 1], a[2 * i], a[2 * i + 1], b[2 * i], b[2 * i + 1]);
    oracle_TorusPolynomial_fft(out, c);
}
/* src/libtfhe/multiplication.cpp:31-43 (N is a runtime argument there too) */
void oracle_mulNaive(int32_t *out, const int32_t *ipoly, const int32_t *tpoly, int32_t n) {
    for (int i = 0; i < n; i++) {
        uint32_t ri = 0;
        for (int j = 0; j <= i; j++) ri += (uint32_t) ipoly[j] * (uint32_t) tpoly[i - j];
        for (int j = i + 1; j < n; j++) ri -= (uint32_t) ipoly[j] * (uint32_t) tpoly[n + i - j];
        out[i] = (int32_t) ri;
    }
}

/* numeric-functions.cpp:59-65 and :71-76 */
int32_t oracle_modSwitchFromTorus32(int32_t phase, int32_t Msize) {
    uint64_t interv = ((UINT64_C(1) << 63) / (uint64_t) Msize) * 2;
    uint64_t half = interv / 2;
    uint64_t phase64 = ((uint64_t) (int64_t) phase << 32) + half;
    return (int32_t) (phase64 / interv);
}
int32_t oracle_modSwitchToTorus32(int32_t mu, int32_t Msize) {
    uint64_t interv = ((UINT64_C(1) << 63) / (uint64_t) Msize) * 2;
    uint64_t phase64 = (uint64_t) (int64_t) mu * interv;
    return (int32_t) (phase64 >> 32);
}

/* ---------------------------------------------------------------------------------------------------------
 * keys
 * --------------------------------------------------------------------------------------------------------- */
struct oracle_keys {
    double *bkfft; /* [n][kpl][2][N/2][2] */
    int32_t *ks;   /* [N][t][base][n+1]   */
};

/* tgsw-fft-operations.cpp:52-57 -> tlwe-fft-operations.cpp:55-61 -> TorusPolynomial_ifft */
void oracle_tgsw_to_fft(double *out, const int32_t *in) {
    for (int p = 0; p < ORC_kpl; p++)
        for (int c = 0; c < 2; c++) oracle_TorusPolynomial_ifft(out + (p * 2 + c) * N, in + (p * 2 + c) * N);
}

typedef struct { oracle_keys *k; const int32_t *bk; int i0, i1; } conv_job;
static void *conv_worker(void *arg) {
    conv_job *j = (conv_job *) arg;
    for (int i = j->i0; i < j->i1; i++)
        oracle_tgsw_to_fft(j->k->bkfft + (size_t) i * ORC_kpl * 2 * N, j->bk + (size_t) i * ORC_kpl * 2 * N);
    return NULL;
}

static oracle_keys *keys_alloc(const int32_t *ks) {
    oracle_keys *k = (oracle_keys *) malloc(sizeof(*k));
    size_t nks = (size_t) N * ORC_ks_t * ORC_ks_base * (ORC_n + 1);
    k->bkfft = (double *) malloc(sizeof(double) * ORC_n * ORC_kpl * 2 * N);
    k->ks = (int32_t *) malloc(sizeof(int32_t) * nks);
    memcpy(k->ks, ks, sizeof(int32_t) * nks);
    return k;
}
/* lwe-bootstrapping-functions-fft.cpp:24-53 (init_LweBootstrappingKeyFFT) */
oracle_keys *oracle_keys_from_coef(const int32_t *bk_coef, const int32_t *ks) {
    ensure_tables();
    oracle_keys *k = keys_alloc(ks);
    enum { T = 8 };
    pthread_t th[T];
    conv_job jobs[T];
    for (int t = 0; t < T; t++) {
        jobs[t].k = k; jobs[t].bk = bk_coef;
        jobs[t].i0 = ORC_n * t / T; jobs[t].i1 = ORC_n * (t + 1) / T;
        pthread_create(&th[t], NULL, conv_worker, &jobs[t]);
    }
    for (int t = 0; t < T; t++) pthread_join(th[t], NULL);
    return k;
}
oracle_keys *oracle_keys_from_fft(const double *bkfft, const int32_t *ks) {
    ensure_tables();
    oracle_keys *k = keys_alloc(ks);
    memcpy(k->bkfft, bkfft, sizeof(double) * ORC_n * ORC_kpl * 2 * N);
    return k;
}
void oracle_keys_get_bkfft(const oracle_keys *k, double *out) {
    memcpy(out, k->bkfft, sizeof(double) * ORC_n * ORC_kpl * 2 * N);
}
void oracle_keys_free(oracle_keys *k) {
    if (!k) return;
    free(k->bkfft);
    free(k->ks);
    free(k);
}

/* ---------------------------------------------------------------------------------------------------------
 * bootstrapping
 * --------------------------------------------------------------------------------------------------------- */
/* tgsw-fft-operations.cpp:92-120: decompose both polynomials, 6 forward transforms, accumulate rows p = 0..5
 * into a cleared spectrum pair (tlwe-fft-operations.cpp:80-94), 2 inverse transforms */
void oracle_externMulFFT(int32_t *acc, const double *gsw /*[kpl][2][N/2][2]*/) {
    int32_t deca[ORC_kpl * N];
    double decaFFT[ORC_kpl][N], tmpa[2][N];
    for (int c = 0; c < 2; c++) oracle_decompH(deca + c * ORC_l * N, acc + c * N);
    for (int p = 0; p < ORC_kpl; p++) oracle_IntPolynomial_ifft(decaFFT[p], deca + p * N);
    memset(tmpa, 0, sizeof(tmpa));
    for (int p = 0; p < ORC_kpl; p++)
        for (int c = 0; c < 2; c++) {
            const double *b = gsw + (p * 2 + c) * N;
            for (int f = 0; f < NH; f++) {
                double xr, xi;
                cmul(&xr, &xi, decaFFT[p][2 * f], decaFFT[p][2 * f + 1], b[2 * f], b[2 * f + 1]);
                tmpa[c][2 * f] += xr;
                tmpa[c][2 * f + 1] += xi;
            }
        }
    for (int c = 0; c < 2; c++) oracle_TorusPolynomial_fft(acc + c * N, tmpa[c]);
}
void oracle_externMul(const oracle_keys *k, int32_t *acc, int32_t i) {
    oracle_externMulFFT(acc, k->bkfft + (size_t) i * ORC_kpl * 2 * N);
}

/* tlwe-functions.cpp:191-197: result += p * sample, each column through torusPolynomialAddMulRFFT
 * (polynomials.cpp:249-259: product transformed back and truncated on its own, then added mod 2^32) */
void oracle_tLweAddMulRTo(int32_t *result /*[2][N]*/, const int32_t *p /*N*/, const int32_t *sample /*[2][N]*/) {
    int32_t tmpr[N];
    for (int c = 0; c < 2; c++) {
        oracle_mulFFT(tmpr, p, sample + c * N);
        for (int j = 0; j < N; j++) result[c * N + j] = add32(result[c * N + j], tmpr[j]);
    }
}
/* tgsw-functions.cpp:157-170 (tGswExternMulToTLwe; tGswExternProduct, :338-352, computes the same into another sample):
 * dec = tGswTLweDecompH(accum) (:264-270); accum = 0; accum += dec[i] * sample.all_sample[i] for the kpl rows in order */
void oracle_tgswExternMulToTLwe(int32_t *acc /*[2][N] in/out*/, const int32_t *tgsw /*[kpl][2][N]*/) {
    int32_t dec[ORC_kpl * N];
    for (int c = 0; c < 2; c++) oracle_decompH(dec + c * ORC_l * N, acc + c * N);
    memset(acc, 0, 2 * N * sizeof(int32_t));
    for (int i = 0; i < ORC_kpl; i++) oracle_tLweAddMulRTo(acc, dec + i * N, tgsw + (size_t) i * 2 * N);
}

/* lwe-bootstrapping-functions-fft.cpp:68-77 */
void oracle_muxRotate(const oracle_keys *k, int32_t *out, const int32_t *acc, int32_t i, int32_t barai) {
    for (int c = 0; c < 2; c++) oracle_mulByXaiMinusOne(out + c * N, barai, acc + c * N);
    oracle_externMul(k, out, i);
    for (int j = 0; j < 2 * N; j++) out[j] = add32(out[j], acc[j]);
}

/* lwe-bootstrapping-functions-fft.cpp:146-164 (the CPU loop; the FPGA kernel src/kernels/tfhe_blindRotate_FFT.cpp
 * :18-28 does not skip barai == 0, which changes nothing: the product is then exactly zero) */
void oracle_blindRotate(const oracle_keys *k, int32_t *acc, const int32_t *bara) {
    int32_t tmp[2 * N];
    for (int i = 0; i < ORC_n; i++) {
        if (bara[i] == 0) continue;
        oracle_muxRotate(k, tmp, acc, i, bara[i]);
        memcpy(acc, tmp, sizeof(tmp));
    }
}

/* src/libtfhe/lwe.cpp:41-53 with index 0 */
void oracle_extract(int32_t *out, const int32_t *acc) {
    out[0] = acc[0];
    for (int j = 1; j < N; j++) out[j] = neg32(acc[N - j]);
    out[N] = acc[N];
}

/* lwe-bootstrapping-functions-fft.cpp:222-253 and :180-209 */
void oracle_bootstrap_woKS(const oracle_keys *k, int32_t *out, const int32_t *in, int32_t mu) {
    int32_t bara[ORC_n], testvect[N], acc[2 * N];
    int32_t barb = oracle_modSwitchFromTorus32(in[ORC_n], N2);
    for (int i = 0; i < ORC_n; i++) bara[i] = oracle_modSwitchFromTorus32(in[i], N2);
    for (int i = 0; i < N; i++) testvect[i] = mu;
    memset(acc, 0, sizeof(int32_t) * N);
    if (barb != 0) oracle_mulByXai(acc + N, N2 - barb, testvect);
    else memcpy(acc + N, testvect, sizeof(testvect));
    oracle_blindRotate(k, acc, bara);
    oracle_extract(out, acc);
}

/* src/libtfhe/lwe-keyswitch-functions.cpp:228-238 and :114-129 */
void oracle_keyswitch(const oracle_keys *k, int32_t *out, const int32_t *in) {
    const int32_t prec_offset = 1 << (32 - (1 + ORC_ks_basebit * ORC_ks_t));
    memset(out, 0, sizeof(int32_t) * ORC_n);
    out[ORC_n] = in[N];
    for (int i = 0; i < N; i++) {
        uint32_t aibar = (uint32_t) in[i] + (uint32_t) prec_offset;
        for (int j = 0; j < ORC_ks_t; j++) {
            uint32_t aij = (aibar >> (32 - (j + 1) * ORC_ks_basebit)) & (ORC_ks_base - 1);
            if (aij != 0) {
                const int32_t *row = k->ks + (((size_t) i * ORC_ks_t + j) * ORC_ks_base + aij) * (ORC_n + 1);
                for (int q = 0; q <= ORC_n; q++) out[q] = sub32(out[q], row[q]);
            }
        }
    }
}

/* lwe-bootstrapping-functions-fft.cpp:266-279 */
void oracle_bootstrap(const oracle_keys *k, int32_t *out, const int32_t *in, int32_t mu) {
    int32_t u[N + 1];
    oracle_bootstrap_woKS(k, u, in, mu);
    oracle_keyswitch(k, out, u);
}

/* src/libtfhe/boot-gates.cpp: result = (0,cst) + sa*a + sb*b, then tfhe_bootstrap_FFT with MU = 1/8 */
typedef struct { int32_t cst_num, cst_den, sa, sb; } gate_lin;
static const gate_lin g_lin[10] = {
    /* NAND  :41-45   */ {1, 8, -1, -1},
    /* OR    :67-71   */ {1, 8, 1, 1},
    /* AND   :93-97   */ {-1, 8, 1, 1},
    /* XOR   :119-123 */ {1, 4, 2, 2},
    /* XNOR  :145-149 */ {-1, 4, -2, -2},
    /* NOR   :204-208 */ {-1, 8, -1, -1},
    /* ANDNY :230-234 */ {-1, 8, -1, 1},
    /* ANDYN :256-260 */ {-1, 8, 1, -1},
    /* ORNY  :282-286 */ {1, 8, -1, 1},
    /* ORYN  :308-312 */ {1, 8, 1, -1},
};
static void lincomb(int32_t *r, int32_t len, int32_t cst, int32_t sa, const int32_t *a, int32_t sb, const int32_t *b) {
    for (int q = 0; q <= len; q++)
        r[q] = (int32_t) ((uint32_t) sa * (uint32_t) a[q] + (uint32_t) sb * (uint32_t) b[q]);
    r[len] = add32(r[len], cst);
}

void oracle_gate(const oracle_keys *k, int32_t gate, int32_t *out, const int32_t *a, const int32_t *b,
                 const int32_t *c) {
    const int32_t MU = oracle_modSwitchToTorus32(1, 8);
    int32_t tmp[ORC_n + 1];
    if (gate >= ORC_NAND && gate <= ORC_ORYN) {
        const gate_lin *g = &g_lin[gate];
        lincomb(tmp, ORC_n, oracle_modSwitchToTorus32(g->cst_num, g->cst_den), g->sa, a, g->sb, b);
        oracle_bootstrap(k, out, tmp, MU);
    } else if (gate == ORC_MUX) { /* boot-gates.cpp:329-370 */
        int32_t u1[N + 1], u2[N + 1];
        const int32_t AndConst = oracle_modSwitchToTorus32(-1, 8);
        lincomb(tmp, ORC_n, AndConst, 1, a, 1, b);
        oracle_bootstrap_woKS(k, u1, tmp, MU);
        lincomb(tmp, ORC_n, AndConst, -1, a, 1, c);
        oracle_bootstrap_woKS(k, u2, tmp, MU);
        lincomb(u1, N, oracle_modSwitchToTorus32(1, 8), 1, u1, 1, u2);
        oracle_keyswitch(k, out, u1);
    } else if (gate == ORC_NOT) { /* :164-167 */
        for (int q = 0; q <= ORC_n; q++) out[q] = neg32(a[q]);
    } else if (gate == ORC_COPY) { /* :175-178 */
        memcpy(out, a, sizeof(int32_t) * (ORC_n + 1));
    } else { /* CONSTANT :185-189 */
        memset(out, 0, sizeof(int32_t) * ORC_n);
        out[ORC_n] = gate == ORC_CONST1 ? MU : neg32(MU);
    }
}

typedef struct {
    const oracle_keys *k; int32_t gate; int32_t *out; const int32_t *a, *b, *c; int64_t i0, i1;
} gate_job;
static void *gate_worker(void *arg) {
    gate_job *j = (gate_job *) arg;
    const int64_t s = ORC_n + 1;
    for (int64_t i = j->i0; i < j->i1; i++)
        oracle_gate(j->k, j->gate, j->out + i * s, j->a ? j->a + i * s : NULL, j->b ? j->b + i * s : NULL,
                    j->c ? j->c + i * s : NULL);
    return NULL;
}
void oracle_gate_batch(const oracle_keys *k, int32_t gate, int32_t *out, const int32_t *a, const int32_t *b,
                       const int32_t *c, int64_t count, int32_t nthreads) {
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    pthread_t th[256];
    gate_job jobs[256];
    for (int t = 0; t < nthreads; t++) {
        gate_job j = {k, gate, out, a, b, c, count * t / nthreads, count * (t + 1) / nthreads};
        jobs[t] = j;
        pthread_create(&th[t], NULL, gate_worker, &jobs[t]);
    }
    for (int t = 0; t < nthreads; t++) pthread_join(th[t], NULL);
}

/* src/libtfhe/lwe-functions.cpp:72-81 */
int32_t oracle_lwePhase(const int32_t *ct, const int32_t *key, int32_t n) {
    uint32_t axs = 0;
    for (int i = 0; i < n; i++) axs += (uint32_t) ct[i] * (uint32_t) key[i];
    return (int32_t) ((uint32_t) ct[n] - axs);
}

/* ---- the TGSW / TLWE layer with its parameters at run time ------------------------------------------
 * The reference's structs carry N, k, l, Bgbit (tgsw.cpp:7-27, tlwe.cpp); its generic functions loop over them.  Its
 * transform is compiled for N = 1024 only, so the products below are its EXACT schoolbook products
 * (torusPolynomialMultNaive_aux, multiplication.cpp:31-43, N a run-time argument): what any transform-based product must
 * equal up to its rounding (src/test/tgsw_fft_test.cpp:245-285 checks exactly this identity with Karatsuba). */
/* Torus32PolynomialDecompH_old, tgsw-functions.cpp:277-293, with offset / maskMod / halfBg of tgsw.cpp:7-27 */
void oracle_decompH_n(int32_t *out /*[l][N]*/, const int32_t *poly /*N*/, int32_t n, int32_t l, int32_t Bgbit) {
    const uint32_t halfBg = 1u << (Bgbit - 1), maskMod = (1u << Bgbit) - 1;
    uint32_t temp1 = 0;
    for (int i = 0; i < l; i++) temp1 += 1u << (32 - (i + 1) * Bgbit);
    const uint32_t offset = temp1 * halfBg;
    for (int j = 0; j < n; j++) {
        const uint32_t temp0 = (uint32_t) poly[j] + offset;
        for (int p = 0; p < l; p++) out[p * n + j] = (int32_t) (((temp0 >> (32 - (p + 1) * Bgbit)) & maskMod) - halfBg);
    }
}
/* tLweAddMulRTo, tlwe-functions.cpp:191-197: result[i] += p * sample[i], i <= k, every product exact */
void oracle_tLweAddMulRTo_n(int32_t *result /*[k+1][N]*/, const int32_t *p /*N*/, const int32_t *sample /*[k+1][N]*/, int32_t n,
                            int32_t k) {
    int32_t *prod = (int32_t *) malloc(sizeof(int32_t) * (size_t) n);
    for (int i = 0; i <= k; i++) {
        oracle_mulNaive(prod, p, sample + (size_t) i * n, n);
        for (int j = 0; j < n; j++) result[(size_t) i * n + j] = (int32_t) ((uint32_t) result[(size_t) i * n + j] + (uint32_t) prod[j]);
    }
    free(prod);
}
/* tGswExternMulToTLwe, tgsw-functions.cpp:157-170: decomposition of all k+1 polynomials (tGswTLweDecompH, :264-269: row
 * i*l + p = digit p of polynomial i), accumulator cleared, then the kpl row products added */
void oracle_tgswExternMulToTLwe_n(int32_t *acc /*[k+1][N] in/out*/, const int32_t *tgsw /*[(k+1)l][k+1][N]*/, int32_t n, int32_t k,
                                  int32_t l, int32_t Bgbit) {
    const int kpl = (k + 1) * l;
    int32_t *dec = (int32_t *) malloc(sizeof(int32_t) * (size_t) kpl * n);
    for (int i = 0; i <= k; i++) oracle_decompH_n(dec + (size_t) i * l * n, acc + (size_t) i * n, n, l, Bgbit);
    memset(acc, 0, sizeof(int32_t) * (size_t) (k + 1) * n);
    for (int i = 0; i < kpl; i++) oracle_tLweAddMulRTo_n(acc, dec + (size_t) i * n, tgsw + (size_t) i * (k + 1) * n, n, k);
    free(dec);
}

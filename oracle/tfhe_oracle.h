/* TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's gate-bootstrapping hot path (see tfhe_oracle.c).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library. */
#ifndef TFHE_ORACLE_H
#define TFHE_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* liboracle.so: the default 128-bit set (tfhe_gate_bootstrapping.cpp:51-75; the one this reference hard-codes in
 * src/include/tfhe_core.h:13-24).  liboracle_80.so (-DORC_LAMBDA=80): the 80-bit set of :23-48 - PINNED PER STAGE
 * (keyswitch, gadget decomposition, key conversion, external product: tests/golden/stages_80.npz holds the reference's
 * own outputs of its run-time-parameterised stage functions); the COMPOSED bootstrap of that set is unpinned, because
 * the reference's own composed hot path cannot run it (its n, l, Bgbit are compile-time constants). */
#ifndef ORC_LAMBDA
#define ORC_LAMBDA 128
#endif
#if ORC_LAMBDA == 128
enum { ORC_n = 630, ORC_l = 3, ORC_Bgbit = 7 };
#else
enum { ORC_n = 500, ORC_l = 2, ORC_Bgbit = 10 };
#endif
enum { ORC_N = 1024, ORC_2N = 2048, ORC_Ns2 = 512, ORC_k = 1, ORC_kpl = 2 * ORC_l,
       ORC_ks_t = 8, ORC_ks_basebit = 2, ORC_ks_base = 4 };

/* gate ids shared with include/tfhe_hip.h */
enum { ORC_NAND = 0, ORC_OR, ORC_AND, ORC_XOR, ORC_XNOR, ORC_NOR, ORC_ANDNY, ORC_ANDYN, ORC_ORNY, ORC_ORYN, ORC_MUX,
       ORC_NOT, ORC_COPY, ORC_CONST1, ORC_CONST0 };

typedef struct oracle_keys oracle_keys;

/* trig tables / bit reversal (for the table-equality test against the reference header) */
int32_t oracle_trig_table_len(void);
void oracle_get_trig_tables(double *rev, double *dir); /* in the reference's interleaved layout, 4088 doubles */
void oracle_get_bit_reversed(int32_t *out);            /* 2048 entries */
uint64_t oracle_trig_table_hash(void);

/* scalar helpers */
int32_t oracle_modSwitchFromTorus32(int32_t phase, int32_t Msize);
int32_t oracle_modSwitchToTorus32(int32_t mu, int32_t Msize);

/* polynomial primitives; spectra are double[N/2][2] (re,im) in the reference's LagrangeHalfC order */
void oracle_fft_reverse(double *re, double *im); /* 2048-point, in place */
void oracle_fft_direct(double *re, double *im);
void oracle_IntPolynomial_ifft(double *out, const int32_t *poly);
void oracle_TorusPolynomial_ifft(double *out, const int32_t *poly);
void oracle_TorusPolynomial_fft(int32_t *out, const double *spec);
void oracle_decompH(int32_t *out /*[l][N]*/, const int32_t *poly);
void oracle_mulByXai(int32_t *out, int32_t a, const int32_t *in);
void oracle_mulByXaiMinusOne(int32_t *out, int32_t a, const int32_t *in);
void oracle_mulFFT(int32_t *out, const int32_t *ipoly, const int32_t *tpoly);
void oracle_mulNaive(int32_t *out, const int32_t *ipoly, const int32_t *tpoly, int32_t N);

/* keys: bk_coef int32[n][kpl][2][N], bkfft double[n][kpl][2][N/2][2], ks int32[N][t][base][n+1] */
oracle_keys *oracle_keys_from_coef(const int32_t *bk_coef, const int32_t *ks);
oracle_keys *oracle_keys_from_fft(const double *bkfft, const int32_t *ks);
void oracle_keys_get_bkfft(const oracle_keys *k, double *out);
void oracle_keys_free(oracle_keys *k);
void oracle_tgsw_to_fft(double *out /*[kpl][2][N/2][2]*/, const int32_t *in /*[kpl][2][N]*/);

/* bootstrapping stages */
void oracle_externMul(const oracle_keys *k, int32_t *acc /*[2][N] in/out*/, int32_t i);
void oracle_externMulFFT(int32_t *acc /*[2][N] in/out*/, const double *gsw_fft /*[kpl][2][N/2][2]*/);
void oracle_tLweAddMulRTo(int32_t *result /*[2][N]*/, const int32_t *p /*N*/, const int32_t *sample /*[2][N]*/);
void oracle_tgswExternMulToTLwe(int32_t *acc /*[2][N] in/out*/, const int32_t *tgsw /*[kpl][2][N]*/);
void oracle_muxRotate(const oracle_keys *k, int32_t *out /*[2][N]*/, const int32_t *acc, int32_t i, int32_t barai);
void oracle_blindRotate(const oracle_keys *k, int32_t *acc /*[2][N] in/out*/, const int32_t *bara);
void oracle_extract(int32_t *out /*N+1*/, const int32_t *acc);
void oracle_bootstrap_woKS(const oracle_keys *k, int32_t *out /*N+1*/, const int32_t *in /*n+1*/, int32_t mu);
void oracle_keyswitch(const oracle_keys *k, int32_t *out /*n+1*/, const int32_t *in /*N+1*/);
void oracle_bootstrap(const oracle_keys *k, int32_t *out /*n+1*/, const int32_t *in /*n+1*/, int32_t mu);

/* gates (c may be NULL except for MUX; b may be NULL for unary gates) */
void oracle_gate(const oracle_keys *k, int32_t gate, int32_t *out, const int32_t *a, const int32_t *b,
                 const int32_t *c);
/* batch over `count` independent gates with `nthreads` worker threads (cpu_baseline driver) */
void oracle_gate_batch(const oracle_keys *k, int32_t gate, int32_t *out, const int32_t *a, const int32_t *b,
                       const int32_t *c, int64_t count, int32_t nthreads);

/* client-side helper for checks: phase = b - <a,s> */
int32_t oracle_lwePhase(const int32_t *ct, const int32_t *key, int32_t n);

/* run-time N, k, l, Bgbit: exact (schoolbook) products */
void oracle_decompH_n(int32_t *out /*[l][N]*/, const int32_t *poly /*N*/, int32_t N, int32_t l, int32_t Bgbit);
void oracle_tLweAddMulRTo_n(int32_t *result /*[k+1][N]*/, const int32_t *p /*N*/, const int32_t *sample /*[k+1][N]*/, int32_t N,
                            int32_t k);
void oracle_tgswExternMulToTLwe_n(int32_t *acc /*[k+1][N] in/out*/, const int32_t *tgsw /*[(k+1)l][k+1][N]*/, int32_t N, int32_t k,
                                  int32_t l, int32_t Bgbit);

#ifdef __cplusplus
}
#endif
#endif

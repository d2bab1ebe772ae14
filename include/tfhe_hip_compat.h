/* tfhe_hip_compat.h - struct-level libtfhe shim over the C-ABI of tfhe_hip.h.
 *
 * Same library (libtfhe_hip.so).  These entry points take the reference's own pointer-rich structs (LweSample arrays,
 * TFheGateBootstrappingCloudKeySet, TLweSample, TGswSampleFFT) so that code written against libtfhe can switch its hot
 * path without touching its data: they flatten the structs (as the reference itself does before its FPGA launch,
 * src/libtfhe/lwe-bootstrapping-functions-fft.cpp:101-127), call the batched C-ABI and scatter the results back.
 *
 * When the reference's headers are included first (TFHE_CORE_H defined) its struct definitions are used; otherwise the
 * layout-compatible views below are (member order and types copied from the cited headers; the reference's structs
 * are standard-layout, so the views are binary compatible - tests/test_compat.py checks sizes and offsets against the
 * reference headers in the authoring container).
 */
#ifndef TFHE_HIP_COMPAT_H
#define TFHE_HIP_COMPAT_H
#include <stdint.h>
#include "tfhe_hip.h"

#ifndef TFHE_CORE_H
typedef int32_t Torus32;                                  /* src/include/tfhe_core.h:206 */
typedef struct LweParams {                                /* src/include/lweparams.h */
    int32_t n; double alpha_min; double alpha_max;
} LweParams;
typedef struct LweSample {                                /* src/include/lwesamples.h:9-20 */
    Torus32 *a; Torus32 b; double current_variance;
} LweSample;
typedef struct LweKeySwitchKey {                          /* src/include/lwekeyswitch.h:11-28 */
    int32_t n, t, basebit, base; const LweParams *out_params;
    LweSample *ks0_raw; LweSample **ks1_raw; LweSample ***ks;
} LweKeySwitchKey;
typedef struct TLweParams {                               /* src/include/tlwe.h */
    int32_t N, k; double alpha_min, alpha_max; LweParams extracted_lweparams;
} TLweParams;
typedef struct TGswParams {                               /* src/include/tgsw.h */
    int32_t l, Bgbit, Bg, halfBg; uint32_t maskMod; const TLweParams *tlwe_params; int32_t kpl;
    Torus32 *h; uint32_t offset;
} TGswParams;
typedef struct TorusPolynomial { Torus32 coefsT[TFHE_HIP_N]; } TorusPolynomial;            /* polynomials.h:16-18 */
typedef struct LagrangeHalfCPolynomial { double coefsC[TFHE_HIP_N / 2][2]; } LagrangeHalfCPolynomial; /* :32-36 */
typedef struct TLweSample {                               /* src/include/tlwe.h:47-63 */
    TorusPolynomial *a; TorusPolynomial *b; double current_variance; int32_t k;
} TLweSample;
typedef struct TLweSampleFFT {
    LagrangeHalfCPolynomial *a; LagrangeHalfCPolynomial *b; double current_variance; int32_t k;
} TLweSampleFFT;
typedef struct TGswSampleFFT {                            /* src/include/tgsw.h:78-96 */
    TLweSampleFFT *all_samples; TLweSampleFFT **sample; int32_t k, l;
} TGswSampleFFT;
typedef struct LweBootstrappingKeyFFT {                   /* src/include/lwebootstrappingkey.h:36-59 */
    const LweParams *in_out_params; const TGswParams *bk_params; const TLweParams *accum_params;
    const LweParams *extract_params; const TGswSampleFFT *bkFFT; const LweKeySwitchKey *ks;
} LweBootstrappingKeyFFT;
typedef struct TFheGateBootstrappingParameterSet {        /* src/include/tfhe_gate_bootstrapping_structures.h */
    int32_t ks_t, ks_basebit; const LweParams *in_out_params; const TGswParams *tgsw_params;
} TFheGateBootstrappingParameterSet;
typedef struct TFheGateBootstrappingCloudKeySet {
    const TFheGateBootstrappingParameterSet *params; const void *bk; const LweBootstrappingKeyFFT *bkFFT;
} TFheGateBootstrappingCloudKeySet;
#endif /* TFHE_CORE_H */

#ifdef __cplusplus
extern "C" {
#endif

/* uploads bkFFT + ks of a reference cloud keyset (what bootsNAND reads through bk->bkFFT, boot-gates.cpp:49) */
int tfhe_hip_load_cloud_keyset(tfhe_hip_ctx *ctx, const TFheGateBootstrappingCloudKeySet *keyset);
/* count gates on arrays of reference LweSample structs (new_gate_bootstrapping_ciphertext_array layout) */
int tfhe_hip_gate_batch_samples(tfhe_hip_ctx *ctx, int gate, LweSample *result, const LweSample *a,
                                const LweSample *b, const LweSample *c, int32_t count);

/* the same over a device group (tfhe_hip.h): the arrays are split into contiguous shards, one per member */
int tfhe_hip_group_load_cloud_keyset(tfhe_hip_group *group, const TFheGateBootstrappingCloudKeySet *keyset);
int tfhe_hip_group_gate_batch_samples(tfhe_hip_group *group, int gate, LweSample *result, const LweSample *a,
                                      const LweSample *b, const LweSample *c, int32_t count);

/* libtfhe-style batched gates: bootsXXX of src/include/tfhe_gate_bootstrapping_functions.h:14-87 over `count`
 * independent gates.  They use a process-wide default device group - the GPUs listed in $TFHE_HIP_DEVICES ("0,1,2,3":
 * each call's gates are sharded over them), else the one GPU $TFHE_HIP_DEVICE (default 0) - and, like the reference's
 * die_dramatically (tfhe_gate_bootstrapping.cpp:11-15), abort on fatal errors.
 * KEY CACHING RULE.  The reference copies the whole bkFFT to its device on every call
 * (lwe-bootstrapping-functions-fft.cpp:112-133); here `bk` is uploaded on first use and kept.  Every call compares the
 * pointer AND a fingerprint of the key's contents (a hash over a sample of bkFFT and of the keyswitch key, about 0.1 ms)
 * with what was uploaded, and uploads again when either differs - a keyset freed and another created at the same
 * address, or bkFFT rewritten in place, is noticed.  tfhe_hip_default_context_reset() drops the cached key explicitly. */
void bootsNAND_batch(LweSample *result, const LweSample *ca, const LweSample *cb, int32_t count,
                     const TFheGateBootstrappingCloudKeySet *bk);
void bootsOR_batch(LweSample *result, const LweSample *ca, const LweSample *cb, int32_t count,
                   const TFheGateBootstrappingCloudKeySet *bk);
void bootsAND_batch(LweSample *result, const LweSample *ca, const LweSample *cb, int32_t count,
                    const TFheGateBootstrappingCloudKeySet *bk);
void bootsXOR_batch(LweSample *result, const LweSample *ca, const LweSample *cb, int32_t count,
                    const TFheGateBootstrappingCloudKeySet *bk);
void bootsXNOR_batch(LweSample *result, const LweSample *ca, const LweSample *cb, int32_t count,
                     const TFheGateBootstrappingCloudKeySet *bk);
void bootsNOR_batch(LweSample *result, const LweSample *ca, const LweSample *cb, int32_t count,
                    const TFheGateBootstrappingCloudKeySet *bk);
void bootsANDNY_batch(LweSample *result, const LweSample *ca, const LweSample *cb, int32_t count,
                      const TFheGateBootstrappingCloudKeySet *bk);
void bootsANDYN_batch(LweSample *result, const LweSample *ca, const LweSample *cb, int32_t count,
                      const TFheGateBootstrappingCloudKeySet *bk);
void bootsORNY_batch(LweSample *result, const LweSample *ca, const LweSample *cb, int32_t count,
                     const TFheGateBootstrappingCloudKeySet *bk);
void bootsORYN_batch(LweSample *result, const LweSample *ca, const LweSample *cb, int32_t count,
                     const TFheGateBootstrappingCloudKeySet *bk);
void bootsMUX_batch(LweSample *result, const LweSample *a, const LweSample *b, const LweSample *c, int32_t count,
                    const TFheGateBootstrappingCloudKeySet *bk);
void bootsNOT_batch(LweSample *result, const LweSample *ca, int32_t count, const TFheGateBootstrappingCloudKeySet *bk);
void bootsCOPY_batch(LweSample *result, const LweSample *ca, int32_t count, const TFheGateBootstrappingCloudKeySet *bk);
void bootsCONSTANT_batch(LweSample *result, int32_t value, int32_t count, const TFheGateBootstrappingCloudKeySet *bk);

/* the reference's device seam with its exact signature (src/include/tfhe.h:42): blind-rotates ONE accumulator on the
 * first GPU of the default group; `bk` is uploaded on first use and whenever its fingerprint changes (rule above) */
void tfhe_hip_blindRotate_FFT(TLweSample *accum, const TGswSampleFFT *bk, const int32_t *bara, const int32_t n,
                              const TGswParams *bk_params);
/* tfhe_bootstrap_FFT (with_ks != 0) / tfhe_bootstrap_woKS_FFT (with_ks == 0) of src/include/tfhe.h:44-45 on the
 * reference's structs: one sample, first GPU of the default group; key caching rule as above (owner = bk) */
void tfhe_hip_bootstrap_FFT(LweSample *result, const LweBootstrappingKeyFFT *bk, int32_t mu, const LweSample *x, int with_ks);
/* releases the default group and its cached key (optional; for tests and orderly shutdown) */
void tfhe_hip_default_context_reset(void);

#ifdef __cplusplus
}
#endif
#endif /* TFHE_HIP_COMPAT_H */

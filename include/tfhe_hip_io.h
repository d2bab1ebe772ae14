/* tfhe_hip_io.h - the reference's wire formats, read and written straight from/to the flat layouts of tfhe_hip.h
 * (SURVEY.md 8(f) N2).  Same library (libtfhe_hip.so); host code only, no GPU needed except for the two load calls.
 *
 * Formats (reference src/libtfhe/tfhe_io.cpp, src/libtfhe/tfhe_generic_streams.cpp):
 *   text section   "-----BEGIN <TITLE>-----\n" then "name: value\n" lines in lexicographic name order, integers as
 *                  "%10ld", doubles as "%.8lf", then "-----END <TITLE>-----\n"   (tfhe_generic_streams.cpp:46-56,166-173)
 *   parameter set  GATEBOOTSPARAMS{ks_basebit,ks_t} LWEPARAMS{alpha_max,alpha_min,n} TLWEPARAMS{N,alpha_max,alpha_min,k}
 *                  TGSWPARAMS{Bgbit,l}                                              (tfhe_io.cpp:1031-1035)
 *   cloud keyset   parameter set, LWEKSPARAMS{basebit,n,t}, then binary, little-endian, unaligned:
 *                  int32 200, double variance, int32[N][t][base][n+1]   keyswitch key   (tfhe_io.cpp:757-785)
 *                  int32 201, double variance, int32[n][kpl][k+1][N]    bootstrapping key (tfhe_io.cpp:883-906)
 *                  - exactly the `ks` and `bk_coef` layouts of tfhe_hip.h, so the payload is handed over as it lies.
 *                  The FFT key is not part of the format: the reference rebuilds it after reading (tfhe_io.cpp:1095),
 *                  here tfhe_hip_load_keys_coef does that on the GPU.
 *   ciphertext     int32 42, int32 a[n], int32 b, double current_variance = 2536 bytes   (tfhe_io.cpp:101-107,1214)
 *
 * Only the default 128-bit parameter set is accepted (tfhe_gate_bootstrapping.cpp:51-75); anything else is
 * TFHE_HIP_ERR_ARG with the offending field in tfhe_hip_last_error().  Malformed input never aborts (the reference
 * abort()s, tfhe_io.cpp:94, generic_streams.cpp:83).
 */
#ifndef TFHE_HIP_IO_H
#define TFHE_HIP_IO_H
#include "tfhe_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

#define TFHE_HIP_IO_SAMPLE_BYTES (4 + 4 * (TFHE_HIP_n + 1) + 8) /* 2536 for n = 630 */

/* ---- parameter set: new_tfheGateBootstrappingParameterSet_fromFile / export_..._toFile (tfhe_io.cpp:1050-1060) ---- */
/* checks that buf starts with the default parameter set; *consumed = bytes of the text sections */
int tfhe_hip_io_read_params(const void *buf, size_t len, size_t *consumed);
/* writes the default parameter set as the reference prints it; returns TFHE_HIP_ERR_ARG if cap is too small */
int tfhe_hip_io_write_params(void *buf, size_t cap, size_t *written);

/* ---- cloud keyset: new_tfheGateBootstrappingCloudKeySet_fromFile / export_..._toFile (tfhe_io.cpp:1087-1119) ----- */
/* upper bound of the serialized size (header text + both binary sections) */
size_t tfhe_hip_io_cloud_keyset_bytes(void);
/* bk_coef int32[630][6][2][1024], ks int32[1024][8][4][631] (caller-allocated); variances[2] = {keyswitch key,
 * bootstrapping key} as stored in the file (may be NULL) */
int tfhe_hip_io_read_cloud_keyset(const void *buf, size_t len, int32_t *bk_coef, int32_t *ks, double *variances,
                                  size_t *consumed);
int tfhe_hip_io_write_cloud_keyset(void *buf, size_t cap, size_t *written, const int32_t *bk_coef, const int32_t *ks,
                                   const double *variances);
int tfhe_hip_io_read_cloud_keyset_file(const char *path, int32_t *bk_coef, int32_t *ks, double *variances);
int tfhe_hip_io_write_cloud_keyset_file(const char *path, const int32_t *bk_coef, const int32_t *ks,
                                        const double *variances);
/* read + upload + on-GPU tGswToFFTConvert in one call: what an application does instead of
 * new_tfheGateBootstrappingCloudKeySet_fromFile followed by tfhe_hip_load_cloud_keyset */
int tfhe_hip_load_cloud_keyset_file(tfhe_hip_ctx *ctx, const char *path);
int tfhe_hip_load_cloud_keyset_buffer(tfhe_hip_ctx *ctx, const void *buf, size_t len);

/* ---- ciphertexts: import/export_gate_bootstrapping_ciphertext_{from,to}File (tfhe_io.cpp:1214-1226), `count`
 * samples back to back as a loop of those calls writes them ---------------------------------------------------- */
/* samples int32[count][631]; variances double[count] or NULL */
int tfhe_hip_io_read_samples(const void *buf, size_t len, int32_t *samples, double *variances, int64_t count,
                             size_t *consumed);
int tfhe_hip_io_write_samples(void *buf, size_t cap, size_t *written, const int32_t *samples, const double *variances,
                              int64_t count);

#ifdef __cplusplus
}
#endif
#endif /* TFHE_HIP_IO_H */

/* tfhe_hip.h - C-ABI of the MI355X gate-bootstrapping engine (libtfhe_hip.so).
 *
 * This is the drop-in boundary for the hot path of emwjacobson/tfhe (reference under /root/reference, cited as
 * file:line below).  In the reference the device seam is
 *     EXPORT void tfhe_blindRotate_FFT(TLweSample*, const TGswSampleFFT*, const int32_t* bara, int32_t n,
 *                                      const TGswParams*)         src/include/tfhe.h:42,
 *                                                                 src/libtfhe/lwe-bootstrapping-functions-fft.cpp:89
 * which flattens the structs into PODs and launches one FPGA kernel per ciphertext (src/kernels/
 * tfhe_blindRotate_FFT.cpp:36).  Here the seam is batched and widened to everything that runs per gate:
 * pre-combination, modulus switch, blind rotation, sample extraction and keyswitch.  Plain pointers and sizes
 * only; all functions return 0 on success and a negative code on error (message: tfhe_hip_last_error()).
 *
 * Default parameter set only (reference src/include/tfhe_core.h:13-24, tfhe_gate_bootstrapping.cpp:51-75):
 *   n = 630, N = 1024, k = 1, l = 3, Bgbit = 7, keyswitch t = 8, basebit = 2.
 *
 * Flat layouts (little-endian, C order; Torus32 = int32_t with wrap-around arithmetic):
 *   LWE_n sample   int32[n+1]  = a[0..n-1], b            (reference struct LweSample, lwesamples.h:9-20)
 *   LWE_N sample   int32[N+1]  = a[0..N-1], b            (extracted sample, before keyswitch)
 *   TLWE sample    int32[2][N] = a, b                     (reference TLweSample / TLweSample_FPGA, polynomials.h:45)
 *   spectrum       double[N/2][2] = LagrangeHalfCPolynomial.coefsC (re,im), P(w^(2i+1)), polynomials.h:32-36
 *   bk_coef        int32[n][6][2][N]      = bk[i].all_sample[p].a[c].coefsT   (order of tfhe_io.cpp:899-905)
 *   bkfft          double[n][6][2][N/2][2] = bkFFT[i].all_samples[p].a[c].coefsC
 *   ks             int32[N][t][4][n+1]    = ks[i][j][d] as a[0..n-1], b       (order of tfhe_io.cpp:777-784)
 *
 * Results are bit-identical to the reference CPU path (its own "optim" build flags, no FMA contraction): see
 * DESIGN.md "Arithmetic contract".
 */
#ifndef TFHE_HIP_H
#define TFHE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The parameter set is fixed at build time; one library per set (tfhe_gate_bootstrapping.cpp:23-75):
 *   libtfhe_hip.so     TFHE_HIP_LAMBDA 128 (default): n = 630, l = 3, Bgbit = 7
 *   libtfhe_hip_80.so  TFHE_HIP_LAMBDA 80:            n = 500, l = 2, Bgbit = 10
 * N = 1024, k = 1 and the keyswitch decomposition (t = 8, base 4) are common.  A host compiled against this header
 * defines TFHE_HIP_LAMBDA to match the library it links and can check with tfhe_hip_param_set(). */
#ifndef TFHE_HIP_LAMBDA
#define TFHE_HIP_LAMBDA 128
#endif
#if TFHE_HIP_LAMBDA == 128
#define TFHE_HIP_n 630
#define TFHE_HIP_L 3
#define TFHE_HIP_BGBIT 7
#elif TFHE_HIP_LAMBDA == 80
#define TFHE_HIP_n 500
#define TFHE_HIP_L 2
#define TFHE_HIP_BGBIT 10
#else
#error "TFHE_HIP_LAMBDA must be 128 or 80"
#endif
#define TFHE_HIP_N 1024
#define TFHE_HIP_KPL (2 * TFHE_HIP_L)
#define TFHE_HIP_KS_T 8
#define TFHE_HIP_KS_BASE 4

/* gate ids; the linear pre-combination of each is the one of reference src/libtfhe/boot-gates.cpp */
enum tfhe_hip_gate {
    TFHE_HIP_NAND = 0,   /* boot-gates.cpp:35  */
    TFHE_HIP_OR = 1,     /* :61  */
    TFHE_HIP_AND = 2,    /* :87  */
    TFHE_HIP_XOR = 3,    /* :113 */
    TFHE_HIP_XNOR = 4,   /* :139 */
    TFHE_HIP_NOR = 5,    /* :198 */
    TFHE_HIP_ANDNY = 6,  /* :224 */
    TFHE_HIP_ANDYN = 7,  /* :250 */
    TFHE_HIP_ORNY = 8,   /* :276 */
    TFHE_HIP_ORYN = 9,   /* :302 */
    TFHE_HIP_MUX = 10,   /* :329 two bootstraps without keyswitch + one keyswitch */
    TFHE_HIP_NOT = 11,   /* :164 no bootstrap */
    TFHE_HIP_COPY = 12,  /* :175 no bootstrap */
    TFHE_HIP_CONST1 = 13,/* :185 bootsCONSTANT(1) */
    TFHE_HIP_CONST0 = 14 /* :185 bootsCONSTANT(0) */
};

enum {
    TFHE_HIP_OK = 0,
    TFHE_HIP_ERR_ARG = -1,     /* bad argument (null pointer, negative count, unknown gate) */
    TFHE_HIP_ERR_HIP = -2,     /* a HIP runtime call failed */
    TFHE_HIP_ERR_NOKEY = -3,   /* keys not loaded */
    TFHE_HIP_ERR_TABLE = -4    /* twiddle-table self check failed (host libm differs) */
};

typedef struct tfhe_hip_ctx tfhe_hip_ctx;

/* One context per GPU (and per process in the one-process-per-GPU model).  Replaces the reference's global
 * `FPGA_Processor fpga` singleton (src/include/fpga.h:13-34, src/libtfhe/fpga.cpp:8-37): owns the stream, the
 * device copies of the keys and the batch workspaces. */
int tfhe_hip_create(tfhe_hip_ctx **out, int device);
void tfhe_hip_destroy(tfhe_hip_ctx *ctx);
const char *tfhe_hip_last_error(void);

/* Arithmetic of the FP64 stage, per context (default EXACT; $TFHE_HIP_ARITH=fast makes FAST the default of new
 * contexts).  The reference has no counterpart: its transform is fixed (src/libtfhe/polynomials.cpp:13-175).
 *   EXACT  every ciphertext equals the reference CPU path's bit for bit (the reference's 2048-point radix-2 transform,
 *          operation by operation, no FMA).
 *   FAST   the same negacyclic transforms in their minimal form (512 complex points in both directions, fused
 *          multiply-adds): decrypted bits identical, Torus32 coefficients within the tolerances of the reference's own
 *          tests - forward spectra |delta| <= 1e-10 (src/test/test-MulToTLwe.cpp:365-371), FFT products within 1 LSB of
 *          the exact product (src/test/test-multiplication.cpp:66-77), an external product and one CMux within 4 LSB of
 *          the reference's; a bootstrapped sample carries noise of the reference's distribution (same rms, unbiased)
 *          but not the reference's noise: after the first gadget digit that rounds differently the two accumulators
 *          decorrelate, as for any two FFT implementations (tests/test_fast_mode.py).  Keyswitch and all integer
 *          stages stay exact.  About 1.5x the gate throughput; a single gate in 2.1 ms instead of 3.0 ms.  Results are
 *          deterministic and independent of the position in the batch AND of the batch size: the latency kernel (batches
 *          of up to 768 bootstraps; 512 with the 80-bit set) accumulates the 2l row products in the throughput kernel's order, so a gate gives the
 *          same ciphertext in a batch of 1 and of 16,384, and a netlist the same pool at every level width.
 * The call waits for the context's stream; with keys loaded it derives the FAST key (a re-indexing of the same
 * doubles, 62 MB more device memory). */
enum { TFHE_HIP_ARITH_EXACT = 0, TFHE_HIP_ARITH_FAST = 1 };
int tfhe_hip_set_arithmetic(tfhe_hip_ctx *ctx, int mode);
int tfhe_hip_get_arithmetic(tfhe_hip_ctx *ctx);

/* Key upload.  Replaces the per-call copy of the whole bkFFT into the OpenCL buffer
 * (lwe-bootstrapping-functions-fft.cpp:112-133) by a one-time upload.
 * _coef : takes the coefficient-domain TGSW key and performs tGswToFFTConvert on the GPU
 *         (init_LweBootstrappingKeyFFT, lwe-bootstrapping-functions-fft.cpp:24-53; tgsw-fft-operations.cpp:52-57)
 * _fft  : takes an already converted key in the reference's LagrangeHalfC layout.
 * Host pointers; the caller may free its copies afterwards. */
int tfhe_hip_load_keys_coef(tfhe_hip_ctx *ctx, const int32_t *bk_coef, const int32_t *ks);
int tfhe_hip_load_keys_fft(tfhe_hip_ctx *ctx, const double *bkfft, const int32_t *ks);

/* Keys replicated by the caller's own transport - SURVEY.md 8(e): "once per key: RCCL ncclBroadcast of 61.9 MB + 82.7 MB" -
 * between contexts of different processes (one process per GPU).  The reference has no counterpart beyond the per-call
 * host-to-device copy of lwe-bootstrapping-functions-fft.cpp:112-133.
 * tfhe_hip_keys_device_buffers: the context's three device-resident key buffers (bootstrapping key in the FFT domain,
 *   keyswitch rows, keyswitch byte limbs), allocated if the context has none yet: ptrs[3], bytes[3].  On the rank that
 *   loaded keys they hold them; on the others they are the destination of the broadcast (byte for byte; all ranks run
 *   the same library build).
 * tfhe_hip_keys_adopt: declares the buffers filled (the transport must have completed) - derives what the context's
 *   arithmetic mode needs and marks the keys present.  Within one process use a device group instead. */
int tfhe_hip_keys_device_buffers(tfhe_hip_ctx *ctx, void *ptrs[3], size_t bytes[3]);
int tfhe_hip_keys_adopt(tfhe_hip_ctx *ctx);
/* 64-bit content fingerprints of the three key buffers (computed on the device; equal contents give equal values on
 * every device and rank): what the ranks compare after the broadcast, before anyone adopts. */
int tfhe_hip_keys_fingerprint(tfhe_hip_ctx *ctx, uint64_t out[3]);
/* device copy of the FFT key back in the reference layout (parity check of the conversion) */
int tfhe_hip_get_bkfft(tfhe_hip_ctx *ctx, double *bkfft_out);

/* ---- batched gates, host buffers in and out (PCIe inclusive) --------------------------------------------------
 * out[count][n+1]; a,b,c[count][n+1]; b/c may be NULL where the gate does not read them.
 * Semantics per element = bootsNAND/.../bootsMUX of boot-gates.cpp.  Batches above 4,096 gates run in chunks with the
 * copies of the neighbouring chunks overlapped on a second stream (tfhe_hip_last_kernel_ms then reports the last chunk). */
int tfhe_hip_gate_batch(tfhe_hip_ctx *ctx, int gate, int32_t *out, const int32_t *a, const int32_t *b,
                        const int32_t *c, int64_t count);
/* tfhe_bootstrap_FFT (with_ks != 0, out[count][n+1]; lwe-bootstrapping-functions-fft.cpp:266) or
 * tfhe_bootstrap_woKS_FFT (with_ks == 0, out[count][N+1]; :222) */
int tfhe_hip_bootstrap_batch(tfhe_hip_ctx *ctx, int32_t *out, const int32_t *in, int32_t mu, int64_t count,
                             int with_ks);
/* lweKeySwitch (lwe-keyswitch-functions.cpp:228): in[count][N+1] -> out[count][n+1] */
int tfhe_hip_keyswitch_batch(tfhe_hip_ctx *ctx, int32_t *out, const int32_t *in, int64_t count);

/* tGswTorus32PolynomialDecompH (tgsw-functions.cpp:297-331; the generic Torus32PolynomialDecompH_old, :277-293, with the
 * library's l and Bgbit): polys[count][N] -> out[count][l][N], digits in [-Bg/2, Bg/2).  The stage the CMux applies to the
 * rotated differences, on its own for parity checks and leveled circuits. */
int tfhe_hip_tgsw_decomp_batch(tfhe_hip_ctx *ctx, int32_t *out, const int32_t *polys, int64_t count);

/* ---- the same with every buffer already resident in device memory; asynchronous on the context's stream ---- */
int tfhe_hip_gate_batch_dev(tfhe_hip_ctx *ctx, int gate, int32_t *d_out, const int32_t *d_a, const int32_t *d_b,
                            const int32_t *d_c, int64_t count);
int tfhe_hip_bootstrap_batch_dev(tfhe_hip_ctx *ctx, int32_t *d_out, const int32_t *d_in, int32_t mu, int64_t count,
                                 int with_ks);
int tfhe_hip_keyswitch_batch_dev(tfhe_hip_ctx *ctx, int32_t *d_out, const int32_t *d_in, int64_t count);
/* tfhe_hip_gate_batch_dev for callers that cannot allocate (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  Same arguments, semantics and results as
 * tfhe_hip_gate_batch_dev, but the work buffers (extracted samples, keyswitch workspace) are slices of a static
 * device array of the library (about 21 MB per GPU, held until the process ends) instead of growable allocations:
 * the call performs no hipMalloc / hipFree.  count > TFHE_HIP_NOALLOC_MAX_GATES is rejected with TFHE_HIP_ERR_ARG.
 * All calls on one GPU share that array: calls that overlap on one GPU (two contexts, two streams) race. */
#define TFHE_HIP_NOALLOC_MAX_GATES 1024
int tfhe_hip_gate_batch_dev_noalloc(tfhe_hip_ctx *ctx, int gate, int32_t *d_out, const int32_t *d_a, const int32_t *d_b,
                                    const int32_t *d_c, int64_t count);
int tfhe_hip_sync(tfhe_hip_ctx *ctx);
/* raw stream handle (hipStream_t) so that callers can record events on the stream the kernels run on */
void *tfhe_hip_stream(tfhe_hip_ctx *ctx);
/* device time of the kernels of the most recent *_batch / *_batch_dev call, from HIP events on the stream
 * (blind rotation incl. pre-combination and extraction; keyswitch).  Synchronises the stream. */
int tfhe_hip_last_kernel_ms(tfhe_hip_ctx *ctx, float *blind_rotate_ms, float *keyswitch_ms);

/* ---- one level of a netlist on a device-resident ciphertext pool ----------------------------
 * The reference evaluates circuits as sequential loops of bootsXXX calls (src/test/test-addition-boot.cpp:30-115);
 * here all gates of one dependency level run as one batch, inputs gathered from and outputs scattered to rows of
 * d_pool (int32[pool_rows][n+1] in device memory).
 * gates: HOST array int32[count][5] = {gate id, out row, a row, b row, c row}; unused operands are ignored.
 * A level reads the pool as it was before the call: a row may not be written twice, nor read and written, within
 * one call (rejected with TFHE_HIP_ERR_ARG before anything is launched).  Returns after the level has completed. */
int tfhe_hip_gate_level_dev(tfhe_hip_ctx *ctx, int32_t *d_pool, int64_t pool_rows, const int32_t *gates,
                            int64_t count);
/* device-memory helpers, so that C callers of the *_dev entry points need no HIP of their own */
int tfhe_hip_dev_alloc(tfhe_hip_ctx *ctx, size_t bytes, void **out);
int tfhe_hip_dev_free(tfhe_hip_ctx *ctx, void *ptr);
int tfhe_hip_copy_to_dev(tfhe_hip_ctx *ctx, void *dst, const void *src, size_t bytes);
int tfhe_hip_copy_from_dev(tfhe_hip_ctx *ctx, void *dst, const void *src, size_t bytes);

/* ---- stage-level entry points (host buffers), for parity tests and the FFT microbench ----------------------- */
/* tfhe_blindRotate_FFT (the reference seam itself): acc[count][2][N] in/out, bara[count][n] in [0,2N) */
int tfhe_hip_blind_rotate_batch(tfhe_hip_ctx *ctx, int32_t *acc, const int32_t *bara, int64_t count);
/* tGswFFTExternMulToTLwe (tgsw-fft-operations.cpp:92): acc[count][2][N] in/out, key_index[count] in [0,n) */
int tfhe_hip_extern_mul_batch(tfhe_hip_ctx *ctx, int32_t *acc, const int32_t *key_index, int64_t count);
/* IntPolynomial_ifft / TorusPolynomial_ifft / TorusPolynomial_fft (polynomials.cpp:347,364,378) */
int tfhe_hip_int_ifft_batch(tfhe_hip_ctx *ctx, double *spectra, const int32_t *polys, int64_t count);
int tfhe_hip_torus_ifft_batch(tfhe_hip_ctx *ctx, double *spectra, const int32_t *polys, int64_t count);
int tfhe_hip_torus_fft_batch(tfhe_hip_ctx *ctx, int32_t *polys, const double *spectra, int64_t count);
/* torusPolynomialMultFFT (polynomials.cpp:241): out = ipoly * tpoly mod X^N+1 */
int tfhe_hip_negacyclic_mul_batch(tfhe_hip_ctx *ctx, int32_t *out, const int32_t *ipolys, const int32_t *tpolys,
                                  int64_t count);
/* device-resident variant used by bench.py's FFT microbench leg */
int tfhe_hip_negacyclic_mul_batch_dev(tfhe_hip_ctx *ctx, int32_t *d_out, const int32_t *d_ipolys,
                                      const int32_t *d_tpolys, int64_t count);

/* ---- leveled TGSW / TLWE operations on caller-supplied operands ---------------------------------
 * acc[count][2][N] in/out; operand g of the call is entry index[g] of the operand array (ntgsw entries), so that many
 * accumulators can share a few TGSW samples.  Results in the context's arithmetic mode (EXACT: the reference's bit for bit).
 * tGswFFTExternMulToTLwe (tgsw-fft-operations.cpp:92-120): gsw_fft double[ngsw][2l][2][N/2][2] = TGswSampleFFT
 *   .all_samples[p].a[c].coefsC (what tGswToFFTConvert, :52-57, = tfhe_hip_torus_ifft_batch per polynomial, produces). */
int tfhe_hip_tgsw_fft_extern_mul_batch(tfhe_hip_ctx *ctx, int32_t *acc, const double *gsw_fft, const int32_t *gsw_index,
                                       int64_t ngsw, int64_t count);
/* tGswExternMulToTLwe (tgsw-functions.cpp:157-170; tGswExternProduct, :338-352, is the same computation into another
 * sample): tgsw int32[ntgsw][2l][2][N] = TGswSample.all_sample[p].a[c].coefsT, coefficient domain; every (row, column)
 * product goes through torusPolynomialAddMulRFFT (polynomials.cpp:249-259) as in the reference. */
int tfhe_hip_tgsw_extern_mul_batch(tfhe_hip_ctx *ctx, int32_t *acc, const int32_t *tgsw, const int32_t *tgsw_index,
                                   int64_t ntgsw, int64_t count);
/* tLweAddMulRTo (tlwe-functions.cpp:191-197): result[g] += poly[g] * sample[g]; result, sample int32[count][2][N],
 * poly int32[count][N] (an IntPolynomial) */
int tfhe_hip_tlwe_addmul_r_batch(tfhe_hip_ctx *ctx, int32_t *result, const int32_t *poly, const int32_t *sample,
                                 int64_t count);

/* the same product at N = 2048 (BASELINE configs[4]): polys int32[count][2048].  The reference has no transform for
 * N = 2048 (tfhe_core.h:14); the product is composed of four N = 1024 products (even/odd split) and agrees with the
 * reference's naive product torusPolynomialMultNaive_aux (multiplication.cpp:31-43) within 2 LSB for |ipoly| < 2N */
int tfhe_hip_negacyclic_mul2048_batch(tfhe_hip_ctx *ctx, int32_t *out, const int32_t *ipolys, const int32_t *tpolys,
                                      int64_t count);
int tfhe_hip_negacyclic_mul2048_batch_dev(tfhe_hip_ctx *ctx, int32_t *d_out, const int32_t *d_ipolys,
                                          const int32_t *d_tpolys, int64_t count);

/* ---- TGSW / TLWE operations with N, k and the gadget (l, Bgbit) as run-time parameters ------------------
 * The reference's TGSW / TLWE layer reads N, k, l, Bgbit from its parameter structs (TGswParams / TLweParams, tgsw.cpp:7-27;
 * its own tests run N = 512 / 1024 / 2048, k = 1 / 2, (l, Bgbit) = (4, 8) / (3, 10): src/test/tgsw_fft_test.cpp:28-35).
 * Here: N in {512, 1024, 2048} - one real N/2-point folded transform per polynomial at every N, no composition of smaller
 * products -, k in {1, 2}, any gadget with l * Bgbit <= 32; the same in both arithmetic modes (this fork compiles the
 * reference's transform for N = 1024 only, tfhe_core.h:14, so there is no other-N arithmetic to be bit-identical to).
 * Parity: the reference's exact products (torusPolynomialMultNaive_aux, multiplication.cpp:31-43, N at run time) and its
 * generic decomposition (Torus32PolynomialDecompH_old, tgsw-functions.cpp:277-293) - products within 1 LSB (2 at N = 2048)
 * as src/test/test-multiplication.cpp:66-77, external products within a few LSB of the exact sum of row products
 * (the identity src/test/tgsw_fft_test.cpp:245-285 checks), decomposition exact.  Layouts as above with N, k, l general:
 * TLWE sample int32[k+1][N]; TGSW sample int32[(k+1)l][k+1][N] (row i*l + p = digit p of polynomial i, tgsw-functions.cpp
 * :264-269); its FFT form double[(k+1)l][k+1][N/2][2] = coefsC (P(w^(2i+1)), w = exp(i pi / N)). */
/* torusPolynomialMultFFT (polynomials.cpp:241) at ring dimension N */
int tfhe_hip_negacyclic_mul_n_batch(tfhe_hip_ctx *ctx, int32_t N, int32_t *out, const int32_t *ipolys, const int32_t *tpolys,
                                    int64_t count);
int tfhe_hip_negacyclic_mul_n_batch_dev(tfhe_hip_ctx *ctx, int32_t N, int32_t *d_out, const int32_t *d_ipolys,
                                        const int32_t *d_tpolys, int64_t count);
/* IntPolynomial_ifft / TorusPolynomial_ifft / TorusPolynomial_fft (polynomials.cpp:347,364,378) at ring dimension N;
 * tGswToFFTConvert (tgsw-fft-operations.cpp:52-57) = tfhe_hip_torus_ifft_n_batch over the (k+1)l(k+1) polynomials */
int tfhe_hip_int_ifft_n_batch(tfhe_hip_ctx *ctx, int32_t N, double *spectra, const int32_t *polys, int64_t count);
int tfhe_hip_torus_ifft_n_batch(tfhe_hip_ctx *ctx, int32_t N, double *spectra, const int32_t *polys, int64_t count);
int tfhe_hip_torus_fft_n_batch(tfhe_hip_ctx *ctx, int32_t N, int32_t *polys, const double *spectra, int64_t count);
/* Torus32PolynomialDecompH_old (tgsw-functions.cpp:277-293): polys[count][N] -> out[count][l][N] (any N >= 1) */
int tfhe_hip_tgsw_decomp_n_batch(tfhe_hip_ctx *ctx, int32_t N, int32_t l, int32_t Bgbit, int32_t *out, const int32_t *polys,
                                 int64_t count);
/* tGswFFTExternMulToTLwe (tgsw-fft-operations.cpp:92-120) / tGswExternMulToTLwe (tgsw-functions.cpp:157-170):
 * acc[g] <- gsw[index[g]] (*) acc[g], acc int32[count][k+1][N] in/out */
int tfhe_hip_tgsw_fft_extern_mul_n_batch(tfhe_hip_ctx *ctx, int32_t N, int32_t k, int32_t l, int32_t Bgbit, int32_t *acc,
                                         const double *gsw_fft, const int32_t *gsw_index, int64_t ngsw, int64_t count);
int tfhe_hip_tgsw_extern_mul_n_batch(tfhe_hip_ctx *ctx, int32_t N, int32_t k, int32_t l, int32_t Bgbit, int32_t *acc,
                                     const int32_t *tgsw, const int32_t *tgsw_index, int64_t ntgsw, int64_t count);
/* tLweAddMulRTo (tlwe-functions.cpp:191-197): result[g] += poly[g] * sample[g]; result, sample int32[count][k+1][N] */
int tfhe_hip_tlwe_addmul_r_n_batch(tfhe_hip_ctx *ctx, int32_t N, int32_t k, int32_t *result, const int32_t *poly,
                                   const int32_t *sample, int64_t count);

/* ---- compiled netlist (levels without host round-trips) ---------------------------------------
 * A whole levelized circuit over a device-resident pool: `gates` holds the levels back to back ({gate, out, a, b, c}
 * rows as for tfhe_hip_gate_level_dev), level_counts[l] the number of gates of level l.  Every level is validated
 * against the level contract at creation, the job lists are uploaded once, and the launches of all levels are
 * recorded into one hipGraph bound to d_pool (if the runtime cannot capture them, run enqueues them one by one -
 * tfhe_hip_netlist_is_graph tells which).  tfhe_hip_netlist_run is asynchronous on the context's stream
 * (tfhe_hip_sync waits); it may be called any number of times, e.g. after new inputs were written to the pool.
 * A netlist is tied to its context and pool; destroy it before either.  Runs of one context are stream-ordered.
 * The reference has no counterpart: its applications call bootsXXX gate by gate (src/test/test-addition-boot.cpp). */
typedef struct tfhe_hip_netlist tfhe_hip_netlist;
int tfhe_hip_netlist_create(tfhe_hip_ctx *ctx, tfhe_hip_netlist **out, int32_t *d_pool, int64_t pool_rows,
                            const int32_t *gates, const int64_t *level_counts, int64_t nlevels);
int tfhe_hip_netlist_run(tfhe_hip_netlist *netlist);
int tfhe_hip_netlist_is_graph(const tfhe_hip_netlist *netlist);
void tfhe_hip_netlist_destroy(tfhe_hip_netlist *netlist);

/* ---- device group: several GPUs behind one handle ------------------------------------------------
 * The reference drives one accelerator through one global object (src/include/fpga.h:34) and has no multi-device code.
 * A group holds one context, stream and host thread per listed device (a device may be listed more than once: that
 * many contexts share it).  Keys are uploaded once, to the first member, and replicated device to device; a batch is
 * split into contiguous shards, member i taking gates [i*count/ndev, (i+1)*count/ndev) up to rounding, with no
 * exchange between the members (gates are independent).  Results equal the single-context call's bit for bit.
 * tfhe_hip_group_gate_batch returns when every member has finished; errors name the failing member. */
typedef struct tfhe_hip_group tfhe_hip_group;
int tfhe_hip_group_create(tfhe_hip_group **out, const int *devices, int ndev);
void tfhe_hip_group_destroy(tfhe_hip_group *group);
int tfhe_hip_group_size(const tfhe_hip_group *group);
tfhe_hip_ctx *tfhe_hip_group_ctx(tfhe_hip_group *group, int member);   /* borrowed; e.g. for the *_dev entry points */
int tfhe_hip_group_load_keys_coef(tfhe_hip_group *group, const int32_t *bk_coef, const int32_t *ks);
int tfhe_hip_group_load_keys_fft(tfhe_hip_group *group, const double *bkfft, const int32_t *ks);
int tfhe_hip_group_set_arithmetic(tfhe_hip_group *group, int mode);
/* One level of a netlist over the group (gates as for tfhe_hip_gate_level_dev; the same level contract, checked over the
 * whole level): d_pools[member] = that member's replica of the ciphertext pool, int32[pool_rows][n+1] in ITS device memory
 * (tfhe_hip_dev_alloc on tfhe_hip_group_ctx(group, member)), all replicas holding the same rows on entry.  The gates are
 * split into contiguous blocks; every member evaluates its block and the rows it wrote are copied device to device into
 * all other replicas, so that all replicas are equal again on return (and equal to a single context's pool, bit for bit).
 * A level that violates the contract is refused before any member runs; after a run-time error (a HIP failure on one
 * member) the replicas may differ: restore them from one that the caller trusts.
 * The reference has no counterpart (one device, one gate per call). */
int tfhe_hip_group_gate_level(tfhe_hip_group *group, int32_t *const *d_pools, int64_t pool_rows, const int32_t *gates,
                              int64_t count);
/* how member `member` received its keys: 0 uploaded from the host (member 0), 1 device-to-device copy with peer access
 * enabled (xGMI where the GPUs are linked), 2 peer copy staged by the runtime (no peer access offered), 3 copy within one
 * device (the member shares member 0's GPU).  Replicas are verified against member 0's fingerprints before use. */
int tfhe_hip_group_key_path(const tfhe_hip_group *group, int member);
int tfhe_hip_group_gate_batch(tfhe_hip_group *group, int gate, int32_t *out, const int32_t *a, const int32_t *b,
                              const int32_t *c, int64_t count);

/* the parameter set this library was built for: out = {lambda, n, N, k, l, Bgbit, ks_t, ks_basebit} */
void tfhe_hip_param_set(int32_t out[8]);

#define TFHE_HIP_CLIENT_RNG_IS_NOT_CRYPTOGRAPHIC 1
/* ---- client-side bulk operations (the part that parallelises) ------------------------------
 * SECURITY NOTE.  Everything in this section (and tfhe_amd/csrc/client_keys.c, whose streams it reproduces) draws keys,
 * masks and noise from xoshiro256** streams derived from ONE 64-bit seed.  That generator is linear and invertible:
 * whoever sees a ciphertext's public mask words can rewind to its noise.  These entry points are REPRODUCIBLE FIXTURE
 * GENERATORS for tests, benchmarks and demos - the same bytes on every machine from a seed - and must not be used to
 * protect real data.  A deployment generates keys and ciphertexts with the reference's own client functions
 * (new_random_gate_bootstrapping_secret_keyset, bootsSymEncrypt; tfhe_gate_bootstrapping.cpp:97-159) over a
 * cryptographic generator and hands the cloud key to tfhe_hip_load_keys_* / tfhe_hip_load_cloud_keyset.
 *
 * bootsSymEncrypt (tfhe_gate_bootstrapping.cpp:154-159: mu = +-1/8, noise 2^-15) of `count` bits under the secret
 * LWE key (host int32[n], key bits 0/1), and lwePhase (lwe-functions.cpp:63-72; bootsSymDecrypt = phase > 0).
 * The reference's generator (std::default_random_engine + normal_distribution) is not portable, so these reproduce
 * the portable generator of tfhe_amd/csrc/client_keys.c bit for bit: sample i of a call is a function of
 * (seed, first_index + i, bit, key) - equal to tfhe_client_encrypt for first_index = 0.  The DISTRIBUTIONS are the
 * reference's: uniform masks and key bits, and noise of its law - gaussian32 (numeric-functions.cpp:22-34): a centred
 * Gaussian of the parameter set's standard deviation truncated toward zero to a Torus32 - drawn by integer table
 * inversion (tfhe_amd/csrc/gauss_cdt.h: 1,024 steps per sigma, tail to 9.16 sigma) and checked against the reference's
 * own gaussian32 (tests/test_oracle_vs_ref.py, tests/test_client.py).  Parity with the reference's streams is
 * semantic (its own lwePhase decrypts these samples; tests/test_gpu_parity.py). */
int tfhe_hip_encrypt_batch(tfhe_hip_ctx *ctx, int32_t *out, uint64_t seed, int64_t first_index, const int32_t *lwe_key,
                           const uint8_t *bits, int64_t count);
/* the same with the ciphertexts written to device memory (e.g. rows of a netlist pool); lwe_key and bits stay host */
int tfhe_hip_encrypt_batch_dev(tfhe_hip_ctx *ctx, int32_t *d_out, uint64_t seed, int64_t first_index,
                               const int32_t *lwe_key, const uint8_t *bits, int64_t count);
/* tfhe_createLweBootstrappingKey (lwe-bootstrapping-functions.cpp:185-217) on the GPU: draws bk[i] = TGSW(s_i) for
 * the given secret keys (host int32[n], int32[N], bits 0/1) with the portable generator (stream 0x10000 + i of `seed`,
 * as tfhe_client_keygen), transforms it and installs it with the keyswitch key `ks` (host, layout of
 * tfhe_hip_load_keys_coef; NULL keeps the keyswitch key already on the device, see tfhe_hip_keygen_cloud).
 * bk_coef_out (host int32[n][2l][2][N], may be NULL) receives the coefficient form, equal to tfhe_client_keygen's bit
 * for bit.  The secret keys are wiped from device memory before returning, on every exit path. */
int tfhe_hip_keygen_bk(tfhe_hip_ctx *ctx, uint64_t seed, const int32_t *lwe_key, const int32_t *tlwe_key,
                       const int32_t *ks, int32_t *bk_coef_out);
/* the whole cloud keyset on the GPU: tfhe_hip_keygen_bk plus lweCreateKeySwitchKey (lwe-keyswitch-functions.cpp:163-215),
 * both bit-identical to tfhe_client_keygen for the same seed and secret keys.  The client draws the 24,576 keyswitch
 * samples from one sequential stream; each GPU thread jumps to its sample's position with precomputed powers of the
 * generator's GF(2) transition matrix.  bk_coef_out / ks_out (host, layouts of tfhe_hip_load_keys_coef) may be NULL. */
int tfhe_hip_keygen_cloud(tfhe_hip_ctx *ctx, uint64_t seed, const int32_t *lwe_key, const int32_t *tlwe_key,
                          int32_t *bk_coef_out, int32_t *ks_out);
int tfhe_hip_phase_batch(tfhe_hip_ctx *ctx, int32_t *phases, const int32_t *lwe_key, const int32_t *samples,
                         int64_t count);

#ifdef __cplusplus
}
#endif
#endif /* TFHE_HIP_H */

// libtfhe_hip_gates.so: the reference's OWN single-gate entry points - bootsNAND ... bootsMUX, bootsNOT, bootsCOPY,
// bootsCONSTANT (src/include/tfhe_gate_bootstrapping_functions.h:44-87, defined in src/libtfhe/boot-gates.cpp), the
// bootstraps tfhe_bootstrap_FFT / tfhe_bootstrap_woKS_FFT and the device seam tfhe_blindRotate_FFT
// (src/include/tfhe.h:42-45) - with their exact names and signatures, each forwarding
// to the batched shim with a batch of one ("single-gate bootsNAND etc. forward to batch of 1").
//
// A separate library on purpose: these names are libtfhe's.  An application that still links libtfhe for key
// generation, encryption and I/O puts this library BEFORE libtfhe on its link line, or starts with
//     LD_PRELOAD=libtfhe_hip_gates.so ./application
// and every gate it evaluates runs on the GPU with no change to its source (2.9 ms per gate, 2.1 ms with
// TFHE_HIP_ARITH=fast, against 65-100 ms on a host core); the cloud key is uploaded on the first gate (key caching
// rule of tfhe_hip_compat.h).  Throughput needs the batched calls: this is the functional drop-in.
#include "tfhe_hip_compat.h"

#define TFHE_HIP_EXPORT extern "C" __attribute__((visibility("default")))

#define BINARY_GATE(NAME)                                                                                              \
    TFHE_HIP_EXPORT void boots##NAME(LweSample *result, const LweSample *ca, const LweSample *cb,                     \
                                     const TFheGateBootstrappingCloudKeySet *bk) {                                     \
        boots##NAME##_batch(result, ca, cb, 1, bk);                                                                    \
    }
BINARY_GATE(NAND)   // boot-gates.cpp:35
BINARY_GATE(OR)     // :61
BINARY_GATE(AND)    // :87
BINARY_GATE(XOR)    // :113
BINARY_GATE(XNOR)   // :139
BINARY_GATE(NOR)    // :198
BINARY_GATE(ANDNY)  // :224
BINARY_GATE(ANDYN)  // :250
BINARY_GATE(ORNY)   // :276
BINARY_GATE(ORYN)   // :302

TFHE_HIP_EXPORT void bootsMUX(LweSample *result, const LweSample *a, const LweSample *b, const LweSample *c,
                              const TFheGateBootstrappingCloudKeySet *bk) {  // boot-gates.cpp:329
    bootsMUX_batch(result, a, b, c, 1, bk);
}
TFHE_HIP_EXPORT void bootsNOT(LweSample *result, const LweSample *ca, const TFheGateBootstrappingCloudKeySet *bk) {  // :164
    bootsNOT_batch(result, ca, 1, bk);
}
TFHE_HIP_EXPORT void bootsCOPY(LweSample *result, const LweSample *ca, const TFheGateBootstrappingCloudKeySet *bk) {  // :175
    bootsCOPY_batch(result, ca, 1, bk);
}
TFHE_HIP_EXPORT void bootsCONSTANT(LweSample *result, int32_t value, const TFheGateBootstrappingCloudKeySet *bk) {  // :185
    bootsCONSTANT_batch(result, value, 1, bk);
}
// lwe-bootstrapping-functions-fft.cpp:89
TFHE_HIP_EXPORT void tfhe_blindRotate_FFT(TLweSample *accum, const TGswSampleFFT *bk, const int32_t *bara, const int32_t n,
                                          const TGswParams *bk_params) {
    tfhe_hip_blindRotate_FFT(accum, bk, bara, n, bk_params);
}
// lwe-bootstrapping-functions-fft.cpp:266 and :222 - the whole bootstrap (modulus switch, blind rotation, extraction, and
// for the first one the keyswitch) instead of the seam alone
TFHE_HIP_EXPORT void tfhe_bootstrap_FFT(LweSample *result, const LweBootstrappingKeyFFT *bk, int32_t mu, const LweSample *x) {
    tfhe_hip_bootstrap_FFT(result, bk, mu, x, 1);
}
TFHE_HIP_EXPORT void tfhe_bootstrap_woKS_FFT(LweSample *result, const LweBootstrappingKeyFFT *bk, int32_t mu,
                                             const LweSample *x) {
    tfhe_hip_bootstrap_FFT(result, bk, mu, x, 0);
}

// TEST INFRASTRUCTURE ONLY - builds the *reference itself* (emwjacobson/tfhe) as a CPU library inside the
// authoring container, so that the C restatement in oracle/tfhe_oracle.c and the golden fixtures in tests/golden/
// can be pinned against the real thing.  Nothing in the product (tfhe_amd/) links or loads this.
//
// The reference sources are compiled where they lie (/root/reference/src/libtfhe/*.cpp, see oracle/Makefile); this
// translation unit only supplies the two pieces of the fork that cannot run without a Xilinx FPGA:
//
//  1. `FPGA_Processor fpga` (reference src/libtfhe/fpga.cpp:8-37): the global whose constructor opens an OpenCL
//     platform and exits when no xclbin is found.  Here it only fills `omegaxminus1` (fpga.cpp:11-15), the single
//     member that CPU code reads (polynomials.cpp:294-298).  The cl:: members default-construct to null handles.
//  2. `tfhe_blindRotate_FFT` (reference src/libtfhe/lwe-bootstrapping-functions-fft.cpp:89-165): the fork replaced
//     the CPU loop by an OpenCL launch and left the original loop as a comment at :146-164.  The loop below is that
//     commented loop, calling the reference's own `tfhe_MuxRotate_FFT` (:68-77), which is compiled from the
//     reference file through its own TFHE_TEST_ENVIRONMENT / INCLUDE_* switches (:6-18,21,80,169,213,257).
//
// Every arithmetic routine (FFT, decomposition, external product, keyswitch, gates) is therefore reference code.
#include <iostream>
#include <cassert>
#include <cmath>
#include <vector>
#include <string>
#include "tfhe.h"
#include "fpga.h"

using namespace std;

FPGA_Processor::FPGA_Processor() {
    omegaxminus1 = (cplx *) malloc(sizeof(cplx) * Value_2N);
    for (int32_t x = 0; x < Value_2N; x++)
        omegaxminus1[x] = cplx(cos(x * M_PI / Value_N) - 1., sin(x * M_PI / Value_N));
}
FPGA_Processor::~FPGA_Processor() { free(omegaxminus1); }
std::vector<cl::Device> FPGA_Processor::get_xilinx_devices() { return std::vector<cl::Device>(); }
std::vector<unsigned char> FPGA_Processor::read_binary_file(const std::string &) { return std::vector<unsigned char>(); }
FPGA_Processor fpga;

#define TFHE_TEST_ENVIRONMENT
#define INCLUDE_TFHE_INIT_LWEBOOTSTRAPPINGKEY_FFT
#define INCLUDE_TFHE_BLIND_ROTATE_AND_EXTRACT_FFT
#define INCLUDE_TFHE_BOOTSTRAP_WO_KS_FFT
#define INCLUDE_TFHE_BOOTSTRAP_FFT
#include "lwe-bootstrapping-functions-fft.cpp"   // found through -I$(REF)/src/libtfhe

extern "C" void tfhe_blindRotate_FFT(TLweSample *accum, const TGswSampleFFT *bkFFT, const int32_t *bara,
                                     const int32_t n, const TGswParams *bk_params) {
    TLweSample *temp = new_TLweSample(bk_params->tlwe_params);
    TLweSample *temp2 = temp;
    TLweSample *temp3 = accum;
    for (int32_t i = 0; i < n; i++) {
        const int32_t barai = bara[i];
        if (barai == 0) continue;
        tfhe_MuxRotate_FFT(temp2, temp3, bkFFT + i, barai, bk_params);
        swap(temp2, temp3);
    }
    if (temp3 != accum) tLweCopy(accum, temp3, bk_params->tlwe_params);
    delete_TLweSample(temp);
}

// exported so the harness can exercise one CMux step of the reference (it has external linkage in the
// reference file but no header declaration)
extern "C" void ref_shim_MuxRotate_FFT(TLweSample *result, const TLweSample *accum, const TGswSampleFFT *bki,
                                       const int32_t barai, const TGswParams *bk_params) {
    tfhe_MuxRotate_FFT(result, accum, bki, barai, bk_params);
}

// Micro-benchmark: issue cost of FP64 vector instructions on gfx950 (cycles per wave instruction), one and two waves
// per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o fp64_rate fp64_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#pragma clang fp contract(off)

template <int OP>
__global__ void k(double *out, unsigned long long *cyc, double a0, double b0, int iters) {
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = a0 + i * 0.001 + threadIdx.x * 1e-6;
    const double b = b0, c = b0 * 0.5;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) x[i] = x[i] + b;
                if (OP == 1) x[i] = x[i] * b;
                if (OP == 2) x[i] = __builtin_fma(x[i], b, c);
                if (OP == 3) { x[i] = x[i] * b; x[i] = x[i] + c; }        // mul then dependent add (two instructions)
                if (OP == 4) x[i] = (double) (int) x[i] + b;               // cvt_i32_f64 + cvt_f64_i32 + add
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
}

template <int OP>
void run(const char *name, int insts_per_iter, int waves_per_block) {
    const int iters = 2000, blocks = 256;
    double *out;
    unsigned long long *cyc;
    hipMalloc(&out, sizeof(double) * blocks * 64 * waves_per_block);
    hipMalloc(&cyc, 8 * blocks * waves_per_block);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, out, cyc, 1.0, 1.0000001, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * waves_per_block);
    hipMemcpy(h.data(), cyc, 8 * h.size(), hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double) v;
    const double per_wave = sum / h.size() / ((double) iters * insts_per_iter);
    printf("%-28s waves/SIMD %d: %.2f cycles per instruction per wave  -> %.2f cycles of SIMD time per instruction\n", name,
           waves_per_block / 4, per_wave, per_wave / (waves_per_block / 4));
    hipFree(out);
    hipFree(cyc);
}

// dependent chains: NCH independent accumulators, each a serial chain of adds
template <int NCH>
__global__ void kdep(double *out, unsigned long long *cyc, double a0, double b0, int iters) {
    double x[NCH];
#pragma unroll
    for (int i = 0; i < NCH; i++) x[i] = a0 + i * 0.001 + threadIdx.x * 1e-6;
    const double b = b0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 128 / NCH; u++) {
#pragma unroll
            for (int i = 0; i < NCH; i++) x[i] = x[i] * b;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NCH; i++) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
}
template <int NCH>
void rundep(int waves_per_block) {
    const int iters = 2000, blocks = 256;
    double *out;
    unsigned long long *cyc;
    hipMalloc(&out, sizeof(double) * blocks * 64 * waves_per_block);
    hipMalloc(&cyc, 8 * blocks * waves_per_block);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(kdep<NCH>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, out, cyc, 1.0, 1.0000001, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * waves_per_block);
    hipMemcpy(h.data(), cyc, 8 * h.size(), hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double) v;
    const double per_wave = sum / h.size() / ((double) iters * 128);
    printf("v_mul_f64, %d independent chains, waves/SIMD %d: %.2f cycles per instruction per wave -> %.2f of SIMD time\n", NCH,
           waves_per_block / 4, per_wave, per_wave / (waves_per_block / 4));
    hipFree(out);
    hipFree(cyc);
}

int main() {
    for (int w : {4, 8, 12, 16}) {
        rundep<1>(w);
        rundep<2>(w);
        rundep<4>(w);
    }
    for (int w : {4, 8}) {
        run<0>("v_add_f64", 128, w);
        run<1>("v_mul_f64", 128, w);
        run<2>("v_fma_f64", 128, w);
        run<3>("v_mul_f64 + v_add_f64 pair", 256, w);
        run<4>("cvt pair + add", 384, w);
    }
    return 0;
}

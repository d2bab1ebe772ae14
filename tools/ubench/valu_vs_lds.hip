// Micro-benchmark: does LDS traffic of one wave slow the FP64 issue of the other wave on the same SIMD?
// Block = 8 waves (2 per SIMD): waves 0-3 run a dependent-free v_mul_f64 stream, waves 4-7 either idle (mode 0),
// stream ds_write_b128 + ds_read_b128 through LDS (mode 1: 16 KB per 16 instructions), or issue ds_read_b128 only (2).
// Prints cycles per FP64 instruction of the arithmetic waves.  hipcc --offload-arch=gfx950 -O3 -o valu_vs_lds valu_vs_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#pragma clang fp contract(off)

struct alignas(16) q4 { double a, b; };

__global__ __launch_bounds__(512) void k(double *out, unsigned long long *cyc, double b0, int iters, int mode) {
    __shared__ q4 lds[8 * 512];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wave < 4) {
        double x[8];
#pragma unroll
        for (int i = 0; i < 8; i++) x[i] = 1.0 + i * 0.001 + lane * 1e-6;
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 16; u++)
#pragma unroll
                for (int i = 0; i < 8; i++) x[i] = x[i] * b0;
        }
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        double s = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) s += x[i];
        out[blockIdx.x * 256 + wave * 64 + lane] = s;
        if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
    } else if (mode != 0) {
        q4 *mine = lds + wave * 512;
        q4 v[8];
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = {1.0 + i, 2.0 + lane};
        for (int it = 0; it < iters * 3; it++) {
            if (mode == 1) {
#pragma unroll
                for (int i = 0; i < 8; i++) mine[i * 64 + lane] = v[i];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const q4 t = mine[i * 64 + ((lane + it) & 63)];
                v[i].a += t.b;
                v[i].b = t.a;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
        }
        double s = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) s += v[i].a + v[i].b;
        if (s == 12345.678) out[0] = s;
    }
}

int main() {
    const int iters = 2000, blocks = 256;
    double *out;
    unsigned long long *cyc;
    hipMalloc(&out, sizeof(double) * blocks * 256);
    hipMalloc(&cyc, 8 * blocks * 4);
    for (int mode = 0; mode < 3; mode++) {
        for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, out, cyc, 1.0000001, iters, mode);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * 4);
        hipMemcpy(h.data(), cyc, 8 * h.size(), hipMemcpyDeviceToHost);
        double sum = 0;
        for (auto v : h) sum += (double) v;
        printf("mode %d (%s): %.2f cycles per v_mul_f64 in the arithmetic wave\n", mode,
               mode == 0 ? "partner wave idle" : mode == 1 ? "partner wave writes+reads LDS, b128" : "partner wave reads LDS, b128",
               sum / h.size() / ((double) iters * 128));
    }
    return 0;
}

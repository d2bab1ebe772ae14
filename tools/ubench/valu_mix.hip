// Micro-benchmark: issue cost (cycles per wave instruction) of the non-FMA VALU instructions of the CMux on gfx950, at
// two waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_mix valu_mix.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X X X X X X X X
template <int OP>
__global__ void k(unsigned *out, unsigned long long *cyc, unsigned seed, int iters) {
    unsigned a = seed + threadIdx.x, b = seed * 3 + 1;
    double d0 = 1.5 + threadIdx.x, d1 = 2.5, d2 = 3.5, d3 = 4.5;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        if (OP == 0) { REP8(asm volatile("v_cvt_f64_i32 %0, %4\n v_cvt_f64_i32 %1, %4\n v_cvt_f64_i32 %2, %4\n v_cvt_f64_i32 %3, %4" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(a));) }
        if (OP == 1) { REP8(asm volatile("v_bfe_i32 %0, %1, 3, 7\n v_bfe_i32 %0, %1, 10, 7\n v_bfe_i32 %0, %1, 17, 7\n v_bfe_i32 %0, %1, 24, 7" : "=v"(b) : "v"(a));) }
        if (OP == 2) { REP8(asm volatile("v_trunc_f64 %0, %4\n v_trunc_f64 %1, %4\n v_trunc_f64 %2, %4\n v_trunc_f64 %3, %4" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(d3));) }
        if (OP == 3) { REP8(asm volatile("v_add_u32 %0, %1, %1\n v_add_u32 %0, %1, %1\n v_add_u32 %0, %1, %1\n v_add_u32 %0, %1, %1" : "=v"(b) : "v"(a));) }
        if (OP == 4) { REP8(asm volatile("v_fma_f64 %0, %4, %4, %4\n v_fma_f64 %1, %4, %4, %4\n v_fma_f64 %2, %4, %4, %4\n v_fma_f64 %3, %4, %4, %4" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(d3));) }
        if (OP == 5) { REP8(asm volatile("v_cvt_i32_f64 %0, %1\n v_cvt_i32_f64 %0, %1\n v_cvt_i32_f64 %0, %1\n v_cvt_i32_f64 %0, %1" : "=v"(b) : "v"(d3));) }
        if (OP == 6) { REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %0, %1\n v_mov_b32 %0, %1\n v_mov_b32 %0, %1" : "=v"(b) : "v"(a));) }
        if (OP == 7) { REP8(asm volatile("v_add_f64 %0, %4, %4\n v_add_f64 %1, %4, %4\n v_add_f64 %2, %4, %4\n v_add_f64 %3, %4, %4" : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3) : "v"(d3));) }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = b + (unsigned) (d0 + d1 + d2 + d3);
    if (threadIdx.x % 64 == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
}
template <int OP>
void run(const char *name, int waves_per_block) {
    const int iters = 4000, blocks = 256;
    unsigned *out;
    unsigned long long *cyc;
    hipMalloc(&out, 4 * blocks * 64 * waves_per_block);
    hipMalloc(&cyc, 8 * blocks * waves_per_block);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(64 * waves_per_block), 0, 0, out, cyc, 7u, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * waves_per_block);
    hipMemcpy(h.data(), cyc, 8 * h.size(), hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h) sum += (double) v;
    const double per_wave = sum / h.size() / ((double) iters * 32);
    printf("%-16s waves/SIMD %d: %.2f memtime ticks per instruction per wave -> %.2f of SIMD time\n", name, waves_per_block / 4, per_wave,
           per_wave / (waves_per_block / 4));
    hipFree(out);
    hipFree(cyc);
}
int main() {
    for (int w : {4, 8}) {
        run<4>("v_fma_f64", w);
        run<7>("v_add_f64", w);
        run<0>("v_cvt_f64_i32", w);
        run<5>("v_cvt_i32_f64", w);
        run<2>("v_trunc_f64", w);
        run<1>("v_bfe_i32", w);
        run<3>("v_add_u32", w);
        run<6>("v_mov_b32", w);
    }
    return 0;
}

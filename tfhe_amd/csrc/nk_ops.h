// TGSW / TLWE operations with the ring dimension N, the mask size k and the gadget (l, Bgbit) as run-time parameters
// (SURVEY.md 8(f) N4): kernels and C-ABI entry points.  Included at the end of tfhe_hip.hip (same translation unit: it uses
// the context, the error helpers and the host<->device copy helpers defined there).
//
// Reference: its TGSW / TLWE layer is generic in these parameters (TGswParams / TLweParams, src/libtfhe/tgsw.cpp:7-27;
// tGswExternMulToTLwe tgsw-functions.cpp:157-170; tLweAddMulRTo tlwe-functions.cpp:191-197; Torus32PolynomialDecompH_old
// tgsw-functions.cpp:277-293; parameter sets of its own tests: src/test/tgsw_fft_test.cpp:28-35, N = 512 / 1024 / 2048,
// k = 1 / 2, (l, Bgbit) = (4, 8) / (3, 10)).  The fork's transform exists for N = 1024 only (tfhe_core.h:14), so these run the
// folded transforms of nfft_core.h in every arithmetic mode, and parity is defined against the reference's exact products
// (multiplication.cpp:31-43) as in its tests (src/test/test-multiplication.cpp:66-77, tgsw_fft_test.cpp:245-285).
#pragma once
#include "nfft_core.h"

namespace {

constexpr int kNWaves = 4;  // waves per workgroup of the kernels below
template <int LH>
constexpr size_t n_lds_bytes(int waves = kNWaves) {
    return (size_t) (nfft::Dim<LH>::kTableEntries + waves * nfft::Dim<LH>::NH) * sizeof(c64);
}
// the product kernel is bound by the latency of its operand loads: as many waves per CU as LDS and registers hold
// (N = 512, 1024: sixteen waves per workgroup at <= 128 registers; N = 2048: eight - tables 32 KB + 8 x 16 KB = all of LDS)
template <int LH>
constexpr int n_mul_waves() {
    return LH == 10 ? 8 : 16;
}
template <int LH>
__device__ __forceinline__ void n_load_tables(c64 *lds, const c64 *tables) {
    for (int i = threadIdx.x; i < nfft::Dim<LH>::kTableEntries; i += blockDim.x) lds[i] = tables[i];
    __syncthreads();
}
// phase-0 register image of a polynomial: register r = coefficient bitrev_R(r) << 6 | lane (+ NH in the imaginary part)
template <int LH, class F>
__device__ __forceinline__ void n_load_poly(c64 (&v)[nfft::Dim<LH>::VPL], int lane, F coef) {
    using D = nfft::Dim<LH>;
#pragma unroll
    for (int r = 0; r < D::VPL; r++) {
        const int j = (nfft::bitrev(r, D::R) << 6) | lane;
        v[r] = {coef(j), coef(j + D::NH)};
    }
}
// spectrum in last-phase layout -> the same values where the inverse transform wants them (register renaming)
template <int LH>
__device__ __forceinline__ void n_to_inverse_input(c64 (&y)[nfft::Dim<LH>::VPL], const c64 (&z)[nfft::Dim<LH>::VPL]) {
    using D = nfft::Dim<LH>;
#pragma unroll
    for (int r = 0; r < D::VPL; r++) y[r] = z[nfft::bitrev(r, D::R)];
}

// products of an IntPolynomial with a TorusPolynomial in Z[X]/(X^N+1) (torusPolynomialMultFFT, polynomials.cpp:241, at any N):
// out[q] (+)= ipolys[q / ishare] * tpolys[q]; one wave per product, workgroups walk over the batch
template <int LH, bool ADD>
__global__ __launch_bounds__(64 * n_mul_waves<LH>()) void k_n_mul(const int32_t *__restrict__ ipolys, const int32_t *__restrict__ tpolys,
                                                        int32_t *__restrict__ out, const c64 *__restrict__ tables, long long count,
                                                        int ishare) {
    using D = nfft::Dim<LH>;
    constexpr int W = n_mul_waves<LH>();
    extern __shared__ __align__(16) unsigned char lds_raw[];
    c64 *tab = reinterpret_cast<c64 *>(lds_raw);
    n_load_tables<LH>(tab, tables);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    c64 *scratch = tab + D::kTableEntries + wave * D::NH;
#pragma unroll 1
    for (long long q = (long long) blockIdx.x * W + wave; q < count; q += (long long) gridDim.x * W) {
        const int32_t *ip = ipolys + (q / ishare) * D::N, *tp = tpolys + q * D::N;
        c64 a[D::VPL], b[D::VPL];
        n_load_poly<LH>(a, lane, [&](int j) { return (double) ip[j]; });
        n_load_poly<LH>(b, lane, [&](int j) { return tp[j] * (1.0 / 4294967296.0); });
        nfft::transform<LH, false>(a, lane, scratch, tab);
        wave_sync();
        nfft::transform<LH, false>(b, lane, scratch, tab);
        wave_sync();
#pragma unroll
        for (int r = 0; r < D::VPL; r++) a[r] = brf::cmul(a[r], b[r]);
        n_to_inverse_input<LH>(b, a);
        nfft::transform<LH, true>(b, lane, scratch, tab);
        int32_t lo[D::VPL], hi[D::VPL];
        nfft::finish_inverse<LH>(lo, hi, b, lane, tab);
        int32_t *o = out + q * D::N;
#pragma unroll
        for (int r = 0; r < D::VPL; r++) {
            const int j = (r << 6) | lane;
            o[j] = ADD ? (int32_t) ((uint32_t) o[j] + (uint32_t) lo[r]) : lo[r];
            o[j + D::NH] = ADD ? (int32_t) ((uint32_t) o[j + D::NH] + (uint32_t) hi[r]) : hi[r];
        }
        wave_sync();
    }
}
// IntPolynomial_ifft / TorusPolynomial_ifft (polynomials.cpp:347-376) at any N: spectra in the reference's coefsC order
template <int LH>
__global__ __launch_bounds__(64 * kNWaves) void k_n_fwd(const int32_t *__restrict__ polys, c64 *__restrict__ spectra,
                                                        const c64 *__restrict__ tables, long long count, int torus) {
    using D = nfft::Dim<LH>;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    c64 *tab = reinterpret_cast<c64 *>(lds_raw);
    n_load_tables<LH>(tab, tables);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    c64 *scratch = tab + D::kTableEntries + wave * D::NH;
    const long long q = (long long) blockIdx.x * kNWaves + wave;
    if (q >= count) return;
    const int32_t *p = polys + q * D::N;
    const double s = torus ? 1.0 / 4294967296.0 : 1.0;
    c64 v[D::VPL];
    n_load_poly<LH>(v, lane, [&](int j) { return p[j] * s; });
    nfft::transform<LH, false>(v, lane, scratch, tab);
#pragma unroll
    for (int r = 0; r < D::VPL; r++) {
        bool conj;
        const int slot = nfft::ref_slot<LH>((r << 6) | lane, conj);
        spectra[q * D::NH + slot] = conj ? c64{v[r].re, -v[r].im} : v[r];
    }
}
template <int LH>
__device__ __forceinline__ c64 n_from_ref(const c64 *C, int k) {
    bool conj;
    const c64 t = C[nfft::ref_slot<LH>(k, conj)];
    return conj ? c64{t.re, -t.im} : t;
}
// TorusPolynomial_fft (polynomials.cpp:378-393) at any N
template <int LH>
__global__ __launch_bounds__(64 * kNWaves) void k_n_inv(const c64 *__restrict__ spectra, int32_t *__restrict__ polys,
                                                        const c64 *__restrict__ tables, long long count) {
    using D = nfft::Dim<LH>;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    c64 *tab = reinterpret_cast<c64 *>(lds_raw);
    n_load_tables<LH>(tab, tables);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    c64 *scratch = tab + D::kTableEntries + wave * D::NH;
    const long long q = (long long) blockIdx.x * kNWaves + wave;
    if (q >= count) return;
    c64 y[D::VPL];
#pragma unroll
    for (int r = 0; r < D::VPL; r++) y[r] = n_from_ref<LH>(spectra + q * D::NH, (nfft::bitrev(r, D::R) << 6) | lane);
    nfft::transform<LH, true>(y, lane, scratch, tab);
    int32_t lo[D::VPL], hi[D::VPL];
    nfft::finish_inverse<LH>(lo, hi, y, lane, tab);
#pragma unroll
    for (int r = 0; r < D::VPL; r++) {
        polys[q * D::N + ((r << 6) | lane)] = lo[r];
        polys[q * D::N + ((r << 6) | lane) + D::NH] = hi[r];
    }
}
// Torus32PolynomialDecompH_old (tgsw-functions.cpp:277-293): out[g][p][j] = digit p of in[g][j]
struct NGadget {
    uint32_t offset, mask;  // tgsw.cpp:7-27
    int32_t half, l, bgbit;
};
__device__ __forceinline__ int32_t n_digit(const NGadget &g, int32_t coef, int p) {
    const uint32_t u = (uint32_t) coef + g.offset;
    return (int32_t) ((u >> (32 - (p + 1) * g.bgbit)) & g.mask) - g.half;
}
__global__ void k_n_decomp(int32_t *__restrict__ out, const int32_t *__restrict__ in, int n, NGadget g, long long total) {
    const long long idx = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long long poly = idx / n;
    const int j = (int) (idx % n);
    for (int p = 0; p < g.l; p++) out[(poly * g.l + p) * n + j] = n_digit(g, in[idx], p);
}
// External product acc_out[g] = gsw[index[g]] (*) acc_in[g]; one wave per (g, output polynomial c2).
//   FFTDOM  tGswFFTExternMulToTLwe (tgsw-fft-operations.cpp:92-120): gsw = spectra [ngsw][(k+1)l][k+1][N/2] in the reference's
//           coefsC order; the (k+1)l row products are summed in the evaluation domain (rows in order), one inverse transform
//   !FFTDOM tGswExternMulToTLwe (tgsw-functions.cpp:157-170): gsw = coefficients int32 [ntgsw][(k+1)l][k+1][N]; every row
//           product is a product of its own (tLweAddMulRTo -> torusPolynomialAddMulR), added as Torus32
// The forward transforms of the digits are repeated by the k+1 waves of an accumulator (each needs all rows).
template <int LH, bool FFTDOM>
__global__ __launch_bounds__(64 * kNWaves) void k_n_extmul(const int32_t *__restrict__ acc_in, int32_t *__restrict__ acc_out,
                                                           const void *__restrict__ gsw, const int32_t *__restrict__ index,
                                                           const c64 *__restrict__ tables, long long count, int k, NGadget gd) {
    using D = nfft::Dim<LH>;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    c64 *tab = reinterpret_cast<c64 *>(lds_raw);
    n_load_tables<LH>(tab, tables);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    c64 *scratch = tab + D::kTableEntries + wave * D::NH;
    const long long q = (long long) blockIdx.x * kNWaves + wave;
    if (q >= count * (k + 1)) return;
    const long long g = q / (k + 1);
    const int c2 = (int) (q % (k + 1)), rows = (k + 1) * gd.l;
    const int32_t *a = acc_in + g * (k + 1) * D::N;
    const long long sample = index[g];
    c64 sum[D::VPL];
    int32_t slo[D::VPL], shi[D::VPL];
#pragma unroll
    for (int r = 0; r < D::VPL; r++) sum[r] = {0.0, 0.0}, slo[r] = 0, shi[r] = 0;
#pragma unroll 1
    for (int row = 0; row < rows; row++) {
        const int c = row / gd.l, p = row % gd.l;
        c64 v[D::VPL];
        n_load_poly<LH>(v, lane, [&](int j) { return (double) n_digit(gd, a[c * D::N + j], p); });
        nfft::transform<LH, false>(v, lane, scratch, tab);
        wave_sync();
        if (FFTDOM) {
            const c64 *G = reinterpret_cast<const c64 *>(gsw) + ((sample * rows + row) * (k + 1) + c2) * D::NH;
#pragma unroll
            for (int r = 0; r < D::VPL; r++) brf::cmac(sum[r], v[r], n_from_ref<LH>(G, (r << 6) | lane));
        } else {
            const int32_t *T = reinterpret_cast<const int32_t *>(gsw) + ((sample * rows + row) * (k + 1) + c2) * D::N;
            c64 b[D::VPL];
            n_load_poly<LH>(b, lane, [&](int j) { return T[j] * (1.0 / 4294967296.0); });
            nfft::transform<LH, false>(b, lane, scratch, tab);
            wave_sync();
#pragma unroll
            for (int r = 0; r < D::VPL; r++) v[r] = brf::cmul(v[r], b[r]);
            n_to_inverse_input<LH>(b, v);
            nfft::transform<LH, true>(b, lane, scratch, tab);
            wave_sync();
            int32_t lo[D::VPL], hi[D::VPL];
            nfft::finish_inverse<LH>(lo, hi, b, lane, tab);
#pragma unroll
            for (int r = 0; r < D::VPL; r++) {
                slo[r] = (int32_t) ((uint32_t) slo[r] + (uint32_t) lo[r]);
                shi[r] = (int32_t) ((uint32_t) shi[r] + (uint32_t) hi[r]);
            }
        }
    }
    if (FFTDOM) {
        c64 y[D::VPL];
        n_to_inverse_input<LH>(y, sum);
        nfft::transform<LH, true>(y, lane, scratch, tab);
        nfft::finish_inverse<LH>(slo, shi, y, lane, tab);
    }
    int32_t *o = acc_out + (g * (k + 1) + c2) * D::N;
#pragma unroll
    for (int r = 0; r < D::VPL; r++) {
        o[(r << 6) | lane] = slo[r];
        o[((r << 6) | lane) + D::NH] = shi[r];
    }
}

// ---- host side ----------------------------------------------------------------------------------------------------------
int n_log(int32_t n) { return n == 512 ? 8 : n == 1024 ? 9 : n == 2048 ? 10 : 0; }
int n_check_gadget(int32_t k, int32_t l, int32_t bgbit, NGadget &g) {
    if (k < 1 || k > 2) return fail(TFHE_HIP_ERR_ARG, "k must be 1 or 2");
    if (l < 1 || bgbit < 1 || bgbit > 31 || l * bgbit > 32) return fail(TFHE_HIP_ERR_ARG, "gadget must satisfy 1 <= l, 1 <= Bgbit <= 31, l * Bgbit <= 32");
    uint32_t t = 0;
    for (int i = 0; i < l; i++) t += 1u << (32 - (i + 1) * bgbit);
    g.half = 1 << (bgbit - 1);
    g.offset = t * (uint32_t) g.half;
    g.mask = (1u << bgbit) - 1;
    g.l = l;
    g.bgbit = bgbit;
    return TFHE_HIP_OK;
}
template <int LH>
int n_tables(tfhe_hip_ctx *c, const c64 **out) {
    c64 *&slot = c->ntab[LH - 8];
    if (!slot) {
        std::vector<c64> h(nfft::Dim<LH>::kTableEntries);
        nfft::build_tables<LH>(h.data());
        HIPCHK(hipMalloc((void **) &slot, h.size() * sizeof(c64)));
        HIPCHK(hipMemcpyAsync(slot, h.data(), h.size() * sizeof(c64), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    *out = slot;
    return TFHE_HIP_OK;
}
#define N_DISPATCH(n, ...)                                                                           \
    switch (n_log(n)) {                                                                              \
        case 8: { constexpr int LH = 8; __VA_ARGS__; } break;                                        \
        case 9: { constexpr int LH = 9; __VA_ARGS__; } break;                                        \
        case 10: { constexpr int LH = 10; __VA_ARGS__; } break;                                      \
        default: return fail(TFHE_HIP_ERR_ARG, "N must be 512, 1024 or 2048");                       \
    }

template <int LH, bool ADD>
int n_launch_mul(tfhe_hip_ctx *c, int32_t *d_out, const int32_t *d_i, const int32_t *d_t, long long count, int ishare) {
    const c64 *tab;
    if (int r = n_tables<LH>(c, &tab)) return r;
    constexpr int W = n_mul_waves<LH>();
    constexpr size_t lds = n_lds_bytes<LH>(W);
    static_assert(lds <= 160 * 1024, "LDS of one CU");
    if (int r = set_max_lds_once((const void *) k_n_mul<LH, ADD>, (int) lds, c->device)) return r;
    const long long per_cu = (160 * 1024) / (long long) lds;  // resident workgroups per CU
    const long long blocks = std::min<long long>((count + W - 1) / W, 256 * per_cu);
    hipLaunchKernelGGL((k_n_mul<LH, ADD>), dim3((unsigned) blocks), dim3(64 * W), lds, c->stream, d_i, d_t, d_out, tab, count, ishare);
    HIPCHK(hipGetLastError());
    return TFHE_HIP_OK;
}
template <int LH, bool FFTDOM>
int n_launch_extmul(tfhe_hip_ctx *c, const int32_t *d_in, int32_t *d_out, const void *d_gsw, const int32_t *d_index, long long count,
                    int k, const NGadget &g) {
    const c64 *tab;
    if (int r = n_tables<LH>(c, &tab)) return r;
    if (int r = set_max_lds_once((const void *) k_n_extmul<LH, FFTDOM>, (int) n_lds_bytes<LH>(), c->device)) return r;
    const long long waves = count * (k + 1);
    hipLaunchKernelGGL((k_n_extmul<LH, FFTDOM>), dim3((unsigned) ((waves + kNWaves - 1) / kNWaves)), dim3(64 * kNWaves),
                       n_lds_bytes<LH>(), c->stream, d_in, d_out, d_gsw, d_index, tab, count, k, g);
    HIPCHK(hipGetLastError());
    return TFHE_HIP_OK;
}

}  // namespace

extern "C" {

int tfhe_hip_negacyclic_mul_n_batch_dev(tfhe_hip_ctx *c, int32_t N, int32_t *d_out, const int32_t *d_ipolys, const int32_t *d_tpolys,
                                        int64_t count) {
    if (int r = check_ctx(c, false)) return r;
    if (count < 0 || (count > 0 && (!d_out || !d_ipolys || !d_tpolys))) return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    if (!n_log(N)) return fail(TFHE_HIP_ERR_ARG, "N must be 512, 1024 or 2048");
    if (count == 0) return TFHE_HIP_OK;
    c->ev_br = c->ev_ks = false;
    HIPCHK(hipEventRecord(c->ev[0], c->stream));
    N_DISPATCH(N, if (int r = (n_launch_mul<LH, false>(c, d_out, d_ipolys, d_tpolys, count, 1))) return r)
    HIPCHK(hipEventRecord(c->ev[1], c->stream));
    c->ev_br = true;
    return TFHE_HIP_OK;
}
int tfhe_hip_negacyclic_mul_n_batch(tfhe_hip_ctx *c, int32_t N, int32_t *out, const int32_t *ipolys, const int32_t *tpolys,
                                    int64_t count) {
    if (int r = check_ctx(c, false)) return r;
    if (count < 0 || (count > 0 && (!out || !ipolys || !tpolys))) return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    if (!n_log(N)) return fail(TFHE_HIP_ERR_ARG, "N must be 512, 1024 or 2048");
    if (count == 0) return TFHE_HIP_OK;
    const size_t pb = (size_t) count * N * sizeof(int32_t);
    if (int r = h2d(c, c->in_a, ipolys, pb)) return r;
    if (int r = h2d(c, c->in_b, tpolys, pb)) return r;
    HIPCHK(c->outb.reserve(pb));
    if (int r = tfhe_hip_negacyclic_mul_n_batch_dev(c, N, (int32_t *) c->outb.p, (const int32_t *) c->in_a.p, (const int32_t *) c->in_b.p,
                                                    count))
        return r;
    return d2h(c, out, c->outb.p, pb);
}
static int n_fwd_batch(tfhe_hip_ctx *c, int32_t N, double *spectra, const int32_t *polys, int64_t count, int torus) {
    if (int r = check_ctx(c, false)) return r;
    if (count < 0 || (count > 0 && (!spectra || !polys))) return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    if (!n_log(N)) return fail(TFHE_HIP_ERR_ARG, "N must be 512, 1024 or 2048");
    if (count == 0) return TFHE_HIP_OK;
    const size_t pb = (size_t) count * N * sizeof(int32_t), sb = (size_t) count * (N / 2) * sizeof(c64);
    if (int r = h2d(c, c->in_a, polys, pb)) return r;
    HIPCHK(c->outb.reserve(sb));
    const unsigned blocks = (unsigned) ((count + kNWaves - 1) / kNWaves);
    N_DISPATCH(N, {
        const c64 *tab;
        if (int r = n_tables<LH>(c, &tab)) return r;
        if (int r = set_max_lds_once((const void *) k_n_fwd<LH>, (int) n_lds_bytes<LH>(), c->device)) return r;
        hipLaunchKernelGGL((k_n_fwd<LH>), dim3(blocks), dim3(64 * kNWaves), n_lds_bytes<LH>(), c->stream, (const int32_t *) c->in_a.p,
                           (c64 *) c->outb.p, tab, (long long) count, torus);
    })
    HIPCHK(hipGetLastError());
    return d2h(c, spectra, c->outb.p, sb);
}
int tfhe_hip_int_ifft_n_batch(tfhe_hip_ctx *c, int32_t N, double *spectra, const int32_t *polys, int64_t count) {
    return n_fwd_batch(c, N, spectra, polys, count, 0);
}
int tfhe_hip_torus_ifft_n_batch(tfhe_hip_ctx *c, int32_t N, double *spectra, const int32_t *polys, int64_t count) {
    return n_fwd_batch(c, N, spectra, polys, count, 1);
}
int tfhe_hip_torus_fft_n_batch(tfhe_hip_ctx *c, int32_t N, int32_t *polys, const double *spectra, int64_t count) {
    if (int r = check_ctx(c, false)) return r;
    if (count < 0 || (count > 0 && (!spectra || !polys))) return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    if (!n_log(N)) return fail(TFHE_HIP_ERR_ARG, "N must be 512, 1024 or 2048");
    if (count == 0) return TFHE_HIP_OK;
    const size_t pb = (size_t) count * N * sizeof(int32_t), sb = (size_t) count * (N / 2) * sizeof(c64);
    if (int r = h2d(c, c->in_a, spectra, sb)) return r;
    HIPCHK(c->outb.reserve(pb));
    const unsigned blocks = (unsigned) ((count + kNWaves - 1) / kNWaves);
    N_DISPATCH(N, {
        const c64 *tab;
        if (int r = n_tables<LH>(c, &tab)) return r;
        if (int r = set_max_lds_once((const void *) k_n_inv<LH>, (int) n_lds_bytes<LH>(), c->device)) return r;
        hipLaunchKernelGGL((k_n_inv<LH>), dim3(blocks), dim3(64 * kNWaves), n_lds_bytes<LH>(), c->stream, (const c64 *) c->in_a.p,
                           (int32_t *) c->outb.p, tab, (long long) count);
    })
    HIPCHK(hipGetLastError());
    return d2h(c, polys, c->outb.p, pb);
}
int tfhe_hip_tgsw_decomp_n_batch(tfhe_hip_ctx *c, int32_t N, int32_t l, int32_t Bgbit, int32_t *out, const int32_t *polys,
                                 int64_t count) {
    if (int r = check_ctx(c, false)) return r;
    if (count < 0 || N < 1 || (count > 0 && (!out || !polys))) return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    NGadget g;
    if (int r = n_check_gadget(1, l, Bgbit, g)) return r;
    if (count == 0) return TFHE_HIP_OK;
    const size_t pb = (size_t) count * N * sizeof(int32_t);
    if (int r = h2d(c, c->in_a, polys, pb)) return r;
    HIPCHK(c->outb.reserve(pb * l));
    const long long total = (long long) count * N;
    hipLaunchKernelGGL(k_n_decomp, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, c->stream, (int32_t *) c->outb.p,
                       (const int32_t *) c->in_a.p, (int) N, g, total);
    HIPCHK(hipGetLastError());
    return d2h(c, out, c->outb.p, pb * l);
}
static int n_extmul(tfhe_hip_ctx *c, int32_t N, int32_t k, int32_t l, int32_t Bgbit, int32_t *acc, const void *gsw, size_t gsw_elem_bytes,
                    const int32_t *index, int64_t ngsw, int64_t count, bool fftdom) {
    if (int r = check_ctx(c, false)) return r;
    if (count < 0 || ngsw < 0 || (count > 0 && (!acc || !gsw || !index || ngsw == 0))) return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    if (!n_log(N)) return fail(TFHE_HIP_ERR_ARG, "N must be 512, 1024 or 2048");
    NGadget g;
    if (int r = n_check_gadget(k, l, Bgbit, g)) return r;
    if (count == 0) return TFHE_HIP_OK;
    if (int r = check_index(index, count, ngsw)) return r;
    const size_t ab = (size_t) count * (k + 1) * N * sizeof(int32_t);
    const size_t gb = (size_t) ngsw * (k + 1) * l * (k + 1) * (fftdom ? N / 2 : N) * gsw_elem_bytes;
    if (int r = h2d(c, c->in_a, acc, ab)) return r;
    if (int r = h2d(c, c->in_b, gsw, gb)) return r;
    if (int r = h2d(c, c->in_c, index, (size_t) count * sizeof(int32_t))) return r;
    HIPCHK(c->outb.reserve(ab));
    if (fftdom) {
        N_DISPATCH(N, if (int r = (n_launch_extmul<LH, true>(c, (const int32_t *) c->in_a.p, (int32_t *) c->outb.p, c->in_b.p,
                                                             (const int32_t *) c->in_c.p, count, k, g))) return r)
    } else {
        N_DISPATCH(N, if (int r = (n_launch_extmul<LH, false>(c, (const int32_t *) c->in_a.p, (int32_t *) c->outb.p, c->in_b.p,
                                                              (const int32_t *) c->in_c.p, count, k, g))) return r)
    }
    return d2h(c, acc, c->outb.p, ab);
}
int tfhe_hip_tgsw_fft_extern_mul_n_batch(tfhe_hip_ctx *c, int32_t N, int32_t k, int32_t l, int32_t Bgbit, int32_t *acc,
                                         const double *gsw_fft, const int32_t *gsw_index, int64_t ngsw, int64_t count) {
    return n_extmul(c, N, k, l, Bgbit, acc, gsw_fft, sizeof(c64), gsw_index, ngsw, count, true);
}
int tfhe_hip_tgsw_extern_mul_n_batch(tfhe_hip_ctx *c, int32_t N, int32_t k, int32_t l, int32_t Bgbit, int32_t *acc, const int32_t *tgsw,
                                     const int32_t *tgsw_index, int64_t ntgsw, int64_t count) {
    return n_extmul(c, N, k, l, Bgbit, acc, tgsw, sizeof(int32_t), tgsw_index, ntgsw, count, false);
}
int tfhe_hip_tlwe_addmul_r_n_batch(tfhe_hip_ctx *c, int32_t N, int32_t k, int32_t *result, const int32_t *poly, const int32_t *sample,
                                   int64_t count) {
    if (int r = check_ctx(c, false)) return r;
    if (count < 0 || (count > 0 && (!result || !poly || !sample))) return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    if (!n_log(N)) return fail(TFHE_HIP_ERR_ARG, "N must be 512, 1024 or 2048");
    if (k < 1 || k > 2) return fail(TFHE_HIP_ERR_ARG, "k must be 1 or 2");
    if (count == 0) return TFHE_HIP_OK;
    const size_t ab = (size_t) count * (k + 1) * N * sizeof(int32_t);
    if (int r = h2d(c, c->in_a, result, ab)) return r;
    if (int r = h2d(c, c->in_b, sample, ab)) return r;
    if (int r = h2d(c, c->in_c, poly, ab / (k + 1))) return r;
    N_DISPATCH(N, if (int r = (n_launch_mul<LH, true>(c, (int32_t *) c->in_a.p, (const int32_t *) c->in_c.p, (const int32_t *) c->in_b.p,
                                                      count * (k + 1), k + 1))) return r)
    return d2h(c, result, c->in_a.p, ab);
}

}  // extern "C"

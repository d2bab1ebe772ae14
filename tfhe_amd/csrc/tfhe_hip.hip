// libtfhe_hip.so - HIP kernels (gfx950 / MI355X) and the C-ABI of include/tfhe_hip.h.
//
// Kernel inventory (reference functions each one replaces are cited at the kernel):
//   k_blind_rotate<MODE>   one wavefront = one ciphertext through all n = 630 CMux iterations; the accumulator
//                          lives in LDS, spectra live in registers, the FFT-domain bootstrapping key streams from
//                          L2/HBM with 16-byte coalesced loads.  Fuses gate pre-combination, modulus switch,
//                          test-vector rotation, blind rotation and sample extraction.
//   k_keyswitch            batched gather-subtract over the keyswitch key, 32 gates per thread so each key row
//                          fetched from L2 serves 32 ciphertexts.
//   k_bk_convert / k_bk_permute / k_ks_pack     one-time key layout conversion on upload.
//   k_poly_fwd / k_poly_inv / k_poly_mul        stage-level transforms (parity tests, FFT microbench).
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (the arithmetic contract of br_core.h forbids FMA
// contraction).
#include <hip/hip_runtime.h>
#include <type_traits>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <string>
#include <memory>
#include <vector>
#include "br_core.h"
#include "fast_core.h"
#include "gauss_cdt.h"
#include "internal.h"
#include "../../include/tfhe_hip.h"

#pragma clang fp contract(off)

using namespace brc;

namespace {

constexpr int kExt = kN + 1;      // extracted LWE_N sample length
constexpr int kCt = kLwe + 1;     // LWE_n sample length
constexpr int kKsRow = (kCt + 3) / 4 * 4;  // keyswitch row padded to a multiple of 4 dwords (632)
constexpr int kKsT = 8, kKsBasebit = 2;

// ---------------------------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------------------------
// Knock-out builds (-DTFHE_HIP_KO=mask; diagnostic only, results are WRONG, never shipped: tools/ko_build.sh): what the
// kernel costs without its transposes (1), twiddle reads (2), key loads (4), rotated accumulator reads (8)
#ifndef TFHE_HIP_KO
#define TFHE_HIP_KO 0
#endif
#ifndef TFHE_HIP_READS_FIRST
#define TFHE_HIP_READS_FIRST 0  // A/B: forward steps of the FAST kernel request their transposed values before the filler MAC
#endif
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct TwLds {  // this lane's column of the per-lane twiddle image in LDS (br_core.h: tw_slot_index)
    const c64 *t;  // image + lane
    __device__ __forceinline__ c64 operator()(int slot) const {
#if TFHE_HIP_KO & 2
        return c64{__builtin_amdgcn_readfirstlane(slot) * 1e-3 + 0.7, 0.7};
#else
        return t[slot * 64];
#endif
    }
};

// twiddles fetched ahead of the level that uses them
template <int N>
struct TwPre {
    c64 w[N];
};
template <int N, int BASE>
struct TwView {
    const TwPre<N> &p;
    __device__ __forceinline__ c64 operator()(int slot) const { return p.w[slot - BASE]; }
};
template <int N>
__device__ __forceinline__ void tw_preload(TwPre<N> &p, const TwLds &twl, int base) {
#pragma unroll
    for (int i = 0; i < N; i++) p.w[i] = twl(base + i);
}

// Exchanges between lanes of ONE wave go through LDS without a workgroup barrier: LDS operations of a wave are
// executed in issue order, so only the compiler has to be kept from reordering them.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
    __builtin_amdgcn_wave_barrier();
}

// Twiddles of the lane-dependent phases are requested before the transpose that precedes them, so that their LDS
// latency overlaps it (these two routines serve the latency-bound kernels; the throughput kernel has its own schedule).
__device__ __forceinline__ void wave_fwd(c64 (&v)[8], int lane, c64 *scratch, const PhaseAConsts &twu, const TwLds &twl) {
    TwPre<4> twb, twc;
    tw_preload(twb, twl, TWS_FB);
    fwd_phaseA(v, twu);
#pragma unroll
    for (int r = 0; r < 8; r++) scratch[swz(fwd_xA(lane, r))] = v[r];
    wave_sync();
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = scratch[swz(fwd_xB(lane, r))];
    tw_preload(twc, twl, TWS_FC);
    wave_sync();
    fwd_phaseB(v, TwView<4, TWS_FB>{twb});
#pragma unroll
    for (int r = 0; r < 8; r++) scratch[swz(fwd_xB(lane, r))] = v[r];
    wave_sync();
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = scratch[swz(fwd_xC(lane, r))];
    wave_sync();
    fwd_phaseC(v, TwView<4, TWS_FC>{twc});
}

// element index inside one half (fixed y3) of the 1024-entry inverse working array
__device__ __forceinline__ int inv_elem(int y) { return ((y >> 4) << 3) | (y & 7); }

template <bool BOUNDED>
__device__ __forceinline__ void wave_inv(int32_t (&out)[16], c64 (&v)[16], int lane, c64 *scratch,
                                         const PhaseAConsts &twu, const TwLds &twl) {
    TwPre<7> pb;   // levels 4..6
    TwPre<8> pc;   // levels 7..9
    TwPre<8> pf;   // last merge
    tw_preload(pb, twl, TWS_IB);
    inv_phaseA(v, twu);
    c64 w[16];
    // layout A (regs y3y2y1y0) -> layout B (regs y6y5y4,y3), one half (y3) at a time through the 8 KB scratch
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int q = 0; q < 8; q++) scratch[swz(inv_elem(inv_yA(lane, (h << 3) | q)))] = v[(h << 3) | q];
        wave_sync();
#pragma unroll
        for (int q = 0; q < 8; q++) w[(q << 1) | h] = scratch[swz(inv_elem(inv_yB(lane, (q << 1) | h)))];
        if (h == 1) tw_preload(pc, twl, TWS_IC);
        wave_sync();
    }
    inv_phaseB(w, TwView<7, TWS_IB>{pb});
    // layout B -> layout C (regs y9y8y7,y3)
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int q = 0; q < 8; q++) scratch[swz(inv_elem(inv_yB(lane, (q << 1) | h)))] = w[(q << 1) | h];
        wave_sync();
#pragma unroll
        for (int q = 0; q < 8; q++) v[(q << 1) | h] = scratch[swz(inv_elem(inv_yC(lane, (q << 1) | h)))];
        if (h == 1) tw_preload(pf, twl, TWS_FIN);
        wave_sync();
    }
    inv_phaseC(v, TwView<8, TWS_IC>{pc});
    inv_final<BOUNDED>(out, v, TwView<8, TWS_FIN>{pf});
}

// spectrum held as (lane, r) <-> physical slot x = r<<6|lane  ->  inverse-transform input registers, via scratch
__device__ __forceinline__ void spectrum_to_inv_input(c64 (&v)[16], const c64 (&s)[8], int lane, c64 *scratch) {
#pragma unroll
    for (int r = 0; r < 8; r++) scratch[swz(fwd_xC(lane, r))] = s[r];
    wave_sync();
#pragma unroll
    for (int rho = 0; rho < 16; rho++) {
        const int m = (br4(rho) << 6) | lane;
        if (br4(rho) < 8) {
            v[rho] = scratch[swz(gray(m))];
        } else {
            const c64 t = scratch[swz(gray(1023 - m))];
            v[rho] = {t.re, -t.im};
        }
    }
    wave_sync();
}

// complex product as the reference's std::complex<double> operator* evaluates it (polynomials.cpp:313-323)
__device__ __forceinline__ void cmac(c64 &acc, const c64 a, const c64 b) {
    const double ac = a.re * b.re, bd = a.im * b.im, ad = a.re * b.im, bc = a.im * b.re;
    acc.re += ac - bd;
    acc.im += ad + bc;
}

__device__ __forceinline__ void load_tw_to_lds(c64 *tw_lds, const c64 *tw, int entries = kTwLaneEntries) {
    for (int i = threadIdx.x; i < entries; i += blockDim.x) tw_lds[i] = tw[i];
    __syncthreads();
}

// ---------------------------------------------------------------------------------------------------------------
// blind rotation
// ---------------------------------------------------------------------------------------------------------------
// one bootstrap of a netlist level: tmp = (0,cst) + sa*pool[ia] + sb*pool[ib]  (ib < 0: no second operand)
struct BrJob {
    int32_t ia, ib, sa, sb, cst, pad0, pad1, pad2;
};
// one keyswitch of a netlist level: pool[out] = KS(ext[e1] (+ ext[e2]) + (0,bconst))  (e2 < 0: single operand)
struct KsJob {
    int32_t e1, e2, bconst, out;
};

struct BrArgs {
    const BrJob *jobs;     // MODE_GATE, optional: per-wave job list over a ciphertext pool at `a`
    const int32_t *a, *b;  // MODE_GATE: LWE_n inputs [B][n+1] (b may be null)
    int32_t sa, sb, cst;   //            tmp = (0,cst) + sa*a + sb*b          (boot-gates.cpp)
    int32_t mu;            //            test vector constant
    int32_t *out;          // MODE_GATE: extracted LWE_N samples [B][N+1]
    int32_t *acc_io;       // MODE_RAW / MODE_EXTMUL: TLWE accumulators [B][2][N], in/out
    const int32_t *bara;   // MODE_RAW: [B][n] rotation amounts; MODE_EXTMUL: [B] key index
    const c64 *bk;         // [n][6][2][8][64] physical slot order
    int bk_bytes;          // size of the key array behind bk (buffer-resource bound of the throughput kernels' key loads)
    const c64 *tw;         // per-lane twiddle image, kTwLaneEntries
    PhaseAConsts ka;       // in-register-phase twiddles (scalar registers)
    long long B;
#ifdef TFHE_HIP_CLOCK_DIAG  // diagnostic build only (tools/clock_diag.sh): per-wave (shader cycles, 100 MHz ticks)
    unsigned long long *diag;
#endif
};
enum { MODE_GATE = 0, MODE_RAW = 1, MODE_EXTMUL = 2 };

constexpr int kBrLdsPerWave = 2 * kN * 4 + kNh * 16;  // accumulator + scratch = 16 KB

// One CMux:  ACC <- ACC + bk_i (*) ((X^bara - 1) ACC)
//   reference tfhe_MuxRotate_FFT (lwe-bootstrapping-functions-fft.cpp:68-77)
//     = tLweMulByXaiMinusOne (tlwe-functions.cpp:202)  ->  tGswFFTExternMulToTLwe (tgsw-fft-operations.cpp:92-120)
//       -> tLweAddTo (tlwe-functions.cpp:149)
// EXTMUL variant: ACC <- bk_i (*) ACC (the external product alone).
//
// Schedule.  A lone wave spends ~40 % of a CMux waiting for its own LDS transposes (write -> read of the same scratch),
// and two waves per SIMD do not hide all of it.  So every write->read gap is filled with independent arithmetic of the
// same wave:  forward transform t carries the multiply-accumulate of transform t-1 in its two gaps (one key column
// each, followed by the request for the same column of row t, which then has a whole transform to arrive), and the two
// inverse transforms run staggered by one level group, each one's transposes covered by the other's butterflies.
// Order of the floating-point operations per accumulator is unchanged (rows p = 0..5 in order): results identical.
__device__ __forceinline__ void mac_col(c64 (&acc)[8], const c64 (&v)[8], const c64 (&kb)[8]) {
#pragma unroll
    for (int r = 0; r < 8; r++) cmac(acc[r], v[r], kb[r]);
}
// key column (8 values per lane) through a buffer resource: address = base (descriptor) + lane offset (one VGPR, set once) +
// uniform offset of the column and r KB (an SGPR, computed on the scalar unit), instead of a 64-bit address per lane and load
#ifndef TFHE_HIP_SKEW
#define TFHE_HIP_SKEW 0  // A/B: 1 = skewed forward pipeline of the FAST throughput kernel (fwd_body_skew; measured 1.6 % slower)
#endif
#ifndef TFHE_HIP_SKEW_PRIO
#define TFHE_HIP_SKEW_PRIO 0  // A/B: static wave priorities around the exchanges of the skewed pipeline
#endif
#if TFHE_HIP_SKEW_PRIO
#define SKEW_PRIO_EXCHANGE() prio_exchange()
#define SKEW_PRIO_COMPUTE() prio_compute()
#else
#define SKEW_PRIO_EXCHANGE() ((void) 0)
#define SKEW_PRIO_COMPUTE() ((void) 0)
#endif
#ifndef TFHE_HIP_COOP_PREP
#define TFHE_HIP_COOP_PREP 1  // A/B: FAST latency kernel prepares the next differences on all eight waves (a fourth barrier)
#endif
#ifndef TFHE_HIP_BARA_AHEAD
#define TFHE_HIP_BARA_AHEAD 1  // A/B: gate kernels request the input coefficients of iteration i + 1 during iteration i
#endif
#ifndef TFHE_HIP_BARRIER_EVERY
#define TFHE_HIP_BARRIER_EVERY 1  // A/B: the waves of a workgroup meet every 1 | 2 | 4 CMux iterations (power of two)
#endif
struct KeyBuf {
    __amdgpu_buffer_rsrc_t rsrc;
    int voff;  // lane * 16
};
__device__ __forceinline__ void load_col_buf(c64 (&kb)[8], const KeyBuf &k, int soff /*byte offset of the column, uniform*/) {
#if TFHE_HIP_KO & 4
#pragma unroll
    for (int r = 0; r < 8; r++) kb[r] = c64{0.25 + r, 0.5 - (double) (soff & 1)};
#else
    const int s0 = __builtin_amdgcn_readfirstlane(soff);
#pragma unroll
    for (int r = 0; r < 8; r++) {  // the r KB go into the scalar offset too: one lane-offset register serves all loads
        const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(k.rsrc, k.voff, s0 + r * 1024, 0);
        kb[r] = __builtin_bit_cast(c64, w);
    }
#endif
}
// halves (h = fixed y3) of the inverse transposes through the 8 KB scratch
template <int H>
__device__ __forceinline__ void inv_t1_write(c64 *scratch, const c64 (&v)[16], int lane) {
#pragma unroll
    for (int q = 0; q < 8; q++) scratch[swz(inv_elem(inv_yA(lane, (H << 3) | q)))] = v[(H << 3) | q];
}
template <int H>
__device__ __forceinline__ void inv_t1_read(c64 (&w)[16], const c64 *scratch, int lane) {
#pragma unroll
    for (int q = 0; q < 8; q++) w[(q << 1) | H] = scratch[swz(inv_elem(inv_yB(lane, (q << 1) | H)))];
}
template <int H>
__device__ __forceinline__ void inv_t2_write(c64 *scratch, const c64 (&w)[16], int lane) {
#pragma unroll
    for (int q = 0; q < 8; q++) scratch[swz(inv_elem(inv_yB(lane, (q << 1) | H)))] = w[(q << 1) | H];
}
template <int H>
__device__ __forceinline__ void inv_t2_read(c64 (&v)[16], const c64 *scratch, int lane) {
#pragma unroll
    for (int q = 0; q < 8; q++) v[(q << 1) | H] = scratch[swz(inv_elem(inv_yC(lane, (q << 1) | H)))];
}
__device__ __forceinline__ void spec_write(c64 *scratch, const c64 (&s)[8], int lane) {
#pragma unroll
    for (int r = 0; r < 8; r++) scratch[swz(fwd_xC(lane, r))] = s[r];
}
__device__ __forceinline__ void spec_read(c64 (&v)[16], const c64 *scratch, int lane) {
#pragma unroll
    for (int rho = 0; rho < 16; rho++) {
        const int m = (br4(rho) << 6) | lane;
        if (br4(rho) < 8) {
            v[rho] = scratch[swz(gray(m))];
        } else {
            const c64 t = scratch[swz(gray(1023 - m))];
            v[rho] = {t.re, -t.im};
        }
    }
}

// Static wave priority: the two waves of a SIMD run the same program unsynchronised.  A wave inside a butterfly phase
// asks for priority over one that is in a transpose gap (LDS traffic plus the filler MAC), so that the arithmetic of
// one wave is not held up by the other's exchange: measured -5 % wall on the 16,384-gate launch (A/B, same box); the
// opposite assignment gains 1 %, raising the filler MAC as well loses 3 %, raising before the transpose reads rather
// than after them gains another 0.6 % in the forward steps.
__device__ __forceinline__ void prio_compute() { __builtin_amdgcn_s_setprio(2); }
__device__ __forceinline__ void prio_exchange() { __builtin_amdgcn_s_setprio(0); }

// forward transform t (row c*l+p = t) of one CMux with the MAC of transform t-1 in its transpose gaps
template <bool ROTATE>
__device__ __forceinline__ void fwd_step(const int t, c64 (&v)[8], const c64 (&pv)[8], c64 (&tmpa)[2][8],
                                         c64 (&kb)[8], uint32_t (&tlo)[8], uint32_t (&thi)[8], const int32_t *acc,
                                         c64 *scratch, int lane, int bara, const KeyBuf &kbuf, int koff, const PhaseAConsts &twu,
                                         const TwLds &twl) {
    const int c = t >= kL ? 1 : 0, p = t - c * kL;
    if (p == 0) {  // rotated differences of column c, offset added and digit sign bits flipped (decomp_prepare)
        const int32_t *accc = acc + c * kN;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int j = (br3(r) << 6) | lane;
            if (ROTATE) {
                tlo[r] = decomp_prepare(rot_minus_one(accc, bara, j));
                thi[r] = decomp_prepare(rot_minus_one(accc, bara, j + kNh));
            } else {
                tlo[r] = decomp_prepare(accc[j]);
                thi[r] = decomp_prepare(accc[j + kNh]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = {(double) decomp_digit_rt(tlo[r], p), (double) decomp_digit_rt(thi[r], p)};
    // one 8-value key buffer in rotation: holds (row t-1, column 0) on entry, (row t-1, column 1) between the two
    // gaps, (row t, column 0) on exit
    const int row = koff + t * 2 * kNh * (int) sizeof(c64);  // byte offset of (row t, column 0), uniform
    fwd_phaseA(v, twu);
    prio_exchange();
#pragma unroll
    for (int r = 0; r < 8; r++) scratch[swz(fwd_xA(lane, r))] = v[r];
    wave_sync();
    if (t > 0) {
        mac_col(tmpa[0], pv, kb);
        load_col_buf(kb, kbuf, row - kNh * (int) sizeof(c64));
    }
    prio_compute();
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = scratch[swz(fwd_xB(lane, r))];
    wave_sync();
    fwd_phaseB(v, twl);
    prio_exchange();
#pragma unroll
    for (int r = 0; r < 8; r++) scratch[swz(fwd_xB(lane, r))] = v[r];
    wave_sync();
    if (t > 0) mac_col(tmpa[1], pv, kb);
    load_col_buf(kb, kbuf, row);
    prio_compute();
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = scratch[swz(fwd_xC(lane, r))];
    wave_sync();
    fwd_phaseC(v, twl);
}

template <bool ROTATE>
__device__ __forceinline__ void cmux(int32_t *acc, c64 *scratch, int lane, int bara, const KeyBuf &kbuf, int koff,
                                     const PhaseAConsts &twu, const TwLds &twl) {
    c64 tmpa[2][8], kb[8], va[8], vb[8];
    uint32_t tlo[8], thi[8];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int r = 0; r < 8; r++) tmpa[c][r] = {0.0, 0.0};
#pragma unroll 1
    for (int t = 0; t < 2 * kL; t += 2) {
        fwd_step<ROTATE>(t, va, vb, tmpa, kb, tlo, thi, acc, scratch, lane, bara, kbuf, koff, twu, twl);
        fwd_step<ROTATE>(t + 1, vb, va, tmpa, kb, tlo, thi, acc, scratch, lane, bara, kbuf, koff, twu, twl);
    }
    // tail of the forward pipeline and the two inverse transforms A (column 0) and B (column 1), staggered
    c64 xa[16], xb[16], wa[16], wb[16];
    mac_col(tmpa[0], vb, kb);
    load_col_buf(kb, kbuf, koff + (2 * kKpl - 1) * kNh * (int) sizeof(c64));
    prio_exchange();
    spec_write(scratch, tmpa[0], lane);
    wave_sync();
    spec_read(xa, scratch, lane);
    mac_col(tmpa[1], vb, kb);
    wave_sync();
    prio_exchange();
    spec_write(scratch, tmpa[1], lane);
    wave_sync();
    inv_phaseA(xa, twu);
    prio_compute();
    spec_read(xb, scratch, lane);
    wave_sync();
    prio_exchange();
    inv_t1_write<0>(scratch, xa, lane);
    wave_sync();
    inv_phaseA_lo(xb);
    prio_compute();
    inv_t1_read<0>(wa, scratch, lane);
    wave_sync();
    prio_exchange();
    inv_t1_write<1>(scratch, xa, lane);
    wave_sync();
    inv_phaseA_hi(xb, twu);
    prio_compute();
    inv_t1_read<1>(wa, scratch, lane);
    wave_sync();
    prio_exchange();
    inv_t1_write<0>(scratch, xb, lane);
    wave_sync();
    inv_level4(wa, twl);
    prio_compute();
    inv_t1_read<0>(wb, scratch, lane);
    wave_sync();
    prio_exchange();
    inv_t1_write<1>(scratch, xb, lane);
    wave_sync();
    inv_level5(wa, twl);
    prio_compute();
    inv_t1_read<1>(wb, scratch, lane);
    wave_sync();
    inv_level6(wa, twl);
    prio_exchange();
    inv_t2_write<0>(scratch, wa, lane);
    wave_sync();
    inv_level4(wb, twl);
    prio_compute();
    inv_t2_read<0>(xa, scratch, lane);
    wave_sync();
    prio_exchange();
    inv_t2_write<1>(scratch, wa, lane);
    wave_sync();
    inv_level5(wb, twl);
    prio_compute();
    inv_t2_read<1>(xa, scratch, lane);
    wave_sync();
    inv_level6(wb, twl);
    prio_exchange();
    inv_t2_write<0>(scratch, wb, lane);
    wave_sync();
    inv_level7(xa, twl);
    prio_compute();
    inv_t2_read<0>(xb, scratch, lane);
    wave_sync();
    prio_exchange();
    inv_t2_write<1>(scratch, wb, lane);
    wave_sync();
    inv_level8(xa, twl);
    prio_compute();
    inv_t2_read<1>(xb, scratch, lane);
    wave_sync();
    inv_level9(xa, twl);
    {
        int32_t res[16];
        inv_final<true>(res, xa, twl);
#pragma unroll
        for (int rho = 0; rho < 16; rho++) {
            const int y = inv_yC(lane, rho);
            acc[y] = ROTATE ? (int32_t) ((uint32_t) acc[y] + (uint32_t) res[rho]) : res[rho];
        }
    }
    inv_phaseC(xb, twl);
    {
        int32_t res[16];
        inv_final<true>(res, xb, twl);
#pragma unroll
        for (int rho = 0; rho < 16; rho++) {
            const int y = inv_yC(lane, rho);
            acc[kN + y] = ROTATE ? (int32_t) ((uint32_t) acc[kN + y] + (uint32_t) res[rho]) : res[rho];
        }
    }
    wave_sync();
}

// ---------------------------------------------------------------------------------------------------------------
// FAST arithmetic mode (fast_core.h): the same CMux with folded 512-point transforms in both directions and fused
// multiply-adds.  Same pipeline as above: the MAC of transform t-1 in the transpose gaps of transform t; the two inverse
// transforms, now of the forward transform's shape, alternate on the scratch so that each one's transposes are covered
// by the other's butterflies.  The summed spectra enter the inverse transforms by register renaming (no exchange).
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void mac_col_f(c64 (&acc)[8], const c64 (&v)[8], const c64 (&kb)[8]) {
#pragma unroll
    for (int r = 0; r < 8; r++) brf::cmac(acc[r], v[r], kb[r]);
}
__device__ __forceinline__ void t_write_A(c64 *scratch, const c64 (&v)[8], int lane) {
#if TFHE_HIP_KO & 1
    return;
#endif
#pragma unroll
    for (int r = 0; r < 8; r++) scratch[swz(fwd_xA(lane, r))] = v[r];
}
__device__ __forceinline__ void t_write_B(c64 *scratch, const c64 (&v)[8], int lane) {
#if TFHE_HIP_KO & 1
    return;
#endif
#pragma unroll
    for (int r = 0; r < 8; r++) scratch[swz(fwd_xB(lane, r))] = v[r];
}
__device__ __forceinline__ void t_read_B(c64 (&v)[8], const c64 *scratch, int lane) {
#if TFHE_HIP_KO & 1
    return;
#endif
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = scratch[swz(fwd_xB(lane, r))];
}
__device__ __forceinline__ void t_read_C(c64 (&v)[8], const c64 *scratch, int lane) {
#if TFHE_HIP_KO & 1
    return;
#endif
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = scratch[swz(fwd_xC(lane, r))];
}
__device__ __forceinline__ void wave_fwd_fast(c64 (&v)[8], int lane, c64 *scratch, const PhaseAConsts &twu, const TwLds &twl) {
    brf::fwd_phaseA(v, twu);
    t_write_A(scratch, v, lane);
    wave_sync();
    t_read_B(v, scratch, lane);
    wave_sync();
    brf::fwd_phaseBC<brf::FS_FB>(v, twl);
    t_write_B(scratch, v, lane);
    wave_sync();
    t_read_C(v, scratch, lane);
    wave_sync();
    brf::fwd_phaseBC<brf::FS_FC>(v, twl);
}
// s: spectrum as the forward transform leaves it (register r = Y_(r<<6|lane)); lo/hi: coefficients j = r<<6|lane, j + 512
template <bool BOUNDED>
__device__ __forceinline__ void wave_inv_fast(int32_t (&lo)[8], int32_t (&hi)[8], const c64 (&s)[8], int lane, c64 *scratch,
                                              const PhaseAConsts &twu, const TwLds &twl) {
    c64 v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = s[br3(r)];
    brf::inv_phaseA(v, twu);
    t_write_A(scratch, v, lane);
    wave_sync();
    t_read_B(v, scratch, lane);
    wave_sync();
    brf::inv_phaseBC<brf::FS_IB>(v, twl);
    t_write_B(scratch, v, lane);
    wave_sync();
    t_read_C(v, scratch, lane);
    wave_sync();
    brf::inv_phaseBC<brf::FS_IC>(v, twl);
    brf::inv_final<BOUNDED>(lo, hi, v, twl);
}

template <bool ROTATE>
__device__ __forceinline__ void fwd_step_fast(const int t, c64 (&v)[8], const c64 (&pv)[8], c64 (&tmpa)[2][8],
                                              c64 (&kb)[8], uint32_t (&tlo)[8], uint32_t (&thi)[8], const int32_t *acc,
                                              c64 *scratch, int lane, int bara, const KeyBuf &kbuf, int koff, const PhaseAConsts &twu,
                                              const TwLds &twl) {
    const int c = t >= kL ? 1 : 0, p = t - c * kL;
    if (p == 0) {
        const int32_t *accc = acc + c * kN;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int j = (br3(r) << 6) | lane;
#if TFHE_HIP_KO & 8
            tlo[r] = decomp_prepare(j * bara + c);
            thi[r] = decomp_prepare(j - bara);
            continue;
#endif
            if (ROTATE) {
                tlo[r] = decomp_prepare(rot_minus_one(accc, bara, j));
                thi[r] = decomp_prepare(rot_minus_one(accc, bara, j + kNh));
            } else {
                tlo[r] = decomp_prepare(accc[j]);
                thi[r] = decomp_prepare(accc[j + kNh]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = {(double) decomp_digit_rt(tlo[r], p), (double) decomp_digit_rt(thi[r], p)};
    const int row = koff + t * 2 * kNh * (int) sizeof(c64);  // byte offset of (row t, column 0) of this key sample
    brf::fwd_phaseA(v, twu);
    prio_exchange();
    t_write_A(scratch, v, lane);
    wave_sync();
#if TFHE_HIP_READS_FIRST
    t_read_B(v, scratch, lane);   // requested before the filler: LDS executes a wave's operations in order
    wave_sync();
#endif
    if (t > 0) {
        mac_col_f(tmpa[0], pv, kb);
        load_col_buf(kb, kbuf, row - kNh * (int) sizeof(c64));
    }
    prio_compute();
#if !TFHE_HIP_READS_FIRST
    t_read_B(v, scratch, lane);
    wave_sync();
#endif
    brf::fwd_phaseBC<brf::FS_FB>(v, twl);
    prio_exchange();
    t_write_B(scratch, v, lane);
    wave_sync();
#if TFHE_HIP_READS_FIRST
    t_read_C(v, scratch, lane);
    wave_sync();
#endif
    if (t > 0) mac_col_f(tmpa[1], pv, kb);
    load_col_buf(kb, kbuf, row);
    prio_compute();
#if !TFHE_HIP_READS_FIRST
    t_read_C(v, scratch, lane);
    wave_sync();
#endif
    brf::fwd_phaseBC<brf::FS_FC>(v, twl);
}

// Skewed forward pipeline (TFHE_HIP_SKEW=1, round 3; A/B switch, off: measured 75.9-76.0 against 74.7 ms per 16,384 gates on the
// same box, 75.4 with static priorities - twice the cover per exchange does not help, i.e. the waves are not waiting for their
// own exchanges but for the shared LDS and VALU).  In fwd_step_fast every exchange of transform t is covered by ONE MAC
// of transform t-1 (64 FMA).  Here the last phase of transform t-1 is postponed behind the first phase of transform t, so
// that each exchange is covered by a butterfly phase AND a MAC of the neighbouring transform:
//   body(t) = [M1(t-2)] [digits_t, A_t] [read2(t-1)] [write1(t)] [C_(t-1), M0(t-1)] [read1(t), B_t] [write2(t)]
//   exchange 1 of t (write1 -> read1): C_(t-1) + M0(t-1);  exchange 2 of t (write2 -> read2 in body(t+1)): M1(t-1) + digits + A_(t+1).
// One scratch suffices: read2(t-1) is ISSUED before write1(t) and LDS executes a wave's operations in issue order.  Same
// registers (two value buffers, one rotating key buffer), same arithmetic in the same order per accumulator: results are
// bit-identical to fwd_step_fast's.  X = buffer of transform t, Y = buffer of transform t-1.
template <bool ROTATE>
__device__ __forceinline__ void fwd_digits_fast(const int t, c64 (&v)[8], uint32_t (&tlo)[8], uint32_t (&thi)[8], const int32_t *acc,
                                                int lane, int bara) {
    const int c = t >= kL ? 1 : 0, p = t - c * kL;
    if (p == 0) {
        const int32_t *accc = acc + c * kN;
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int j = (br3(r) << 6) | lane;
#if TFHE_HIP_KO & 8
            tlo[r] = decomp_prepare(j * bara + c);
            thi[r] = decomp_prepare(j - bara);
            continue;
#endif
            if (ROTATE) {
                tlo[r] = decomp_prepare(rot_minus_one(accc, bara, j));
                thi[r] = decomp_prepare(rot_minus_one(accc, bara, j + kNh));
            } else {
                tlo[r] = decomp_prepare(accc[j]);
                thi[r] = decomp_prepare(accc[j + kNh]);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 8; r++) v[r] = {(double) decomp_digit_rt(tlo[r], p), (double) decomp_digit_rt(thi[r], p)};
}
template <bool ROTATE>
__device__ __forceinline__ void fwd_body_skew(const int t, c64 (&X)[8], c64 (&Y)[8], c64 (&tmpa)[2][8], c64 (&kb)[8],
                                              uint32_t (&tlo)[8], uint32_t (&thi)[8], const int32_t *acc, c64 *scratch, int lane,
                                              int bara, const KeyBuf &kbuf, int koff, const PhaseAConsts &twu, const TwLds &twl) {
    constexpr int kColBytes = kNh * (int) sizeof(c64);
    const int row = koff + t * 2 * kColBytes;  // byte offset of (row t, column 0) of this key sample
    if (t >= 2) mac_col_f(tmpa[1], X, kb);                      // M1(t-2): X still holds the spectrum of transform t-2
    if (t >= 1) load_col_buf(kb, kbuf, row - 2 * kColBytes);    // key (t-1, column 0)
    fwd_digits_fast<ROTATE>(t, X, tlo, thi, acc, lane, bara);
    brf::fwd_phaseA(X, twu);
    TwPre<4> twc;
    if (t >= 1) {
        t_read_C(Y, scratch, lane);                             // read2(t-1), issued ahead of write1(t)
        tw_preload(twc, twl, brf::FS_FC);                       // ... and so are the twiddles of C_(t-1)
        wave_sync();
    }
    SKEW_PRIO_EXCHANGE();
    t_write_A(scratch, X, lane);
    wave_sync();
    SKEW_PRIO_COMPUTE();
    if (t >= 1) {
        brf::fwd_phaseBC<brf::FS_FC>(Y, TwView<4, brf::FS_FC>{twc});
        mac_col_f(tmpa[0], Y, kb);                              // M0(t-1)
        load_col_buf(kb, kbuf, row - kColBytes);                // key (t-1, column 1)
    }
    t_read_B(X, scratch, lane);
    wave_sync();
    brf::fwd_phaseBC<brf::FS_FB>(X, twl);
    SKEW_PRIO_EXCHANGE();
    t_write_B(scratch, X, lane);
    wave_sync();
    SKEW_PRIO_COMPUTE();
}

template <bool ROTATE>
__device__ __forceinline__ void cmux_fast(int32_t *acc, c64 *scratch, int lane, int bara, const KeyBuf &kbuf, int koff,
                                          const PhaseAConsts &twu, const TwLds &twl) {
    c64 tmpa[2][8], kb[8], va[8], vb[8];
    uint32_t tlo[8], thi[8];
#pragma unroll
    for (int c = 0; c < 2; c++)
#pragma unroll
        for (int r = 0; r < 8; r++) tmpa[c][r] = {0.0, 0.0};
#if TFHE_HIP_SKEW
#pragma unroll 1
    for (int t = 0; t < 2 * kL; t += 2) {
        fwd_body_skew<ROTATE>(t, va, vb, tmpa, kb, tlo, thi, acc, scratch, lane, bara, kbuf, koff, twu, twl);
        fwd_body_skew<ROTATE>(t + 1, vb, va, tmpa, kb, tlo, thi, acc, scratch, lane, bara, kbuf, koff, twu, twl);
    }
    // drain: M1(2l-2), then the last phase of transform 2l-1 (vb), whose MACs open the tail below
    mac_col_f(tmpa[1], va, kb);
    load_col_buf(kb, kbuf, koff + (2 * kKpl - 2) * kNh * (int) sizeof(c64));
    t_read_C(vb, scratch, lane);
    wave_sync();
    brf::fwd_phaseBC<brf::FS_FC>(vb, twl);
#else
#pragma unroll 1
    for (int t = 0; t < 2 * kL; t += 2) {
        fwd_step_fast<ROTATE>(t, va, vb, tmpa, kb, tlo, thi, acc, scratch, lane, bara, kbuf, koff, twu, twl);
        fwd_step_fast<ROTATE>(t + 1, vb, va, tmpa, kb, tlo, thi, acc, scratch, lane, bara, kbuf, koff, twu, twl);
    }
#endif
    // tail of the forward pipeline, then the inverse transforms A (column 0) and B (column 1)
    c64 xa[8], xb[8];
    mac_col_f(tmpa[0], vb, kb);
    load_col_buf(kb, kbuf, koff + (2 * kKpl - 1) * kNh * (int) sizeof(c64));
#pragma unroll
    for (int r = 0; r < 8; r++) xa[r] = tmpa[0][br3(r)];
    brf::inv_phaseA(xa, twu);
    prio_exchange();
    t_write_A(scratch, xa, lane);
    wave_sync();
    mac_col_f(tmpa[1], vb, kb);
#pragma unroll
    for (int r = 0; r < 8; r++) xb[r] = tmpa[1][br3(r)];
    brf::inv_phaseA(xb, twu);
    prio_compute();
    t_read_B(xa, scratch, lane);
    wave_sync();
    prio_exchange();
    t_write_A(scratch, xb, lane);
    wave_sync();
    TwPre<4> twib, twic;  // one read of the lane-dependent twiddles serves both transforms of the pair
    tw_preload(twib, twl, brf::FS_IB);
    brf::inv_phaseBC<brf::FS_IB>(xa, TwView<4, brf::FS_IB>{twib});
    prio_compute();
    t_read_B(xb, scratch, lane);
    wave_sync();
    prio_exchange();
    t_write_B(scratch, xa, lane);
    wave_sync();
    brf::inv_phaseBC<brf::FS_IB>(xb, TwView<4, brf::FS_IB>{twib});
    prio_compute();
    t_read_C(xa, scratch, lane);
    wave_sync();
    prio_exchange();
    t_write_B(scratch, xb, lane);
    wave_sync();
    tw_preload(twic, twl, brf::FS_IC);
    TwPre<8> twif;  // untwist factors, likewise read once for both
    tw_preload(twif, twl, brf::FS_IF);
    brf::inv_phaseBC<brf::FS_IC>(xa, TwView<4, brf::FS_IC>{twic});
    {
        int32_t lo[8], hi[8];
        brf::inv_final<true>(lo, hi, xa, TwView<8, brf::FS_IF>{twif});
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int j = (r << 6) | lane;
            acc[j] = ROTATE ? (int32_t) ((uint32_t) acc[j] + (uint32_t) lo[r]) : lo[r];
            acc[j + kNh] = ROTATE ? (int32_t) ((uint32_t) acc[j + kNh] + (uint32_t) hi[r]) : hi[r];
        }
    }
    prio_compute();
    t_read_C(xb, scratch, lane);
    wave_sync();
    brf::inv_phaseBC<brf::FS_IC>(xb, TwView<4, brf::FS_IC>{twic});
    {
        int32_t lo[8], hi[8];
        brf::inv_final<true>(lo, hi, xb, TwView<8, brf::FS_IF>{twif});
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int j = (r << 6) | lane;
            acc[kN + j] = ROTATE ? (int32_t) ((uint32_t) acc[kN + j] + (uint32_t) lo[r]) : lo[r];
            acc[kN + j + kNh] = ROTATE ? (int32_t) ((uint32_t) acc[kN + j + kNh] + (uint32_t) hi[r]) : hi[r];
        }
    }
    wave_sync();
}

template <int MODE, int WAVES, bool FAST = false>
__global__ __launch_bounds__(64 * WAVES) void k_blind_rotate(BrArgs A) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    c64 *tw_lds = reinterpret_cast<c64 *>(lds_raw);
    constexpr int kTwImage = FAST ? brf::kFastLaneEntries : kTwLaneEntries;  // A.tw is the image of the mode
    load_tw_to_lds(tw_lds, A.tw, kTwImage);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    unsigned char *mine = lds_raw + kTwImage * sizeof(c64) + (size_t) wave * kBrLdsPerWave;
    int32_t *acc = reinterpret_cast<int32_t *>(mine);
    c64 *scratch = reinterpret_cast<c64 *>(mine + 2 * kN * 4);
    const long long g = (long long) blockIdx.x * WAVES + wave;
    if (g >= A.B) return;  // wave-uniform; no workgroup barrier after this point
    const PhaseAConsts twu = A.ka;
    const TwLds twl{tw_lds + lane};
    constexpr bool kKeySync = WAVES > 1;  // one workgroup barrier per CMux iteration (both arithmetic modes)
    constexpr int kSampleBytes = kKpl * 2 * kNh * (int) sizeof(c64);  // one TGSW sample of the key
    const KeyBuf kbuf{__builtin_amdgcn_make_buffer_rsrc(const_cast<c64 *>(A.bk), 0, A.bk_bytes, 0x00020000), lane * 16};

    if (MODE == MODE_GATE) {
        // tfhe_bootstrap_woKS_FFT (lwe-bootstrapping-functions-fft.cpp:222-253) with the gate's linear
        // pre-combination folded in; tfhe_blindRotateAndExtract_FFT (:180-209)
        const int32_t *ga = A.a + g * kCt;
        const int32_t *gb = A.b ? A.b + g * kCt : nullptr;
        int32_t sa = A.sa, sb = A.sb, cst = A.cst;
        if (A.jobs) {  // netlist level: operands are rows of the pool, coefficients per job
            const BrJob *j = A.jobs + g;
            const int ia = __builtin_amdgcn_readfirstlane(j->ia), ib = __builtin_amdgcn_readfirstlane(j->ib);
            sa = __builtin_amdgcn_readfirstlane(j->sa);
            sb = __builtin_amdgcn_readfirstlane(j->sb);
            cst = __builtin_amdgcn_readfirstlane(j->cst);
            ga = A.a + (long long) ia * kCt;
            gb = ib >= 0 ? A.a + (long long) ib * kCt : nullptr;
        }
        auto comb = [&](int i) -> int32_t {
            uint32_t t = (uint32_t) sa * (uint32_t) ga[i];
            if (gb) t += (uint32_t) sb * (uint32_t) gb[i];
            return (int32_t) t;
        };
        const int barb = modswitch_2N((int32_t) ((uint32_t) comb(kLwe) + (uint32_t) cst));
        // ACC = (0, X^(2N-barb) * [mu,...,mu])   (torusPolynomialMulByXai, toruspolynomial-functions.cpp:128-147)
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int j = q * 64 + lane;
            acc[j] = 0;
            acc[kN + j] = ((j + barb) & kN) ? (int32_t) (0u - (uint32_t) A.mu) : A.mu;
        }
        wave_sync();
#ifdef TFHE_HIP_CLOCK_DIAG
        const unsigned long long diag_t0 = __builtin_amdgcn_s_memtime(), diag_r0 = __builtin_amdgcn_s_memrealtime();
#endif
        // The two input coefficients of iteration i + 1 are requested at the top of iteration i and used at the top of the
        // next one: otherwise two dependent global loads sit on the critical path of every iteration, with all waves of
        // the workgroup - just synchronised by the barrier - waiting for them at the same time (index n = the b
        // coefficient: always in range)
#if TFHE_HIP_BARA_AHEAD
        int32_t ra = ga[0], rb = gb ? gb[0] : 0;
#endif
#pragma unroll 1
        for (int i = 0; i < kLwe; i++) {
#if TFHE_HIP_BARA_AHEAD
            const uint32_t lin = (uint32_t) sa * (uint32_t) ra + (gb ? (uint32_t) sb * (uint32_t) rb : 0u);
            const int bara = __builtin_amdgcn_readfirstlane(modswitch_2N((int32_t) lin));
            ra = ga[i + 1];
            if (gb) rb = gb[i + 1];
#else
            const int bara = __builtin_amdgcn_readfirstlane(modswitch_2N(comb(i)));
#endif
            if (kKeySync) {
                // All waves of the workgroup enter iteration i together: the 96 KB of bk_i that every wave streams then
                // come out of the CU's vector cache for all but the first of them instead of once per wave out of L2
                // (1 TB per 16,384-gate launch, 48 of the ~70 GB/s a CU gets from its L2).  A rotation by 0 is not
                // skipped here, so that every wave meets every barrier: (X^0 - 1) ACC = 0 decomposes into zero digits
                // and adds exactly nothing in either arithmetic mode (the exact-mode parity tests cover such iterations), as in the
                // reference's device loop (src/kernels/tfhe_blindRotate_FFT.cpp:18-28).  Measured: 122.0 -> 118.4 ms (exact), 80.7 -> 76.9 ms (fast).
                // Waves that have ended (ragged last workgroup) are left out of the barrier by the hardware.
                if ((i & (TFHE_HIP_BARRIER_EVERY - 1)) == 0) __builtin_amdgcn_s_barrier();
            } else if (bara == 0) {
                continue;  // lwe-bootstrapping-functions-fft.cpp:154
            }
            if (FAST) cmux_fast<true>(acc, scratch, lane, bara, kbuf, i * kSampleBytes, twu, twl);
            else cmux<true>(acc, scratch, lane, bara, kbuf, i * kSampleBytes, twu, twl);
        }
#ifdef TFHE_HIP_CLOCK_DIAG
        if (A.diag && lane == 0) {
            A.diag[2 * g] = __builtin_amdgcn_s_memtime() - diag_t0;
            A.diag[2 * g + 1] = __builtin_amdgcn_s_memrealtime() - diag_r0;
        }
#endif
        // tLweExtractLweSampleIndex(index 0), lwe.cpp:41-53
        int32_t *o = A.out + g * kExt;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int j = q * 64 + lane;
            o[j] = j == 0 ? acc[0] : (int32_t) (0u - (uint32_t) acc[kN - j]);
        }
        if (lane == 0) o[kN] = acc[kN];
    } else {
        int32_t *io = A.acc_io + g * 2 * kN;
#pragma unroll
        for (int q = 0; q < 32; q++) acc[q * 64 + lane] = io[q * 64 + lane];
        wave_sync();
        if (MODE == MODE_RAW) {
            // tfhe_blindRotate_FFT (the reference's device seam), CPU loop at :146-164
            const int32_t *gbara = A.bara + g * kLwe;
            int32_t rn = gbara[0];  // requested one iteration ahead, as in the gate loop
#pragma unroll 1
            for (int i = 0; i < kLwe; i++) {
                const int bara = __builtin_amdgcn_readfirstlane(rn) & (2 * kN - 1);
                rn = gbara[i + 1 < kLwe ? i + 1 : i];
                if (kKeySync) __builtin_amdgcn_s_barrier();
                else if (bara == 0) continue;
                if (FAST) cmux_fast<true>(acc, scratch, lane, bara, kbuf, i * kSampleBytes, twu, twl);
                else cmux<true>(acc, scratch, lane, bara, kbuf, i * kSampleBytes, twu, twl);
            }
        } else {
            const int i = __builtin_amdgcn_readfirstlane(A.bara[g]);
            if (FAST) cmux_fast<false>(acc, scratch, lane, 0, kbuf, i * kSampleBytes, twu, twl);
            else cmux<false>(acc, scratch, lane, 0, kbuf, i * kSampleBytes, twu, twl);
        }
#pragma unroll
        for (int q = 0; q < 32; q++) io[q * 64 + lane] = acc[q * 64 + lane];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// blind rotation, latency shape: ONE WORKGROUP (6 waves) per ciphertext, for batches that cannot fill the chip.
// Per CMux: wave w transforms decomposition row w (column c = w/3, digit p = w%3) and multiplies it with key row w;
// the 2l products P_w are summed in row order by all threads - ((((P_0 + P_1) + P_2) + P_3) + P_4) + P_5 with P_w
// rounded on its own is exactly what the reference's clear-then-AddMul loop computes (tgsw-fft-operations.cpp:108-114),
// so results stay bit-identical.  Critical path per CMux: one forward + half an inverse transform instead of six + two.
// The two inverse transforms run on four waves, two per column (br_core.h "split over two waves").
// LDS: twiddle image (forward slots + split-inverse image, 32 KB) | accumulator 8 KB
//      | per wave: A (scratch, then product column 0) 8 KB, B (product column 1) 8 KB
//      | the 630 rotation amounts (computed once, so that no global load sits on the per-iteration critical path)
//      | prepared differences 8 KB (see prepare_diff)
// ---------------------------------------------------------------------------------------------------------------
template <int BASE>
struct TwReg4 {  // four twiddles of one phase held in registers
    c64 t[4];
    __device__ __forceinline__ c64 operator()(int slot) const { return t[slot - BASE]; }
};
constexpr int kCoopWaves = kKpl;
// Key rows: waves 4 and up (which idle during the inverse transforms) request the row of the next iteration right
// after the row sum, so it is there when their forward transform ends - they share a SIMD with an older wave and are
// the slower ones (5.9 k against 4.1 k cycles per forward step); waves 0..3 request theirs at the top of the iteration
// and have slack before the barrier.  (All waves requesting ahead: 3.39 -> 3.23 ms per gate with six waves, but the 64
// key registers carried through the inverse transform leave no room for anything else: 2.46 -> 2.59 ms with four.)
constexpr int kShapeCoop = 6;  // the name of this launch shape in launch_br / TFHE_HIP_BR_WAVES (whatever kKpl is)
constexpr int kTwHalfEntries = 2 * kTwHalfSlots * 64;          // image of the split inverse, both halves
constexpr int kCoopTwEntries = 8 * 64 + kTwHalfEntries;         // forward slots 0..7 of the main image + that: 32 KB
constexpr int kCoopLds = kCoopTwEntries * (int) sizeof(c64) + 2 * kN * 4 + kCoopWaves * 2 * kNh * (int) sizeof(c64) + 640 * 4 +
                         2 * kN * 4;
static_assert(kCoopLds <= 160 * 1024, "LDS of one CU");
static_assert(kCoopWaves >= 4, "the split inverse needs four waves");
static_assert(TWS_FB == 0 && TWS_FC == 4, "the forward slots are the first eight of the main image");

template <int MODE>
__global__ __launch_bounds__(64 * kCoopWaves) void k_blind_rotate_coop(BrArgs A) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    c64 *tw_lds = reinterpret_cast<c64 *>(lds_raw);
    for (int i = threadIdx.x; i < kCoopTwEntries; i += 64 * kCoopWaves)  // forward slots, then the split-inverse image
        tw_lds[i] = A.tw[i < 8 * 64 ? i : i - 8 * 64 + kTwLaneEntries];
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    int32_t *acc = reinterpret_cast<int32_t *>(lds_raw + kCoopTwEntries * sizeof(c64));
    c64 *bufs = reinterpret_cast<c64 *>(lds_raw + kCoopTwEntries * sizeof(c64) + 2 * kN * 4);
    c64 *mineA = bufs + (size_t) wave * 2 * kNh, *mineB = mineA + kNh;
    const long long g = blockIdx.x;  // host launches exactly B workgroups
    const PhaseAConsts twu = A.ka;
    const TwLds twl{tw_lds + lane};
    const int c = wave >= kL ? 1 : 0, p = wave - c * kL;
    // inverse transforms: waves 0..3, wave = 2 h + column (br_core.h "split over two waves")
    const int icol = wave & 1, ih = (wave >> 1) & 1;
    const TwLds twh{tw_lds + (8 + ih * kTwHalfSlots) * 64 + lane};

    const int32_t *ga = nullptr, *gb = nullptr;
    int32_t sa = A.sa, sb = A.sb, cst = A.cst;
    const int32_t *gbara = nullptr;
    if (MODE == MODE_GATE) {
        ga = A.a + g * kCt;
        gb = A.b ? A.b + g * kCt : nullptr;
        if (A.jobs) {
            const BrJob *j = A.jobs + g;
            const int ia = __builtin_amdgcn_readfirstlane(j->ia), ib = __builtin_amdgcn_readfirstlane(j->ib);
            sa = __builtin_amdgcn_readfirstlane(j->sa);
            sb = __builtin_amdgcn_readfirstlane(j->sb);
            cst = __builtin_amdgcn_readfirstlane(j->cst);
            ga = A.a + (long long) ia * kCt;
            gb = ib >= 0 ? A.a + (long long) ib * kCt : nullptr;
        }
    } else {
        gbara = A.bara + g * kLwe;
    }
    auto comb = [&](int i) -> int32_t {
        uint32_t t = (uint32_t) sa * (uint32_t) ga[i];
        if (gb) t += (uint32_t) sb * (uint32_t) gb[i];
        return (int32_t) t;
    };
    int32_t *bara_lds = reinterpret_cast<int32_t *>(bufs + (size_t) kCoopWaves * 2 * kNh);
    uint32_t *diff = reinterpret_cast<uint32_t *>(bara_lds + 640);  // [2][N]: decomp_prepare((X^a - 1) * ACC), next a
    for (int i = threadIdx.x; i < kLwe; i += 64 * kCoopWaves)
        bara_lds[i] = MODE == MODE_GATE ? modswitch_2N(comb(i)) : (gbara[i] & (2 * kN - 1));
    if (MODE == MODE_GATE) {
        const int barb = modswitch_2N((int32_t) ((uint32_t) comb(kLwe) + (uint32_t) cst));
        for (int j = threadIdx.x; j < kN; j += 64 * kCoopWaves) {
            acc[j] = 0;
            acc[kN + j] = ((j + barb) & kN) ? (int32_t) (0u - (uint32_t) A.mu) : A.mu;
        }
    } else {
        int32_t *io = A.acc_io + g * 2 * kN;
        for (int j = threadIdx.x; j < 2 * kN; j += 64 * kCoopWaves) acc[j] = io[j];
    }
    __syncthreads();

    // first iteration that is not skipped (bara = 0 is skipped: lwe-bootstrapping-functions-fft.cpp:154)
    auto next_active = [&](int i) {
        while (i < kLwe && __builtin_amdgcn_readfirstlane(bara_lds[i]) == 0) i++;
        return i;
    };
    // all threads: diff = decomp_prepare((X^a - 1) * ACC) for the coming iteration (torusPolynomialMulByXaiMinusOne,
    // toruspolynomial-functions.cpp:106-124), so that the forward waves only read, extract their digit and convert
    // (all reads first, then all writes: a write between two reads would stop the compiler from overlapping their
    //  LDS latencies, since it cannot know that `diff` and `acc` do not overlap; measured together with the same change
    //  to the row sum below: see DESIGN.md 4.1)
    constexpr int kCoopThreads = 64 * kCoopWaves;
    auto prepare_diff = [&](int a) {
        constexpr int kIter = (2 * kN + kCoopThreads - 1) / kCoopThreads, kGroup = kIter;
#pragma unroll
        for (int k0 = 0; k0 < kIter; k0 += kGroup) {
            uint32_t tmp[kGroup];
#pragma unroll
            for (int k = 0; k < kGroup; k++) {
                const int e = threadIdx.x + (k0 + k) * kCoopThreads;
                if (e < 2 * kN) tmp[k] = decomp_prepare(rot_minus_one(acc + (e & kN), a, e & (kN - 1)));
            }
#pragma unroll
            for (int k = 0; k < kGroup; k++) {
                const int e = threadIdx.x + (k0 + k) * kCoopThreads;
                if (e < 2 * kN) diff[e] = tmp[k];
            }
        }
    };
    // The summed spectrum of column `col` (512 values, 8 KB) lies, between the sum and the start of the inverse
    // transform, over that column's accumulator mirror and difference buffer (4 KB each): both are dead in that
    // window and are rewritten before their next use.  (Giving the first half a buffer of its own, which makes the
    // barrier after the spectrum reads unnecessary, measured slower: 3.35 against 3.18 ms per gate.)
    auto sum_slot = [&](int col, int idx) -> c64 * {
        c64 *lo = reinterpret_cast<c64 *>(acc + col * kN), *hi = reinterpret_cast<c64 *>(diff + col * kN);
        return idx < kNh / 2 ? lo + idx : hi + (idx - kNh / 2);
    };
    int i = next_active(0);
    int32_t own[8];  // waves 0..3: the coefficients y = half_y(lane, r, ih) of accumulator column icol
    if (wave < 4) {
#pragma unroll
        for (int r = 0; r < 8; r++) own[r] = acc[icol * kN + half_y(lane, r, ih)];
    }
    if (i < kLwe) prepare_diff(__builtin_amdgcn_readfirstlane(bara_lds[i]));
    __syncthreads();

    auto load_key_row = [&](c64 (&kb)[2][8], int it) {  // key row `wave` of TGSW sample `it`: 16 coalesced 1 KB loads
        const c64 *row = A.bk + ((size_t) it * kKpl + wave) * 2 * kNh + lane;
#pragma unroll
        for (int c2 = 0; c2 < 2; c2++)
#pragma unroll
            for (int r = 0; r < 8; r++) kb[c2][r] = row[c2 * kNh + r * 64];
    };
    // One loop per role - waves 0..3 transform forward AND run half an inverse transform, waves 4 and up only transform
    // forward - with the role a compile-time constant: in a common loop the compiler has to keep what either role holds
    // in registers alive through the other's code, runs at the register limit and serialises the LDS reads of the row
    // sum behind their uses (see k_blind_rotate_coop_fast, where this was measured).
    auto run = [&](auto inv_tag) {
    constexpr bool INV = decltype(inv_tag)::value;
    c64 kb[2][8];
    if (!INV && i < kLwe) load_key_row(kb, i);
#pragma unroll 1
    while (i < kLwe) {
        if (INV) load_key_row(kb, i);
        c64 v[8];
        const uint32_t *dcol = diff + c * kN + lane;
#pragma unroll
        for (int r = 0; r < 8; r++)
            v[r] = {(double) decomp_digit(dcol[br3(r) << 6], p), (double) decomp_digit(dcol[(br3(r) << 6) + kNh], p)};
        wave_fwd(v, lane, mineA, twu, twl);
        wave_sync();
#pragma unroll
        for (int r = 0; r < 8; r++) {  // P_w = v * key row w (LagrangeHalfCPolynomialAddMul on a cleared accumulator)
            const double ac0 = v[r].re * kb[0][r].re, bd0 = v[r].im * kb[0][r].im, ad0 = v[r].re * kb[0][r].im,
                         bc0 = v[r].im * kb[0][r].re;
            mineA[r * 64 + lane] = {ac0 - bd0, ad0 + bc0};
            const double ac1 = v[r].re * kb[1][r].re, bd1 = v[r].im * kb[1][r].im, ad1 = v[r].re * kb[1][r].im,
                         bc1 = v[r].im * kb[1][r].re;
            mineB[r * 64 + lane] = {ac1 - bd1, ad1 + bc1};
        }
        const int inext = next_active(i + 1);
        __syncthreads();  // all six products in LDS; every wave is done reading the differences
        // sum over the rows, in row order, by all threads: element e = (column, physical slot x)
        {   // all reads first, as in prepare_diff
            constexpr int kIter = (2 * kNh + kCoopThreads - 1) / kCoopThreads;
            c64 t[kIter];
#pragma unroll
            for (int k = 0; k < kIter; k++) {
                const int e = threadIdx.x + k * kCoopThreads;
                if (e < 2 * kNh) {
                    const c64 *src = bufs + (size_t) (e >> 9) * kNh + (e & (kNh - 1));
                    t[k] = src[0];
#pragma unroll
                    for (int w = 1; w < kCoopWaves; w++) {
                        const c64 u = src[(size_t) w * 2 * kNh];
                        t[k].re += u.re;
                        t[k].im += u.im;
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < kIter; k++) {
                const int e = threadIdx.x + k * kCoopThreads;
                if (e < 2 * kNh) *sum_slot(e >> 9, swz(e & (kNh - 1))) = t[k];
            }
        }
        __syncthreads();  // summed spectra in place
        if constexpr (!INV) {
            if (inext < kLwe) load_key_row(kb, inext);  // arrives during the inverse transforms
            __syncthreads();  // the same barrier as in the other branch
        } else {
            c64 E[8], O[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {  // level-0 operands of this wave's half, straight from the summed spectrum
                const int m0 = (br3(r) << 6) | lane;
                E[r] = *sum_slot(icol, swz(gray(m0)));
                const c64 t = *sum_slot(icol, swz(gray(511 - m0)));
                O[r] = {t.re, -t.im};
            }
            __syncthreads();  // both halves of a column have read the spectrum that lies over its accumulator mirror
            c64 v[8];
            int32_t res[8];
            // the twiddles of each phase are requested before the transpose that precedes it
            TwReg4<TWH_B> tb;
            TwReg4<TWH_C> tc;
            TwReg4<TWH_FIN> tf;
#pragma unroll
            for (int q = 0; q < 4; q++) tb.t[q] = twh(TWH_B + q);
            wave_sync();
            if (ih == 0) half_phaseA<0>(v, E, O, twu);
            else half_phaseA<1>(v, E, O, twu);
#pragma unroll
            for (int r = 0; r < 8; r++) mineA[swz(fwd_xA(lane, r))] = v[r];
            wave_sync();
#pragma unroll
            for (int q = 0; q < 4; q++) tc.t[q] = twh(TWH_C + q);
#pragma unroll
            for (int r = 0; r < 8; r++) v[r] = mineA[swz(fwd_xB(lane, r))];
            wave_sync();
            half_phaseBC<TWH_B>(v, tb);
#pragma unroll
            for (int r = 0; r < 8; r++) mineA[swz(fwd_xB(lane, r))] = v[r];
            wave_sync();
#pragma unroll
            for (int q = 0; q < 4; q++) tf.t[q] = twh(TWH_FIN + q);
#pragma unroll
            for (int r = 0; r < 8; r++) v[r] = mineA[swz(fwd_xC(lane, r))];
            wave_sync();
            half_phaseBC<TWH_C>(v, tc);
            half_final<true>(res, v, tf);
            int32_t *dst = acc + icol * kN;
#pragma unroll
            for (int r = 0; r < 8; r++) {  // tLweAddTo (tlwe-functions.cpp:149)
                own[r] = (int32_t) ((uint32_t) own[r] + (uint32_t) res[r]);
                dst[half_y(lane, r, ih)] = own[r];
            }
        }
        __syncthreads();  // accumulator updated
        if (inext < kLwe) prepare_diff(__builtin_amdgcn_readfirstlane(bara_lds[inext]));
        i = inext;
        __syncthreads();  // next differences in place
    }
    };
    if (wave < 4) run(std::true_type{});
    else run(std::false_type{});

    if (MODE == MODE_GATE) {
        int32_t *o = A.out + g * kExt;
        for (int j = threadIdx.x; j < kN; j += 64 * kCoopWaves) o[j] = j == 0 ? acc[0] : (int32_t) (0u - (uint32_t) acc[kN - j]);
        if (threadIdx.x == 0) o[kN] = acc[kN];
    } else {
        int32_t *io = A.acc_io + g * 2 * kN;
        for (int j = threadIdx.x; j < 2 * kN; j += 64 * kCoopWaves) io[j] = acc[j];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// latency shape in the FAST arithmetic mode: one ciphertext per workgroup of 2l forward waves + one inverse wave per
// accumulator column, with the folded transforms of fast_core.h.  Per CMux, four workgroup barriers:
//   forward wave w: digits of decomposition row w (from the prepared differences), forward transform, spectrum into its
//                   LDS buffer                                                                           -> barrier
//   every wave:     its share of the 2 x 512 FMA chains over the 2l rows (spectra from LDS x key values requested one CMux
//                   ahead), in the throughput kernel's order: a gate's ciphertext does not depend on which kernel its
//                   batch size selects; sums into the column's sum buffer                                -> barrier
//   inverse waves:  inverse transform of their column's sums (512 points, one wave), accumulator update  -> barrier
//   every wave:     one (column, quad) of the rotated, prepared differences of the next CMux from the accumulator mirror
//                   (TFHE_HIP_COOP_PREP; before: each inverse wave its whole column, 2.08 against 1.94 ms per gate; the
//                   forward waves gathering all of theirs for themselves: 2.22 ms)                        -> barrier
// Key values of the next CMux: forward waves request theirs after the second barrier, inverse waves at the end of theirs.
// Rotations by 0 are skipped through a list of the non-zero rotation amounts built once per launch.  The twiddles of
// both directions stay in registers (one ciphertext per CU: registers are not the constraint here).
// LDS: twiddle image 24 KB | accumulator mirror 8 KB | per forward wave: A (transpose scratch, then its spectrum) and B
//      (B of waves 1 and 2: the sums of column 0 / 1; A and B of wave 0: the inverse waves' scratch), 8 KB each
//      | rotation list | prepared differences 8 KB
// ---------------------------------------------------------------------------------------------------------------
// 2l forward waves, one inverse wave per accumulator column, and - where that leaves fewer than eight (the 80-bit set:
// 2l = 4) - helper waves that only take part in the product phase, so that its 16 units split evenly (two per wave)
constexpr int kCoopFastWaves = kKpl + 2 < 8 ? 8 : kKpl + 2;
constexpr int kCoopFastLds = brf::kFastLaneEntries * (int) sizeof(c64) + 2 * kN * 4 + kKpl * 2 * kNh * (int) sizeof(c64) +
                             2 * 640 * 4 + 2 * kN * 4;
static_assert(kCoopFastLds <= 160 * 1024, "LDS of one CU");
static_assert(kLwe < 639, "rotation list: 640 entries, the last one holds the count");
#ifndef TFHE_HIP_COOP_TRACE
#define TFHE_HIP_COOP_TRACE 0  // 1: per-phase clock totals of workgroup 0 on stdout (diagnostic build, tools/ko_build.sh)
#endif
// Prepared differences of one column: lane l holds, as four 16-byte words, the coefficients j = r<<6|l (quads 0, 1:
// r = 0..3, 4..7) and j + 512 (quads 2, 3) - word (column * 4 + quad) * 64 + lane, conflict-free on either side.
template <int MODE>
__global__ __launch_bounds__(64 * kCoopFastWaves) void k_blind_rotate_coop_fast(BrArgs A) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    c64 *tw_lds = reinterpret_cast<c64 *>(lds_raw);
    load_tw_to_lds(tw_lds, A.tw, brf::kFastLaneEntries);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    unsigned char *ptr = lds_raw + brf::kFastLaneEntries * sizeof(c64);
    int32_t *acc = reinterpret_cast<int32_t *>(ptr);
    c64 *bufs = reinterpret_cast<c64 *>(ptr + 2 * kN * 4);
    const bool fwd_role = wave < kKpl, inv_role = wave >= kKpl && wave < kKpl + 2;
    const int ic = inv_role ? wave - kKpl : 0;  // inverse role: accumulator column
    c64 *mineA = bufs + (size_t) (fwd_role ? wave : 0) * 2 * kNh, *mineB = mineA + kNh;
    int32_t *bara_lds = reinterpret_cast<int32_t *>(bufs + (size_t) kKpl * 2 * kNh);
    int32_t *act = bara_lds + 640;                                   // (index << 16 | amount) of the non-zero rotations; [639] = count
    uint32_t *diff = reinterpret_cast<uint32_t *>(act + 640);         // decomp_prepare((X^a - 1) * ACC), see diff_word
    const long long g = blockIdx.x;  // host launches exactly B workgroups
    const PhaseAConsts twu = A.ka;
    const TwLds twl{tw_lds + lane};
    const int c = wave >= kL ? 1 : 0, p = wave - c * kL;
    constexpr int kThreads = 64 * kCoopFastWaves;

    const int32_t *ga = nullptr, *gb = nullptr;
    int32_t sa = A.sa, sb = A.sb, cst = A.cst;
    const int32_t *gbara = nullptr;
    if (MODE == MODE_GATE) {
        ga = A.a + g * kCt;
        gb = A.b ? A.b + g * kCt : nullptr;
        if (A.jobs) {
            const BrJob *j = A.jobs + g;
            const int ia = __builtin_amdgcn_readfirstlane(j->ia), ib = __builtin_amdgcn_readfirstlane(j->ib);
            sa = __builtin_amdgcn_readfirstlane(j->sa);
            sb = __builtin_amdgcn_readfirstlane(j->sb);
            cst = __builtin_amdgcn_readfirstlane(j->cst);
            ga = A.a + (long long) ia * kCt;
            gb = ib >= 0 ? A.a + (long long) ib * kCt : nullptr;
        }
    } else {
        gbara = A.bara + g * kLwe;
    }
    auto comb = [&](int i) -> int32_t {
        uint32_t t = (uint32_t) sa * (uint32_t) ga[i];
        if (gb) t += (uint32_t) sb * (uint32_t) gb[i];
        return (int32_t) t;
    };
    for (int i = threadIdx.x; i < kLwe; i += kThreads)
        bara_lds[i] = MODE == MODE_GATE ? modswitch_2N(comb(i)) : (gbara[i] & (2 * kN - 1));
    if (MODE == MODE_GATE) {
        const int barb = modswitch_2N((int32_t) ((uint32_t) comb(kLwe) + (uint32_t) cst));
        for (int j = threadIdx.x; j < kN; j += kThreads) {
            acc[j] = 0;
            acc[kN + j] = ((j + barb) & kN) ? (int32_t) (0u - (uint32_t) A.mu) : A.mu;
        }
    } else {
        int32_t *io = A.acc_io + g * 2 * kN;
        for (int j = threadIdx.x; j < 2 * kN; j += kThreads) acc[j] = io[j];
    }
    __syncthreads();
    if (wave == 0) {  // a rotation by 0 is skipped (lwe-bootstrapping-functions-fft.cpp:154): list of the others, in order
        int cnt = 0;
        for (int base = 0; base < kLwe; base += 64) {
            const int idx = base + lane;
            const int a = idx < kLwe ? bara_lds[idx] : 0;
            const unsigned long long m = __ballot(a != 0);
            if (a != 0) act[cnt + __popcll(m & ((1ull << lane) - 1))] = (idx << 16) | a;
            cnt += __popcll(m);
        }
        if (lane == 0) act[639] = cnt;
    }
    __syncthreads();
    const int nact = __builtin_amdgcn_readfirstlane(act[639]);
    int32_t own_lo[8], own_hi[8];  // inverse waves: coefficients j = r<<6|lane and j + 512 of accumulator column ic
    if (inv_role) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
            own_lo[r] = acc[ic * kN + ((r << 6) | lane)];
            own_hi[r] = acc[ic * kN + ((r << 6) | lane) + kNh];
        }
    }
    // inverse wave: rotated, prepared differences of its column for rotation amount a (the column's mirror must be current)
    auto prepare_column = [&](int a) {
        const int32_t *col = acc + ic * kN;
        uint32_t d[16];
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int j = (r << 6) | lane | (h << 9);
                const int s = (j - a) & (2 * kN - 1);
                const uint32_t v = (uint32_t) col[s & (kN - 1)];
                const uint32_t rot = (s & kN) ? (0u - v) : v;
                d[h * 8 + r] = decomp_prepare((int32_t) (rot - (uint32_t) (h ? own_hi[r] : own_lo[r])));
            }
        u32x4 *dq = reinterpret_cast<u32x4 *>(diff) + (ic * 4) * 64 + lane;
#pragma unroll
        for (int q = 0; q < 4; q++) dq[q * 64] = u32x4{d[4 * q], d[4 * q + 1], d[4 * q + 2], d[4 * q + 3]};
    };
#if TFHE_HIP_COOP_PREP
    static_assert(kCoopFastWaves == 8, "one (column, quad) of the prepared differences per wave");
    // the same words, spread over all eight waves: wave w prepares quad w & 3 of column w >> 2 (four coefficients per lane,
    // one 16-byte store) from the accumulator mirror - a fourth barrier per CMux, but 1.1 k ticks less on the inverse
    // waves, which are the critical path
    auto prepare_quad = [&](int a) {
        const int cq = wave >> 2, q = wave & 3;
        const int32_t *col = acc + cq * kN;
        uint32_t d[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int j = ((((q & 1) << 2) | i) << 6) | lane | ((q >> 1) << 9);
            const int s = (j - a) & (2 * kN - 1);
            const uint32_t v = (uint32_t) col[s & (kN - 1)];
            const uint32_t rot = (s & kN) ? (0u - v) : v;
            d[i] = decomp_prepare((int32_t) (rot - (uint32_t) col[j]));
        }
        reinterpret_cast<u32x4 *>(diff)[(cq * 4 + q) * 64 + lane] = u32x4{d[0], d[1], d[2], d[3]};
    };
    if (nact > 0) prepare_quad(__builtin_amdgcn_readfirstlane(act[0]) & 0xffff);
#else
    if (inv_role && nact > 0) prepare_column(__builtin_amdgcn_readfirstlane(act[0]) & 0xffff);
#endif
    // The 2l x 2 products of a CMux are accumulated exactly as the throughput kernel does it (cmux_fast: one FMA chain per
    // spectrum value and column over the rows w = 0..2l-1 in order, starting from zero), so that a gate gives the same
    // ciphertext whichever kernel its batch size selects.  The chains are spread over ALL waves: unit u = (slice q = u & 7
    // of 64 spectrum values Z_(q<<6|lane), column u >> 3), wave handles the units wave, wave + W, ...: it reads the 2l
    // spectra of its slice from the forward waves' buffers, multiplies with its own 2l key values per unit (requested one
    // CMux ahead) and leaves the sum in the column's sum buffer.
    constexpr int kW = kCoopFastWaves, kUnits = (16 + kW - 1) / kW;
    c64 *sum0 = bufs + (size_t) 3 * kNh, *sum1 = bufs + (size_t) 5 * kNh;  // the B halves of forward waves 1 and 2 (unused otherwise)
    static_assert(kKpl >= 3, "sum buffers live in the B halves of forward waves 1 and 2");
    auto load_mac_keys = [&](c64 (&kb)[kUnits][kKpl], int it) {  // 1 KB coalesced loads, Z_k order
#pragma unroll
        for (int k = 0; k < kUnits; k++) {
            const int u = wave + k * kW;
            if (u < 16) {
                const c64 *src = A.bk + ((size_t) it * kKpl * 2 + (u >> 3)) * kNh + ((u & 7) << 6) + lane;
#pragma unroll
                for (int w = 0; w < kKpl; w++) kb[k][w] = src[(size_t) w * 2 * kNh];
            }
        }
    };
    auto mac_units = [&](const c64 (&kb)[kUnits][kKpl]) {
        if (kW == 8) {  // both units of a wave share the slice: one read of the spectra
            c64 a[kKpl];
#pragma unroll
            for (int w = 0; w < kKpl; w++) a[w] = bufs[(size_t) w * 2 * kNh + (wave << 6) + lane];
            c64 s0 = {0.0, 0.0}, s1 = {0.0, 0.0};
#pragma unroll
            for (int w = 0; w < kKpl; w++) {
                brf::cmac(s0, a[w], kb[0][w]);
                brf::cmac(s1, a[w], kb[kUnits - 1][w]);
            }
            sum0[(wave << 6) + lane] = s0;
            sum1[(wave << 6) + lane] = s1;
        } else {
#pragma unroll
            for (int k = 0; k < kUnits; k++) {
                const int u = wave + k * kW;
                if (u < 16) {
                    c64 a[kKpl];
#pragma unroll
                    for (int w = 0; w < kKpl; w++) a[w] = bufs[(size_t) w * 2 * kNh + ((u & 7) << 6) + lane];
                    c64 s = {0.0, 0.0};
#pragma unroll
                    for (int w = 0; w < kKpl; w++) brf::cmac(s, a[w], kb[k][w]);
                    (u >> 3 ? sum1 : sum0)[((u & 7) << 6) + lane] = s;
                }
            }
        }
    };
    c64 kb[kUnits][kKpl];
    TwPre<8> twa;  // levels 3..8 of the role's transform
    if (nact > 0) load_mac_keys(kb, __builtin_amdgcn_readfirstlane(act[0]) >> 16);
    if (fwd_role) tw_preload(twa, twl, brf::FS_FB);
    else tw_preload(twa, twl, brf::FS_IB);
    __syncthreads();
#if TFHE_HIP_COOP_TRACE
    long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp;
#define TR(k) { const long long now = clock64(); tr[k] += now - tp; tp = now; }
#else
#define TR(k)
#endif
    // one loop per role (the roles never change: what one role keeps in registers is dead in the other's loop); both run
    // the same three barriers per CMux: spectra in LDS | sums in LDS | next differences in LDS
    if (fwd_role) {
#pragma unroll 1
        for (int pos = 0; pos < nact; pos++) {
#if TFHE_HIP_COOP_TRACE
            tp = clock64();
#endif
            const int nxt = pos + 1 < nact ? __builtin_amdgcn_readfirstlane(act[pos + 1]) : 0;
            c64 v[8];
            {
                const u32x4 *dq = reinterpret_cast<const u32x4 *>(diff) + (c * 4) * 64 + lane;
                const u32x4 q0 = dq[0], q1 = dq[64], q2 = dq[128], q3 = dq[192];
                const uint32_t lo[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
                const uint32_t hi[8] = {q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
#pragma unroll
                for (int r = 0; r < 8; r++)
                    v[r] = {(double) decomp_digit_rt(lo[br3(r)], p), (double) decomp_digit_rt(hi[br3(r)], p)};
            }
            TR(0)
            brf::fwd_phaseA(v, twu);
            t_write_A(mineA, v, lane);
            wave_sync();
            t_read_B(v, mineA, lane);
            wave_sync();
            brf::fwd_phaseBC<brf::FS_FB>(v, TwView<8, brf::FS_FB>{twa});
            t_write_B(mineA, v, lane);
            wave_sync();
            t_read_C(v, mineA, lane);
            wave_sync();
            brf::fwd_phaseBC<brf::FS_FC>(v, TwView<8, brf::FS_FB>{twa});
            TR(1)
#pragma unroll
            for (int r = 0; r < 8; r++) mineA[r * 64 + lane] = v[r];  // spectrum of row w: Z_(r<<6|lane)
            __syncthreads();  // all spectra in LDS; the forward waves are done with the differences
            TR(3)
            mac_units(kb);
            TR(2)
            __syncthreads();  // sums in place
            if (nxt) load_mac_keys(kb, nxt >> 16);  // arrives during the inverse transforms
#if TFHE_HIP_COOP_PREP
            __syncthreads();  // accumulator mirror updated; the inverse waves are done with the scratch
            if (nxt) prepare_quad(nxt & 0xffff);
#endif
            __syncthreads();  // next differences in place
            TR(3)
        }
    } else if (kCoopFastWaves > kKpl + 2 && !inv_role) {  // helper wave: products only
#pragma unroll 1
        for (int pos = 0; pos < nact; pos++) {
            const int nxt = pos + 1 < nact ? __builtin_amdgcn_readfirstlane(act[pos + 1]) : 0;
            __syncthreads();
            mac_units(kb);
            __syncthreads();
            if (nxt) load_mac_keys(kb, nxt >> 16);
#if TFHE_HIP_COOP_PREP
            __syncthreads();
            if (nxt) prepare_quad(nxt & 0xffff);
#endif
            __syncthreads();
        }
    } else {
#pragma unroll 1
        for (int pos = 0; pos < nact; pos++) {
#if TFHE_HIP_COOP_TRACE
            tp = clock64();
#endif
            const int nxt = pos + 1 < nact ? __builtin_amdgcn_readfirstlane(act[pos + 1]) : 0;
            __syncthreads();
            TR(3)
            mac_units(kb);
            TR(2)
            __syncthreads();
            TR(3)
            c64 sp[8];
            {
                const c64 *src = (ic ? sum1 : sum0) + lane;
#pragma unroll
                for (int r = 0; r < 8; r++) sp[r] = src[br3(r) << 6];  // the inverse wants Y_(br3(r)<<6|lane) in register r
            }
            TR(4)
            // transpose scratch: a buffer nobody else touches until the next barrier (column ic of forward wave 0)
            c64 *iscr = ic ? mineB : mineA;
            brf::inv_phaseA(sp, twu);
            t_write_A(iscr, sp, lane);
            wave_sync();
            t_read_B(sp, iscr, lane);
            wave_sync();
            brf::inv_phaseBC<brf::FS_IB>(sp, TwView<8, brf::FS_IB>{twa});
            t_write_B(iscr, sp, lane);
            wave_sync();
            t_read_C(sp, iscr, lane);
            TwPre<8> twb;  // untwist factors, requested ahead of the last three levels
            tw_preload(twb, twl, brf::FS_IF);
            wave_sync();
            brf::inv_phaseBC<brf::FS_IC>(sp, TwView<8, brf::FS_IB>{twa});
            int32_t lo[8], hi[8];
            brf::inv_final<true>(lo, hi, sp, TwView<8, brf::FS_IF>{twb});
            int32_t *dst = acc + ic * kN;
#pragma unroll
            for (int r = 0; r < 8; r++) {  // tLweAddTo (tlwe-functions.cpp:149)
                own_lo[r] = (int32_t) ((uint32_t) own_lo[r] + (uint32_t) lo[r]);
                own_hi[r] = (int32_t) ((uint32_t) own_hi[r] + (uint32_t) hi[r]);
                dst[(r << 6) | lane] = own_lo[r];
                dst[((r << 6) | lane) + kNh] = own_hi[r];
            }
            TR(5)
#if TFHE_HIP_COOP_PREP
            __syncthreads();  // both columns of the mirror are current
            if (nxt) {
                prepare_quad(nxt & 0xffff);
#else
            if (nxt) {
                wave_sync();
                prepare_column(nxt & 0xffff);
#endif
                // this wave's key values of the next CMux: requested only now, when the forward waves' requests (issued
                // after the second barrier, 72 KB through the CU's one vector-memory pipe) have drained - issued next to
                // them they queue behind them and hold up the inverse transform, which is the critical path
                load_mac_keys(kb, nxt >> 16);
            }
            TR(6)
            __syncthreads();
            TR(3)
        }
    }
#if TFHE_HIP_COOP_TRACE
    if (blockIdx.x == 0 && lane == 0)
        printf("wave %d: digits %lld fwd %lld products %lld barriers %lld sum %lld inverse %lld prepare %lld\n", wave, tr[0], tr[1],
               tr[2], tr[3], tr[4], tr[5], tr[6]);
#endif
#undef TR
    if (MODE == MODE_GATE) {
        int32_t *o = A.out + g * kExt;
        for (int j = threadIdx.x; j < kN; j += kThreads) o[j] = j == 0 ? acc[0] : (int32_t) (0u - (uint32_t) acc[kN - j]);
        if (threadIdx.x == 0) o[kN] = acc[kN];
    } else {
        int32_t *io = A.acc_io + g * 2 * kN;
        for (int j = threadIdx.x; j < 2 * kN; j += kThreads) io[j] = acc[j];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// keyswitch: lweKeySwitch (lwe-keyswitch-functions.cpp:228-238) + lweKeySwitchTranslate_fromArray (:114-129)
//   out = (0, b') - sum_{i<N, j<t} ks[i][j][digit_ij],  digit_ij = ((a'_i + 2^15) >> (30 - 2j)) & 3, digit 0 skipped
// grid (ceil(B/G), 3), block 256: thread = one output coefficient q, G ciphertexts in registers.
// ---------------------------------------------------------------------------------------------------------------
// grid ceil(B/32), block 160: thread = 4 consecutive output coefficients (one 16-byte piece of every key row),
// 32 ciphertexts in registers, so each key row fetched from L2 serves 32 ciphertexts.  The digit selects the row
// through two lane-uniform bit masks (no divergence): sel = d1 ? (d0 ? row3 : row2) : (d0 ? row1 : 0).
constexpr int kKsG = 32;
constexpr int kKsThreads = kKsRow / 4;  // 158
__global__ __launch_bounds__(192) void k_keyswitch(const int32_t *__restrict__ in, const int32_t *__restrict__ ks,
                                                   int32_t *__restrict__ out, long long B) {
    const int t = threadIdx.x;
    if (t >= kKsThreads) return;
    const long long g0 = (long long) blockIdx.x * kKsG;
    u32x4 acc[kKsG];
#pragma unroll
    for (int g = 0; g < kKsG; g++) {
        const long long gi = (g0 + g < B) ? g0 + g : B - 1;
        acc[g] = u32x4{0u, 0u, 0u, 0u};
        if (t == kLwe / 4) acc[g][kLwe % 4] = (uint32_t) in[gi * kExt + kN];  // coefficient n = b
    }
    const uint32_t prec_offset = 1u << (32 - (1 + kKsBasebit * kKsT));
    const u32x4 *rows = reinterpret_cast<const u32x4 *>(ks) + t;
#pragma unroll 1
    for (int i = 0; i < kN; i++) {
        uint32_t abar[kKsG];
#pragma unroll
        for (int g = 0; g < kKsG; g++) {
            const long long gi = (g0 + g < B) ? g0 + g : B - 1;
            abar[g] = (uint32_t) in[gi * kExt + i] + prec_offset;
        }
#pragma unroll 2
        for (int j = 0; j < kKsT; j++) {
            const u32x4 *rp = rows + (size_t) (i * kKsT + j) * 3 * (kKsRow / 4);
            const u32x4 r1 = rp[0], r2 = rp[kKsRow / 4], r3 = rp[2 * (kKsRow / 4)];
#pragma unroll
            for (int g = 0; g < kKsG; g++) {
                const uint32_t m0 = (uint32_t) ((int32_t) (abar[g] << (1 + 2 * j)) >> 31);  // digit bit 0
                const uint32_t m1 = (uint32_t) ((int32_t) (abar[g] << (2 * j)) >> 31);      // digit bit 1
                const u32x4 lo = r1 & m0;
                const u32x4 hi = (r3 & m0) | (r2 & ~m0);
                acc[g] -= (hi & m1) | (lo & ~m1);
            }
        }
    }
#pragma unroll
    for (int g = 0; g < kKsG; g++) {
        if (g0 + g >= B) break;
        int32_t *o = out + (g0 + g) * kCt + 4 * t;
        o[0] = (int32_t) acc[g].x;
        o[1] = (int32_t) acc[g].y;
        o[2] = (int32_t) acc[g].z;
        if (4 * t + 3 < kCt) o[3] = (int32_t) acc[g].w;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// keyswitch on the matrix cores.  The gather-subtract above is a product of a one-hot digit matrix with the key:
//     sum_{i,j} ks[i][j][d_gij][q]  =  sum_k A[g][k] * K[k][q],   k = (i, j, d),  A[g][(i,j,d)] = [d_gij == d]
// (the d = 0 column meets an all-zero key row).  K is split into four signed byte limbs, K = sum_l s_l 256^l with
// s_l in [-128,127], so each limb product is an exact int8 x int8 -> int32 MFMA (|sum| <= 8192*128 < 2^31) and
// out = (0,b') - sum_l (A s_l) << 8l  (mod 2^32) is the reference's result bit for bit.
// v_mfma_i32_32x32x32_i8: M = 32 ciphertexts, N = 32 output coefficients, K = 32 = one input coefficient i
// (8 digits j x 4 digit values d).  A fragments are built in registers from the transposed input; B fragments are
// 16-byte loads from the key stored as int8[i][limb][q][j][d].
// ---------------------------------------------------------------------------------------------------------------
constexpr int kKsQ = (kCt + 127) / 128 * 128;  // output coefficients padded to whole workgroups of 4 x 32 (640)
constexpr int kKsYB = kKsQ / 32;             // workgroups over the coefficients (20), each 4 waves x 32*MT ciphertexts
constexpr int kKsMT = 2;      // 32-ciphertext tiles per wave
#ifndef TFHE_HIP_KS_ONEHOT_LDS
#define TFHE_HIP_KS_ONEHOT_LDS 0  // A/B switch of k_keyswitch_mfma's A operand: LDS table (1) or VALU shifts (0, default: measured faster)
#endif
typedef int v4i_t __attribute__((ext_vector_type(4)));
typedef int v16i_t __attribute__((ext_vector_type(16)));

// [N][t][4][n+1] int32 -> [N][4 limbs][640][t][4] int8
__global__ void k_ks_pack_mfma(int8_t *__restrict__ dst, const int32_t *__restrict__ src) {
    const int i = blockIdx.x;
    for (int q = threadIdx.x; q < kKsQ; q += blockDim.x) {
        uint32_t limb[4][8];  // [limb][j] packs 4 bytes (d = 0..3)
#pragma unroll
        for (int l = 0; l < 4; l++)
#pragma unroll
            for (int j = 0; j < 8; j++) limb[l][j] = 0;
        if (q < kCt) {
#pragma unroll
            for (int j = 0; j < kKsT; j++)
#pragma unroll
                for (int d = 1; d < 4; d++) {
                    uint32_t v = (uint32_t) src[(((size_t) i * kKsT + j) * 4 + d) * kCt + q];
#pragma unroll
                    for (int l = 0; l < 4; l++) {
                        const int32_t sb = (int32_t) (int8_t) (v & 0xffu);  // signed low byte
                        limb[l][j] |= ((uint32_t) sb & 0xffu) << (8 * d);
                        v = (v - (uint32_t) sb) >> 8;                        // exact: low byte is now zero
                    }
                }
        }
#pragma unroll
        for (int l = 0; l < 4; l++) {
            uint32_t *o = reinterpret_cast<uint32_t *>(dst + (((size_t) i * 4 + l) * kKsQ + q) * 32);
#pragma unroll
            for (int j = 0; j < 8; j++) o[j] = limb[l][j];
        }
    }
}

// in[B][N+1] -> abarT = a_i + 2^15 (lwe-keyswitch-functions.cpp:119,123) in the order the keyswitch waves read it:
// [ciphertext tile of 64][i / 4][M-tile of 32][ciphertext in tile][i % 4], so that a lane's four consecutive steps are
// one 16-byte load; padding ciphertexts get digit 0
__device__ __forceinline__ size_t ks_abar_index4(long long g, int i) {  // index of the 16-byte group holding (g, i)
    return ((((size_t) (g >> 6) * (kN / 4) + (i >> 2)) * 2 + ((g >> 5) & 1)) << 5) + (g & 31);
}
__global__ void k_ks_transpose(uint32_t *__restrict__ abarT, const int32_t *__restrict__ in, long long B, long long Bp,
                               const KsJob *__restrict__ kj) {
    __shared__ uint32_t tile[32][33];
    const long long g0 = (long long) blockIdx.x * 32;
    const int i0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const uint32_t prec_offset = 1u << (32 - (1 + kKsBasebit * kKsT));
    for (int r = ty; r < 32; r += 8) {
        const long long g = g0 + r;
        uint32_t v = 0u;
        if (g < B) {
            if (kj) {  // netlist level: sum of one or two extracted samples (bootsMUX, boot-gates.cpp:357-361)
                const KsJob j = kj[g];
                v = (uint32_t) in[(long long) j.e1 * kExt + i0 + tx];
                if (j.e2 >= 0) v += (uint32_t) in[(long long) j.e2 * kExt + i0 + tx];
            } else {
                v = (uint32_t) in[g * kExt + i0 + tx];
            }
            v += prec_offset;
        }
        tile[r][tx] = v;
    }
    __syncthreads();
    (void) Bp;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const int gl = threadIdx.x & 31, i4 = threadIdx.x >> 5;  // 32 ciphertexts x 8 groups of four steps
    reinterpret_cast<u32x4 *>(abarT)[ks_abar_index4(g0 + gl, i0 + 4 * i4)] =
        u32x4{tile[gl][4 * i4], tile[gl][4 * i4 + 1], tile[gl][4 * i4 + 2], tile[gl][4 * i4 + 3]};
}

// grid (ceil(Bp/(4*32*MT)), 20, S), block 256: block (x, y) owns the 32 coefficients [32y, 32y+32) and 4 x 32*MT
// ciphertexts, wave w of it the ciphertexts [(4x+w) 32 MT, +32 MT).  The four waves of a block therefore read the SAME key
// tile in every step (they run the same loop side by side, so three of the four reads are served by the CU's vector
// cache), and each key byte leaves L2 once per 256 ciphertexts instead of once per 64: the kernel was bound by exactly
// that stream (21.5 GB through L2 per 16,384-ciphertext launch).  All four limbs accumulate in one pass over i (A
// fragments built once per i).
// Small batches cannot fill the chip with x * 20 blocks (a single gate: 20 blocks streaming 84 MB of key), so the
// input coefficients i are split S ways (blockIdx.z) into `partial`, and k_ks_reduce adds the parts (integer sums:
// any order gives the same bits).
__global__ __launch_bounds__(256, 2) void k_keyswitch_mfma(const uint32_t *__restrict__ abarT, const int8_t *__restrict__ kl,
                                                        const int32_t *__restrict__ in, int32_t *__restrict__ out,
                                                        long long B, long long Bp, const KsJob *__restrict__ kj,
                                                        uint32_t *__restrict__ partial) {
    // The key tiles (4 limbs x 1 KB per step) are fetched once per block - wave w brings limb w - and handed to the other
    // waves through LDS, a GROUP of four steps at a time: two buffers of four tiles, ONE workgroup barrier per group.
    // During group g a wave requests its limbs of group g+2 and, after the group's matrix instructions, writes those of
    // group g+1 (requested one group earlier) into the other buffer; at the barrier that opens group g every wave has
    // finished reading the buffer that is overwritten next, and the tiles written during group g-1 are visible.  The loop body has no branches (the last
    // groups re-request and re-write the final group instead), so that the compiler can count its waits instead of
    // draining the queues at block boundaries: one barrier and no drained wait per step took 1.47 -> see DESIGN.md 4.2.
    __shared__ v4i_t tile[2][4][4][64];  // [buffer][step in group][limb][lane]
    // The A operand is one-hot: byte `digit` of dword jj is 1.  A/B variant (TFHE_HIP_KS_ONEHOT_LDS=1): two digits (four bits
    // of the digit word) index a table of 16 eight-byte entries {1 << 8 d, 1 << 8 d'} - two shift-and-mask pairs and two
    // conflict-free ds_read_b64 per operand instead of twelve shift / mask / shift instructions on the VALU.  Measured on one
    // box, 16,384 ciphertexts: 1.153 ms against 1.130 ms for the VALU form, 37 ciphertexts 0.041 ms either way: the A operand
    // is not what holds the matrix instructions back, and the table's LDS latency costs more than the VALU work it saves.
    __shared__ uint2 onehot[16];
    if (threadIdx.x < 16) onehot[threadIdx.x] = uint2{1u << (8 * (threadIdx.x >> 2)), 1u << (8 * (threadIdx.x & 3))};
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r32 = lane & 31, h = lane >> 5;
    const uint32_t sh_hi = 25 - 8 * h, sh_lo = 21 - 8 * h;  // digits 4h, 4h+1 | 4h+2, 4h+3 of a digit word, as table offsets
    const long long g0 = ((long long) blockIdx.x * 4 + wave) * (32 * kKsMT);
    const bool live = g0 < Bp;  // ragged last block: a wave without ciphertexts still fetches its limb and meets the barriers
    const int q = blockIdx.y * 32 + r32;
    const u32x4 *ap = reinterpret_cast<const u32x4 *>(abarT) + ks_abar_index4(live ? g0 + r32 : r32, 0);  // + 64 per i/4, + 32 per M-tile
    const v4i_t *bp = reinterpret_cast<const v4i_t *>(kl) + (size_t) q * 2 + h;
    v16i_t acc[4][kKsMT];
#pragma unroll
    for (int l = 0; l < 4; l++)
#pragma unroll
        for (int m = 0; m < kKsMT; m++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[l][m][r] = 0;
    const int i_per = kN / (int) gridDim.z, i_lo = (int) blockIdx.z * i_per;  // multiples of 64
    const int groups = i_per / 4;                                             // >= 16
    auto key = [&](int i) { return bp[((size_t) i * 4 + wave) * kKsQ * 2]; };
    v4i_t gk[4];                 // this wave's limb of the four steps of a coming group
    u32x4 a4[kKsMT], a4n[kKsMT];  // digit words of the current / next group (four steps each)
#pragma unroll
    for (int ii = 0; ii < 4; ii++) tile[0][ii][wave][lane] = key(i_lo + ii);
#pragma unroll
    for (int ii = 0; ii < 4; ii++) gk[ii] = key(i_lo + 4 + ii);
#pragma unroll
    for (int m = 0; m < kKsMT; m++) a4n[m] = ap[(size_t) (i_lo >> 2) * 64 + 32 * m];
#pragma unroll 1
    for (int g = 0; g < groups; g++) {
        const int buf = g & 1;
        __syncthreads();
        v4i_t b[2][4];
#pragma unroll
        for (int l = 0; l < 4; l++) b[0][l] = tile[buf][0][l][lane];
        // requests for later groups go out first and stay here (the scheduler otherwise sinks them to their uses)
        const int ga = g + 1 < groups ? g + 1 : groups - 1;
#pragma unroll
        for (int m = 0; m < kKsMT; m++) {
            a4[m] = a4n[m];
            a4n[m] = ap[(size_t) ((i_lo >> 2) + ga) * 64 + 32 * m];
        }
        v4i_t gk2[4];  // this wave's limb of group g + 2
        const int gn = g + 2 < groups ? g + 2 : groups - 1;
#pragma unroll
        for (int ii = 0; ii < 4; ii++) gk2[ii] = key(i_lo + 4 * gn + ii);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ii = 0; ii < 4; ii++) {
            if (ii < 3) {
#pragma unroll
                for (int l = 0; l < 4; l++) b[(ii + 1) & 1][l] = tile[buf][ii + 1][l][lane];
            }
#pragma unroll
            for (int m = 0; m < kKsMT; m++) {
                const uint32_t ab = a4[m][ii];
                // digit j = 4h + jj: ((ab >> (30 - 2j)) & 3); one-hot byte position = digit
#if TFHE_HIP_KS_ONEHOT_LDS
                const uint2 p0 = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(onehot) + ((ab >> sh_hi) & 0x78u));
                const uint2 p1 = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(onehot) + ((ab >> sh_lo) & 0x78u));
                const v4i_t a = {(int) p0.x, (int) p0.y, (int) p1.x, (int) p1.y};
#else
                v4i_t a;
#pragma unroll
                for (int jj = 0; jj < 4; jj++) a[jj] = (int) (1u << ((ab >> (27 - 2 * jj - 8 * h)) & 24u));
                (void) sh_hi, (void) sh_lo;
#endif
#pragma unroll
                for (int l = 0; l < 4; l++)
                    acc[l][m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b[ii & 1][l], acc[l][m], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // tiles of group g + 1 (requested during group g - 1) into the buffer that group g - 1 was read from
#pragma unroll
        for (int ii = 0; ii < 4; ii++) tile[buf ^ 1][ii][wave][lane] = gk[ii];
#pragma unroll
        for (int ii = 0; ii < 4; ii++) gk[ii] = gk2[ii];
    }
    if (!live) return;
    if (q > kLwe) return;
#pragma unroll
    for (int m = 0; m < kKsMT; m++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const long long g = g0 + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (g < B) {
                const uint32_t total = (uint32_t) acc[0][m][r] + ((uint32_t) acc[1][m][r] << 8) +
                                       ((uint32_t) acc[2][m][r] << 16) + ((uint32_t) acc[3][m][r] << 24);
                if (partial) {
                    partial[((size_t) blockIdx.z * Bp + g) * kCt + q] = total;
                    continue;
                }
                uint32_t init = 0u;
                long long orow = g;
                if (kj) {
                    const KsJob j = kj[g];
                    orow = j.out;
                    if (q == kLwe) {
                        init = (uint32_t) in[(long long) j.e1 * kExt + kN] + (uint32_t) j.bconst;
                        if (j.e2 >= 0) init += (uint32_t) in[(long long) j.e2 * kExt + kN];
                    }
                } else if (q == kLwe) {
                    init = (uint32_t) in[g * kExt + kN];
                }
                out[orow * kCt + q] = (int32_t) (init - total);
            }
        }
}

// out = (0, b') - sum of the S partial sums (see k_keyswitch_mfma); one thread per (ciphertext, coefficient)
__global__ __launch_bounds__(256) void k_ks_reduce(const uint32_t *__restrict__ partial, int S, const int32_t *__restrict__ in,
                                                   int32_t *__restrict__ out, long long B, long long Bp,
                                                   const KsJob *__restrict__ kj) {
    const long long idx = (long long) blockIdx.x * 256 + threadIdx.x;
    if (idx >= B * kCt) return;
    const long long g = idx / kCt;
    const int q = (int) (idx - g * kCt);
    uint32_t total = 0u;
    for (int sidx = 0; sidx < S; sidx++) total += partial[((size_t) sidx * Bp + g) * kCt + q];
    uint32_t init = 0u;
    long long orow = g;
    if (kj) {
        const KsJob j = kj[g];
        orow = j.out;
        if (q == kLwe) {
            init = (uint32_t) in[(long long) j.e1 * kExt + kN] + (uint32_t) j.bconst;
            if (j.e2 >= 0) init += (uint32_t) in[(long long) j.e2 * kExt + kN];
        }
    } else if (q == kLwe) {
        init = (uint32_t) in[g * kExt + kN];
    }
    out[orow * kCt + q] = (int32_t) (init - total);
}

// ---------------------------------------------------------------------------------------------------------------
// small element-wise kernels of the gate layer
// ---------------------------------------------------------------------------------------------------------------
// bootsMUX tail (boot-gates.cpp:357-361): u1 <- (0,MU) + u1 + u2 on LWE_N samples
__global__ void k_mux_combine(int32_t *u1, const int32_t *u2, int32_t mu, long long total) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    uint32_t v = (uint32_t) u1[i] + (uint32_t) u2[i];
    if (i % kExt == kN) v += (uint32_t) mu;
    u1[i] = (int32_t) v;
}
// bootsNOT / bootsCOPY / bootsCONSTANT (boot-gates.cpp:164-189)
__global__ void k_linear_gate(int32_t *out, const int32_t *a, int gate, int32_t mu, long long total) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int32_t v;
    if (gate == TFHE_HIP_NOT) v = (int32_t) (0u - (uint32_t) a[i]);
    else if (gate == TFHE_HIP_COPY) v = a[i];
    else v = (i % kCt == kLwe) ? (gate == TFHE_HIP_CONST1 ? mu : (int32_t) (0u - (uint32_t) mu)) : 0;
    out[i] = v;
}

// netlist level: NOT / COPY / CONSTANT on pool rows; desc = {gate, out, a}
__global__ void k_linear_level(int32_t *pool, const int32_t *desc, int32_t mu, long long count) {
    const long long g = blockIdx.x;
    if (g >= count) return;
    const int gate = desc[3 * g], o = desc[3 * g + 1], a = desc[3 * g + 2];
    for (int q = threadIdx.x; q < kCt; q += blockDim.x) {
        int32_t v;
        if (gate == TFHE_HIP_NOT) v = (int32_t) (0u - (uint32_t) pool[(long long) a * kCt + q]);
        else if (gate == TFHE_HIP_COPY) v = pool[(long long) a * kCt + q];
        else v = q == kLwe ? (gate == TFHE_HIP_CONST1 ? mu : (int32_t) (0u - (uint32_t) mu)) : 0;
        pool[(long long) o * kCt + q] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// key conversion
// ---------------------------------------------------------------------------------------------------------------
// tGswToFFTConvert (tgsw-fft-operations.cpp:52-57) = TorusPolynomial_ifft (polynomials.cpp:364-376) on each of
// the n*6*2 polynomials; result stored in physical slot order [poly][r][lane]
__global__ __launch_bounds__(256) void k_bk_convert(const int32_t *__restrict__ coef, c64 *__restrict__ bk,
                                                    const c64 *__restrict__ tw, PhaseAConsts ka, int npoly) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    c64 *tw_lds = reinterpret_cast<c64 *>(lds_raw);
    load_tw_to_lds(tw_lds, tw);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    c64 *scratch = reinterpret_cast<c64 *>(lds_raw + kTwLaneEntries * sizeof(c64)) + wave * kNh;
    const int poly = blockIdx.x * 4 + wave;
    if (poly >= npoly) return;
    const int32_t *p = coef + (size_t) poly * kN;
    c64 v[8];
    const double s = 1.0 / 4294967296.0;  // 2 * 2^-33: the size-2 merge doubles the reference's 2^-33 scaling exactly
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int j = (br3(r) << 6) | lane;
        v[r] = {p[j] * s, p[j + kNh] * s};
    }
    wave_fwd(v, lane, scratch, ka, TwLds{tw_lds + lane});
#pragma unroll
    for (int r = 0; r < 8; r++) bk[(size_t) poly * kNh + r * 64 + lane] = v[r];
}
// reference (logical LagrangeHalfC) order <-> physical slot order: slot x holds bin igray(x)
__global__ void k_bk_permute(c64 *dst, const c64 *src, long long total, int to_physical) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long long poly = i >> 9;
    const int x = (int) (i & 511);
    if (to_physical) dst[i] = src[poly * kNh + igray(x)];
    else dst[poly * kNh + igray(x)] = src[i];
}
// [N][t][4][n+1] -> [N][t][3][632] (digit 0 rows are never read: lwe-keyswitch-functions.cpp:126)
__global__ void k_ks_pack(int32_t *dst, const int32_t *src, long long rows) {
    const long long row = blockIdx.x;
    if (row >= rows) return;
    const long long ij = row / 3;
    const int d = (int) (row % 3) + 1;
    const int32_t *s = src + (ij * 4 + d) * kCt;
    int32_t *o = dst + row * kKsRow;
    for (int q = threadIdx.x; q < kKsRow; q += blockDim.x) o[q] = q < kCt ? s[q] : 0;
}

// ---------------------------------------------------------------------------------------------------------------
// stage-level transforms (one wave per polynomial, 4 waves per block)
// ---------------------------------------------------------------------------------------------------------------
// IntPolynomial_ifft (polynomials.cpp:347) / TorusPolynomial_ifft (:364); spectra written in the reference order
__global__ __launch_bounds__(256) void k_poly_fwd(const int32_t *__restrict__ polys, c64 *__restrict__ spectra,
                                                  const c64 *__restrict__ tw, PhaseAConsts ka, long long count, int torus) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    c64 *tw_lds = reinterpret_cast<c64 *>(lds_raw);
    load_tw_to_lds(tw_lds, tw);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    c64 *scratch = reinterpret_cast<c64 *>(lds_raw + kTwLaneEntries * sizeof(c64)) + wave * kNh;
    const long long poly = (long long) blockIdx.x * 4 + wave;
    if (poly >= count) return;
    const int32_t *p = polys + poly * kN;
    const double s = torus ? 1.0 / 4294967296.0 : 1.0;
    c64 v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int j = (br3(r) << 6) | lane;
        v[r] = {p[j] * s, p[j + kNh] * s};
    }
    wave_fwd(v, lane, scratch, ka, TwLds{tw_lds + lane});
#pragma unroll
    for (int r = 0; r < 8; r++) spectra[poly * kNh + igray(fwd_xC(lane, r))] = v[r];
}
// TorusPolynomial_fft (polynomials.cpp:378-393); spectra read in the reference order
__global__ __launch_bounds__(256) void k_poly_inv(const c64 *__restrict__ spectra, int32_t *__restrict__ polys,
                                                  const c64 *__restrict__ tw, PhaseAConsts ka, long long count) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    c64 *tw_lds = reinterpret_cast<c64 *>(lds_raw);
    load_tw_to_lds(tw_lds, tw);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    c64 *scratch = reinterpret_cast<c64 *>(lds_raw + kTwLaneEntries * sizeof(c64)) + wave * kNh;
    const long long poly = (long long) blockIdx.x * 4 + wave;
    if (poly >= count) return;
    const c64 *C = spectra + poly * kNh;
    c64 v[16];
#pragma unroll
    for (int rho = 0; rho < 16; rho++) {
        const int m = (br4(rho) << 6) | lane;
        if (br4(rho) < 8) v[rho] = C[m];
        else {
            const c64 t = C[1023 - m];
            v[rho] = {t.re, -t.im};
        }
    }
    int32_t res[16];
    wave_inv<false>(res, v, lane, scratch, ka, TwLds{tw_lds + lane});
#pragma unroll
    for (int rho = 0; rho < 16; rho++) polys[poly * kN + inv_yC(lane, rho)] = res[rho];
}
// torusPolynomialMultFFT (polynomials.cpp:241-248): out = ipoly * tpoly
__global__ __launch_bounds__(256) void k_poly_mul(const int32_t *__restrict__ ipolys, const int32_t *__restrict__ tpolys,
                                                  int32_t *__restrict__ out, const c64 *__restrict__ tw,
                                                  PhaseAConsts ka, long long count) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    c64 *tw_lds = reinterpret_cast<c64 *>(lds_raw);
    load_tw_to_lds(tw_lds, tw);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    c64 *scratch = reinterpret_cast<c64 *>(lds_raw + kTwLaneEntries * sizeof(c64)) + wave * kNh;
    const PhaseAConsts twu = ka;
    const TwLds twl{tw_lds + lane};
    // a wave walks over products with the stride of the grid (the host launches at most a few workgroups per CU): the
    // twiddle image is fetched once per workgroup, not once per four products
#pragma unroll 1
    for (long long poly = (long long) blockIdx.x * 4 + wave; poly < count; poly += (long long) gridDim.x * 4) {
        c64 a[8], b[8], prod[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int j = (br3(r) << 6) | lane;
            a[r] = {(double) ipolys[poly * kN + j], (double) ipolys[poly * kN + j + kNh]};
            b[r] = {tpolys[poly * kN + j] * (1.0 / 4294967296.0), tpolys[poly * kN + j + kNh] * (1.0 / 4294967296.0)};
        }
        wave_fwd(a, lane, scratch, twu, twl);
        wave_fwd(b, lane, scratch, twu, twl);
#pragma unroll
        for (int r = 0; r < 8; r++) {  // LagrangeHalfCPolynomialMul, polynomials.cpp:301-310
            const double ac = a[r].re * b[r].re, bd = a[r].im * b[r].im, ad = a[r].re * b[r].im, bc = a[r].im * b[r].re;
            prod[r] = {ac - bd, ad + bc};
        }
        c64 v[16];
        int32_t res[16];
        spectrum_to_inv_input(v, prod, lane, scratch);
        wave_inv<false>(res, v, lane, scratch, twu, twl);
#pragma unroll
        for (int rho = 0; rho < 16; rho++) out[poly * kN + inv_yC(lane, rho)] = res[rho];
        wave_sync();
    }
}

// ---- the same stages in the FAST arithmetic mode (fast_core.h); spectra cross the ABI in the reference order ----------
constexpr int kFftLdsFast = brf::kFastLaneEntries * (int) sizeof(c64) + 4 * kNh * (int) sizeof(c64);
__global__ __launch_bounds__(256) void k_poly_fwd_fast(const int32_t *__restrict__ polys, c64 *__restrict__ spectra,
                                                       const c64 *__restrict__ tw, PhaseAConsts ka, long long count, int torus) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    c64 *tw_lds = reinterpret_cast<c64 *>(lds_raw);
    load_tw_to_lds(tw_lds, tw, brf::kFastLaneEntries);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    c64 *scratch = reinterpret_cast<c64 *>(lds_raw + brf::kFastLaneEntries * sizeof(c64)) + wave * kNh;
    const long long poly = (long long) blockIdx.x * 4 + wave;
    if (poly >= count) return;
    const int32_t *p = polys + poly * kN;
    const double s = torus ? 1.0 / 4294967296.0 : 1.0;
    c64 v[8];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const int j = (br3(r) << 6) | lane;
        v[r] = {p[j] * s, p[j + kNh] * s};
    }
    wave_fwd_fast(v, lane, scratch, ka, TwLds{tw_lds + lane});
#pragma unroll
    for (int r = 0; r < 8; r++) {  // Z_k -> reference slots: C[2k] = Z_k (k < 256), C[1023 - 2k] = conj(Z_k) (k >= 256)
        const int k = (r << 6) | lane;
        spectra[poly * kNh + (k < kNh / 2 ? 2 * k : kN - 1 - 2 * k)] = k < kNh / 2 ? v[r] : c64{v[r].re, -v[r].im};
    }
}
__device__ __forceinline__ c64 fast_from_ref(const c64 *C, int k) {
    if (k < kNh / 2) return C[2 * k];
    const c64 t = C[kN - 1 - 2 * k];
    return {t.re, -t.im};
}
__global__ __launch_bounds__(256) void k_poly_inv_fast(const c64 *__restrict__ spectra, int32_t *__restrict__ polys,
                                                       const c64 *__restrict__ tw, PhaseAConsts ka, long long count) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    c64 *tw_lds = reinterpret_cast<c64 *>(lds_raw);
    load_tw_to_lds(tw_lds, tw, brf::kFastLaneEntries);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    c64 *scratch = reinterpret_cast<c64 *>(lds_raw + brf::kFastLaneEntries * sizeof(c64)) + wave * kNh;
    const long long poly = (long long) blockIdx.x * 4 + wave;
    if (poly >= count) return;
    c64 sp[8];
#pragma unroll
    for (int r = 0; r < 8; r++) sp[r] = fast_from_ref(spectra + poly * kNh, (r << 6) | lane);
    int32_t lo[8], hi[8];
    wave_inv_fast<false>(lo, hi, sp, lane, scratch, ka, TwLds{tw_lds + lane});
#pragma unroll
    for (int r = 0; r < 8; r++) {
        polys[poly * kN + ((r << 6) | lane)] = lo[r];
        polys[poly * kN + ((r << 6) | lane) + kNh] = hi[r];
    }
}
__global__ __launch_bounds__(256) void k_poly_mul_fast(const int32_t *__restrict__ ipolys, const int32_t *__restrict__ tpolys,
                                                       int32_t *__restrict__ out, const c64 *__restrict__ tw,
                                                       PhaseAConsts ka, long long count) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    c64 *tw_lds = reinterpret_cast<c64 *>(lds_raw);
    load_tw_to_lds(tw_lds, tw, brf::kFastLaneEntries);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    c64 *scratch = reinterpret_cast<c64 *>(lds_raw + brf::kFastLaneEntries * sizeof(c64)) + wave * kNh;
    const TwLds twl{tw_lds + lane};
#pragma unroll 1
    for (long long poly = (long long) blockIdx.x * 4 + wave; poly < count; poly += (long long) gridDim.x * 4) {  // as k_poly_mul
        c64 a[8], b[8];   // (requesting the next product's operands ahead of the transforms measured no difference)
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int j = (br3(r) << 6) | lane;
            a[r] = {(double) ipolys[poly * kN + j], (double) ipolys[poly * kN + j + kNh]};
            b[r] = {tpolys[poly * kN + j] * (1.0 / 4294967296.0), tpolys[poly * kN + j + kNh] * (1.0 / 4294967296.0)};
        }
        wave_fwd_fast(a, lane, scratch, ka, twl);
        wave_sync();
        wave_fwd_fast(b, lane, scratch, ka, twl);
        wave_sync();
#pragma unroll
        for (int r = 0; r < 8; r++) a[r] = brf::cmul(a[r], b[r]);
        int32_t lo[8], hi[8];
        wave_inv_fast<false>(lo, hi, a, lane, scratch, ka, twl);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            out[poly * kN + ((r << 6) | lane)] = lo[r];
            out[poly * kN + ((r << 6) | lane) + kNh] = hi[r];
        }
        wave_sync();
    }
}
// N = 2048 product in the FAST mode, one wave per product: with Y = X^2, a = a_e(Y) + X a_o(Y) in Z[Y]/(Y^1024 + 1),
//   ab = (a_e b_e + Y a_o b_o)(Y) + X (a_e b_o + a_o b_e)(Y)
// and the sums are taken in the evaluation domain: FOUR forward and TWO inverse transforms instead of the twelve of
// four separate products, no operand lists and no second pass over the results.  Y evaluates to omega^(4k+1), which
// is the twiddle of the last forward level (slots FS_FC+2, +3 and their quarter turns).  Even / odd coefficients
// sit side by side in memory, so all loads and stores are 8 bytes per lane.
__global__ __launch_bounds__(256) void k_poly_mul2048_fast(const int32_t *__restrict__ ipolys, const int32_t *__restrict__ tpolys,
                                                           int32_t *__restrict__ out, const c64 *__restrict__ tw,
                                                           PhaseAConsts ka, long long count) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    c64 *tw_lds = reinterpret_cast<c64 *>(lds_raw);
    load_tw_to_lds(tw_lds, tw, brf::kFastLaneEntries);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    c64 *scratch = reinterpret_cast<c64 *>(lds_raw + brf::kFastLaneEntries * sizeof(c64)) + wave * kNh;
    const TwLds twl{tw_lds + lane};
    typedef int32_t i32x2 __attribute__((ext_vector_type(2)));
    const c64 ta = twl(brf::FS_FC + 2), tb = twl(brf::FS_FC + 3);
#pragma unroll 1
    for (long long poly = (long long) blockIdx.x * 4 + wave; poly < count; poly += (long long) gridDim.x * 4) {
        const i32x2 *pa = reinterpret_cast<const i32x2 *>(ipolys + poly * 2 * kN);
        const i32x2 *pb = reinterpret_cast<const i32x2 *>(tpolys + poly * 2 * kN);
        c64 ae[8], ao[8], be[8], bo[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const int j = (br3(r) << 6) | lane;
            const i32x2 a0 = pa[j], a1 = pa[j + kNh], b0 = pb[j], b1 = pb[j + kNh];
            ae[r] = {(double) a0.x, (double) a1.x};
            ao[r] = {(double) a0.y, (double) a1.y};
            be[r] = {b0.x * (1.0 / 4294967296.0), b1.x * (1.0 / 4294967296.0)};
            bo[r] = {b0.y * (1.0 / 4294967296.0), b1.y * (1.0 / 4294967296.0)};
        }
        wave_fwd_fast(ae, lane, scratch, ka, twl);
        wave_sync();
        wave_fwd_fast(ao, lane, scratch, ka, twl);
        wave_sync();
        wave_fwd_fast(be, lane, scratch, ka, twl);
        wave_sync();
        wave_fwd_fast(bo, lane, scratch, ka, twl);
        wave_sync();
        c64 se[8], so[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {  // register r holds Z_k, k = r<<6|lane: Y -> ta, tb, i ta, i tb, -ta, -tb, -i ta, -i tb
            const c64 t0 = (r & 1) ? tb : ta;
            const c64 t1 = (r & 2) ? c64{-t0.im, t0.re} : t0;
            const c64 t = (r & 4) ? c64{-t1.re, -t1.im} : t1;
            se[r] = brf::cmul(ae[r], be[r]);
            brf::cmac(se[r], brf::cmul(ao[r], bo[r]), t);
            so[r] = brf::cmul(ae[r], bo[r]);
            brf::cmac(so[r], ao[r], be[r]);
        }
        int32_t le[8], he[8], lo[8], ho[8];
        wave_inv_fast<false>(le, he, se, lane, scratch, ka, twl);
        wave_sync();
        wave_inv_fast<false>(lo, ho, so, lane, scratch, ka, twl);
        i32x2 *po = reinterpret_cast<i32x2 *>(out + poly * 2 * kN);
#pragma unroll
        for (int r = 0; r < 8; r++) {
            po[(r << 6) | lane] = i32x2{le[r], lo[r]};
            po[((r << 6) | lane) + kNh] = i32x2{he[r], ho[r]};
        }
        wave_sync();
    }
}
// bootstrapping key of the FAST mode from the exact mode's (physical slot x holds reference bin igray(x)): the same
// doubles, re-indexed to Z_k order [poly][k] and conjugated where the fold asks for it - no arithmetic, so the key is
// the reference's bkFFT bit for bit in both modes
// rows of a ciphertext pool <-> a contiguous staging buffer (tfhe_hip_group_gate_level: what a member wrote travels to the
// other members' replicas as one peer copy)
__global__ void k_rows_gather(int32_t *__restrict__ stage, const int32_t *__restrict__ pool, const int32_t *__restrict__ rows,
                              long long total) {
    const long long idx = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    stage[idx] = pool[(long long) rows[idx / kCt] * kCt + idx % kCt];
}
__global__ void k_rows_scatter(int32_t *__restrict__ pool, const int32_t *__restrict__ stage, const int32_t *__restrict__ rows,
                               long long total) {
    const long long idx = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    pool[(long long) rows[idx / kCt] * kCt + idx % kCt] = stage[idx];
}
// 64-bit content fingerprint of a device buffer (tfhe_hip_keys_fingerprint): sum over the 8-byte words of mix(word, index).
// The sum is commutative, so the atomic order does not matter: equal contents give equal fingerprints on every device.
__global__ __launch_bounds__(256) void k_fingerprint(const unsigned long long *__restrict__ p, long long words,
                                                     unsigned long long *__restrict__ out) {
    unsigned long long h = 0;
    for (long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (long long) gridDim.x * blockDim.x) {
        unsigned long long z = p[i] + 0x9e3779b97f4a7c15ULL * (unsigned long long) (i + 1);  // splitmix64 finaliser
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        h += z ^ (z >> 31);
    }
    for (int off = 32; off > 0; off >>= 1) h += __shfl_down(h, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out, h);
}

__global__ void k_bk_to_fast(c64 *__restrict__ dst, const c64 *__restrict__ src, long long total) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long long poly = i >> 9;
    const int k = (int) (i & 511);
    const int bin = k < kNh / 2 ? 2 * k : kN - 1 - 2 * k;
    const c64 t = src[poly * kNh + gray(bin)];
    dst[i] = k < kNh / 2 ? t : c64{t.re, -t.im};
}

// ---------------------------------------------------------------------------------------------------------------
// N = 2048 negacyclic product (BASELINE configs[4] microbench) from four N = 1024 products.  With Y = X^2,
// a = a_e(Y) + X a_o(Y) and b alike:  a b = (a_e b_e + Y a_o b_o)(Y) + X (a_e b_o + a_o b_e)(Y)  mod X^2048 + 1, and
// each of the four is a product mod Y^1024 + 1, i.e. the reference's torusPolynomialMultFFT (polynomials.cpp:241) as
// k_poly_mul computes it; multiplication by Y is a negacyclic shift, the additions are exact mod 2^32.  The reference
// itself has no transform for N = 2048 (tfhe_core.h:14 fixes N = 1024); its naive product
// torusPolynomialMultNaive_aux (multiplication.cpp:31-43) is the oracle, reached within 2 LSB.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_split2048(const int32_t *__restrict__ a, const int32_t *__restrict__ b,
                                                   int32_t *__restrict__ I, int32_t *__restrict__ T, long long count) {
    const long long idx = (long long) blockIdx.x * 256 + threadIdx.x;  // one thread per (product, j < 1024)
    if (idx >= count * kN) return;
    const long long g = idx >> 10;
    const int j = (int) (idx & (kN - 1));
    const int32_t ae = a[g * 2 * kN + 2 * j], ao = a[g * 2 * kN + 2 * j + 1];
    const int32_t be = b[g * 2 * kN + 2 * j], bo = b[g * 2 * kN + 2 * j + 1];
    int32_t *Ig = I + g * 4 * kN + j, *Tg = T + g * 4 * kN + j;
    Ig[0] = ae; Tg[0] = be;                  // P0 = a_e b_e
    Ig[kN] = ao; Tg[kN] = bo;                // P1 = a_o b_o
    Ig[2 * kN] = ae; Tg[2 * kN] = bo;        // P2 = a_e b_o
    Ig[3 * kN] = ao; Tg[3 * kN] = be;        // P3 = a_o b_e
}
__global__ __launch_bounds__(256) void k_combine2048(const int32_t *__restrict__ P, int32_t *__restrict__ out,
                                                     long long count) {
    const long long idx = (long long) blockIdx.x * 256 + threadIdx.x;
    if (idx >= count * kN) return;
    const long long g = idx >> 10;
    const int j = (int) (idx & (kN - 1));
    const uint32_t *Pg = reinterpret_cast<const uint32_t *>(P) + g * 4 * kN;
    const uint32_t shifted = j ? Pg[kN + j - 1] : 0u - Pg[kN + kN - 1];  // (Y * P1)[j]
    out[g * 2 * kN + 2 * j] = (int32_t) (Pg[j] + shifted);
    out[g * 2 * kN + 2 * j + 1] = (int32_t) (Pg[2 * kN + j] + Pg[3 * kN + j]);
}

// ---------------------------------------------------------------------------------------------------------------
// leveled operations on caller-supplied operands: tGswExternMulToTLwe (tgsw-functions.cpp:157-170) and
// tLweAddMulRTo (tlwe-functions.cpp:191-197).  The reference computes them row by row through torusPolynomialAddMulRFFT
// (polynomials.cpp:249-259): every (row, column) product is transformed back and truncated on its own, the Torus32
// results are added mod 2^32.  Here: operand lists for k_poly_mul (= torusPolynomialMultFFT) are laid out, all
// products run as one batch, and the rows are summed in integers - the same values in any order.
// ---------------------------------------------------------------------------------------------------------------
// products (g, p, c): I = digit (p % l) of column (p / l) of acc[g] (tGswTLweDecompH, tgsw-functions.cpp:264-270),
// gadget decomposition as a stage of its own (tGswTorus32PolynomialDecompH, tgsw-functions.cpp:297-331): the same
// decomp_prepare / decomp_digit that the CMux applies to the rotated differences.  out[poly][p][j]
__global__ void k_decomp(int32_t *__restrict__ out, const int32_t *__restrict__ in, long long total) {
    const long long i = (long long) blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long long poly = i / kN;
    const int j = (int) (i - poly * kN);
    const uint32_t w = decomp_prepare(in[i]);
#pragma unroll
    for (int p = 0; p < kL; p++) out[(poly * kL + p) * kN + j] = decomp_digit(w, p);
}

// T = tgsw[index[g]][p][c]
__global__ __launch_bounds__(256) void k_tgsw_prepare(const int32_t *__restrict__ acc, const int32_t *__restrict__ tgsw,
                                                      const int32_t *__restrict__ index, int32_t *__restrict__ I,
                                                      int32_t *__restrict__ T, long long count) {
    const long long idx = (long long) blockIdx.x * 256 + threadIdx.x;  // one thread per (g, p, j)
    if (idx >= count * kKpl * kN) return;
    const int j = (int) (idx & (kN - 1));
    const long long gp = idx >> 10;
    const int p = (int) (gp % kKpl);
    const long long g = gp / kKpl;
    const int32_t d = decomp_digit(decomp_prepare(acc[(g * 2 + p / kL) * kN + j]), p % kL);
    const int32_t *row = tgsw + ((size_t) index[g] * kKpl + p) * 2 * kN;
#pragma unroll
    for (int c = 0; c < 2; c++) {
        I[((g * kKpl + p) * 2 + c) * kN + j] = d;
        T[((g * kKpl + p) * 2 + c) * kN + j] = row[c * kN + j];
    }
}
// products (g, c): I = p[g], T = sample[g][c]
__global__ __launch_bounds__(256) void k_addmul_prepare(const int32_t *__restrict__ poly, int32_t *__restrict__ I, long long count) {
    const long long idx = (long long) blockIdx.x * 256 + threadIdx.x;  // one thread per (g, j)
    if (idx >= count * kN) return;
    const long long g = idx >> 10;
    const int j = (int) (idx & (kN - 1));
    I[(g * 2) * kN + j] = I[(g * 2 + 1) * kN + j] = poly[idx];
}
// out[g][c][j] = (ADD ? out : 0) + sum_{p < rows} P[g][p][c][j]
template <bool ADD>
__global__ __launch_bounds__(256) void k_rows_sum(int32_t *__restrict__ out, const int32_t *__restrict__ P, int rows, long long count) {
    const long long idx = (long long) blockIdx.x * 256 + threadIdx.x;  // one thread per (g, c, j)
    if (idx >= count * 2 * kN) return;
    const int j = (int) (idx & (kN - 1)), c = (int) ((idx >> 10) & 1);
    const long long g = idx >> 11;
    uint32_t v = ADD ? (uint32_t) out[idx] : 0u;
    for (int p = 0; p < rows; p++) v += (uint32_t) P[((g * rows + p) * 2 + c) * kN + j];
    out[idx] = (int32_t) v;
}

// ---------------------------------------------------------------------------------------------------------------
// client-side bulk operations on the GPU (the part that parallelises): bootsSymEncrypt
// (tfhe_gate_bootstrapping.cpp:154-159 = lweSymEncrypt, lwe-functions.cpp:44-56) and lwePhase (:63-72).
// The reference draws from std::default_random_engine + normal_distribution, whose streams are not portable; these
// kernels reproduce, bit for bit, the portable generator of tfhe_amd/csrc/client_keys.c (one xoshiro256** stream per
// sample; noise = the reference's law, gaussian32 of numeric-functions.cpp:22-34, by the integer table inversion of
// gauss_cdt.h, standard deviation 2^-15), so a ciphertext is a function of (seed, index, bit, key).
// ---------------------------------------------------------------------------------------------------------------
#if TFHE_HIP_LAMBDA == 128
constexpr int64_t kKsNoiseM = (int64_t) 1 << 25, kBkNoiseM = (int64_t) 1 << 15;  // 2^-15, 2^-25 (x 2^40)
#else
constexpr int64_t kKsNoiseM = 26828165, kBkNoiseM = 7894;  // 2.44e-5, 7.18e-9 (tfhe_gate_bootstrapping.cpp:35-36)
#endif
struct ClientRng {
    uint64_t s[4];
};
__device__ __forceinline__ uint64_t crng_splitmix(uint64_t &x) {
    uint64_t z = (x += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__device__ __forceinline__ void crng_seed(ClientRng &r, uint64_t seed, uint64_t stream) {
    uint64_t x = seed ^ (stream * 0xD1342543DE82EF95ULL + 0x632BE59BD9B4E019ULL);
    for (int i = 0; i < 4; i++) r.s[i] = crng_splitmix(x);
}
__device__ __forceinline__ uint64_t crng_rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
__device__ __forceinline__ uint64_t crng_next(ClientRng &r) {  // xoshiro256**
    const uint64_t res = crng_rotl(r.s[1] * 5, 7) * 9, t = r.s[1] << 17;
    r.s[2] ^= r.s[0]; r.s[3] ^= r.s[1]; r.s[1] ^= r.s[2]; r.s[0] ^= r.s[3]; r.s[2] ^= t; r.s[3] = crng_rotl(r.s[3], 45);
    return res;
}
// gaussian32 (numeric-functions.cpp:22-34) by integer table inversion, as rng_gauss of client_keys.c: two words of the stream
__device__ __forceinline__ int32_t crng_gauss(ClientRng &r, int64_t m, const uint64_t *__restrict__ cdt) {
    const uint64_t w_mag = crng_next(r), w_pos = crng_next(r);
    return gauss_cdt_value(cdt, w_mag, w_pos, m);
}
// one thread per sample (the generator of a sample is sequential); the 630 mask words of 64 samples pass through LDS
// so that the global stores are coalesced rows
__global__ __launch_bounds__(64) void k_client_encrypt(uint64_t seed, const int32_t *__restrict__ key,
                                                       const uint8_t *__restrict__ bits, long long first, long long count,
                                                       int32_t *__restrict__ out, const uint64_t *__restrict__ cdt) {
    __shared__ int32_t tile[64][65];
    const long long c0 = (long long) blockIdx.x * 64, c = c0 + threadIdx.x;
    ClientRng r;
    crng_seed(r, seed, 0x100000000ULL + (uint64_t) (first + c));
    const bool live = c < count;
    uint32_t b = live && bits[c] ? (1u << 29) : (0u - (1u << 29));
    b += (uint32_t) crng_gauss(r, kKsNoiseM, cdt);
    for (int q0 = 0; q0 < kLwe; q0 += 64) {
        const int nq = kLwe - q0 < 64 ? kLwe - q0 : 64;
        for (int q = 0; q < nq; q++) {
            const int32_t a = (int32_t) (uint32_t) (crng_next(r) >> 32);
            tile[threadIdx.x][q] = a;
            b += (uint32_t) a * (uint32_t) key[q0 + q];
        }
        __syncthreads();
        for (int row = 0; row < 64 && c0 + row < count; row++)
            if ((int) threadIdx.x < nq) out[(c0 + row) * kCt + q0 + threadIdx.x] = tile[row][threadIdx.x];
        __syncthreads();
    }
    if (live) out[c * kCt + kLwe] = (int32_t) b;
}
// lwePhase: b - <a, s>; one wave per sample
__global__ __launch_bounds__(256) void k_client_phase(const int32_t *__restrict__ key, const int32_t *__restrict__ ct,
                                                      long long count, int32_t *__restrict__ phases) {
    const long long c = (long long) blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (c >= count) return;
    uint32_t acc = 0;
    for (int q = lane; q < kLwe; q += 64) acc += (uint32_t) ct[c * kCt + q] * (uint32_t) key[q];
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) phases[c] = (int32_t) ((uint32_t) ct[c * kCt + kLwe] - acc);
}

// bootstrapping-key generation (tfhe_createLweBootstrappingKey, lwe-bootstrapping-functions.cpp:185-217: bk[i] =
// TGSW(s_i) under the TLWE key, tGswSymEncryptInt = 6 TLWE zero-encryptions + s_i * H, tgsw-functions.cpp:100-133),
// bit-identical to bk_worker of client_keys.c.  Stage 1: one thread per key bit i draws its stream (noise then mask
// of each row, sequential by construction).  Stage 2: one workgroup per row adds mask * tlwe_key in Z[X]/(X^N+1)
// (torusPolynomialAddMulR at tlwe-functions.cpp:35) - exact mod 2^32 in any summation order - and the gadget term.
__global__ __launch_bounds__(64) void k_client_bk_draw(uint64_t seed, int32_t *__restrict__ bk_coef, const uint64_t *__restrict__ cdt) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= kLwe) return;
    ClientRng r;
    crng_seed(r, seed, 0x10000ULL + (uint64_t) i);
    for (int p = 0; p < kKpl; p++) {
        int32_t *a = bk_coef + (((size_t) i * kKpl + p) * 2 + 0) * kN, *b = a + kN;
        for (int q = 0; q < kN; q++) b[q] = crng_gauss(r, kBkNoiseM, cdt);
        for (int q = 0; q < kN; q++) a[q] = (int32_t) (uint32_t) (crng_next(r) >> 32);
    }
}
__global__ __launch_bounds__(256) void k_client_bk_mask(const int32_t *__restrict__ lwe_key, const int32_t *__restrict__ tlwe_key,
                                                        int32_t *__restrict__ bk_coef) {
    __shared__ uint32_t sa[kN];
    __shared__ uint8_t sk[kN];
    const int row = blockIdx.x, i = row / kKpl, p = row % kKpl, x = threadIdx.x;
    int32_t *a = bk_coef + ((size_t) row * 2 + 0) * kN, *b = a + kN;
    for (int q = x; q < kN; q += 256) {
        sa[q] = (uint32_t) a[q];
        sk[q] = (uint8_t) (tlwe_key[q] != 0);
    }
    __syncthreads();
    uint32_t acc[4];
    for (int u = 0; u < 4; u++) acc[u] = (uint32_t) b[x + 256 * u];
    for (int j = 0; j < kN; j++) {
        if (!sk[j]) continue;  // uniform over the workgroup
        for (int u = 0; u < 4; u++) {
            const int d = x + 256 * u - j;
            if (d >= 0) acc[u] += sa[d];
            else acc[u] -= sa[d + kN];
        }
    }
    const uint32_t h = (uint32_t) lwe_key[i] * (1u << (32 - (p % kL + 1) * kBgbit));
    if (x == 0) {
        if (p / kL == 0) a[0] = (int32_t) (sa[0] + h);
        else acc[0] += h;
    }
    for (int u = 0; u < 4; u++) b[x + 256 * u] = (int32_t) acc[u];
}

// keyswitch-key generation (lweCreateKeySwitchKey, lwe-keyswitch-functions.cpp:163-215: ks[i][j][h] = LWE_n encryption
// of h * s'_i / base^(j+1), h = 0 the trivial zero sample), bit-identical to tfhe_client_keygen, which draws all
// 24,576 samples from ONE xoshiro256** stream, n + 2 draws each.  The generator's state transition is linear over GF(2),
// so thread s reaches its starting state by applying the precomputed 256 x 256 bit matrices T^(636 * 2^k) for the
// set bits k of s (jump[k][row][word]; lane-uniform reads), then draws its sample as the CPU does.
constexpr int kKsDraws = 2 + kLwe;  // noise (two words: gauss_cdt.h) + mask
constexpr int kKsJumpBits = 15;     // N * t * (base - 1) = 24,576 samples < 2^15
__global__ __launch_bounds__(64) void k_client_ks_draw(ClientRng start, const uint64_t *__restrict__ jump,
                                                       const int32_t *__restrict__ lwe_key,
                                                       const int32_t *__restrict__ tlwe_key, int32_t *__restrict__ ks,
                                                       const uint64_t *__restrict__ cdt) {
    __shared__ int32_t tile[64][65];
    const int s0 = blockIdx.x * 64, s = s0 + threadIdx.x;  // sample index = (i * t + j) * 3 + (h - 1); grid covers all
    ClientRng r = start;
    for (int k = 0; k < kKsJumpBits; k++) {
        if (!((s >> k) & 1)) continue;
        const uint64_t *m = jump + (size_t) k * 256 * 4;
        uint64_t out[4] = {0, 0, 0, 0};
        for (int row = 0; row < 256; row++) {
            const uint64_t x = (m[row * 4] & r.s[0]) ^ (m[row * 4 + 1] & r.s[1]) ^ (m[row * 4 + 2] & r.s[2]) ^
                               (m[row * 4 + 3] & r.s[3]);
            out[row >> 6] |= (uint64_t) (__popcll(x) & 1) << (row & 63);
        }
        for (int w = 0; w < 4; w++) r.s[w] = out[w];
    }
    const int ij = s / 3, h = s % 3 + 1, i = ij / kKsT, j = ij % kKsT;
    uint32_t b = (uint32_t) (tlwe_key[i] * h) << (32 - (j + 1) * kKsBasebit);
    b += (uint32_t) crng_gauss(r, kKsNoiseM, cdt);
    // rows of the host layout [N][t][4][n+1]; the mask words of 64 samples pass through LDS for coalesced stores
    for (int q0 = 0; q0 < kLwe; q0 += 64) {
        const int nq = kLwe - q0 < 64 ? kLwe - q0 : 64;
        for (int q = 0; q < nq; q++) {
            const int32_t a = (int32_t) (uint32_t) (crng_next(r) >> 32);
            tile[threadIdx.x][q] = a;
            b += (uint32_t) a * (uint32_t) lwe_key[q0 + q];
        }
        __syncthreads();
        for (int t = 0; t < 64; t++) {
            const int st = s0 + t;
            const size_t rowoff = ((size_t) (st / 3) * 4 + (st % 3 + 1)) * kCt;
            if ((int) threadIdx.x < nq) ks[rowoff + q0 + threadIdx.x] = tile[t][threadIdx.x];
        }
        __syncthreads();
    }
    ks[((size_t) ij * 4 + h) * kCt + kLwe] = (int32_t) b;
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
thread_local std::string g_err;
int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
#define HIPCHK(x)                                                                                          \
    do {                                                                                                   \
        hipError_t e_ = (x);                                                                               \
        if (e_ != hipSuccess)                                                                              \
            return fail(TFHE_HIP_ERR_HIP, std::string(#x) + ": " + hipGetErrorString(e_));                 \
    } while (0)

// Twiddle tables.  The reference hard-codes them (src/include/tfhe_core.h:26,93); they equal double-precision
// libm cos/sin of the angle folded into the first octant.  Regenerated here and checked against a fingerprint of
// the reference values, so that a host libm that rounds differently is detected instead of silently changing bits.
double oct_cos(int j, int n) {
    const double w = 2.0 * M_PI;
    if (4 * j == n) return 0.0;
    if (8 * j <= n) return cos(w * j / n);
    if (4 * j <= n) return sin(w * (n / 4 - j) / n);
    if (8 * j < 3 * n) return -sin(w * (j - n / 4) / n);
    return -cos(w * (n / 2 - j) / n);
}
double oct_sin(int j, int n) {
    const double w = 2.0 * M_PI;
    if (8 * j < n) return sin(w * j / n);
    if (4 * j <= n) return cos(w * (n / 4 - j) / n);
    if (8 * j <= 3 * n) return cos(w * (j - n / 4) / n);
    return sin(w * (n / 2 - j) / n);
}
constexpr uint64_t kTwFingerprint = 0x749e3b06997a84d7ULL;
bool build_twiddles(std::vector<c64> &tw) {
    tw.resize(kTwEntries);
    uint64_t h = 1469598103934665603ULL;
    auto mix = [&](double v) {
        uint64_t u;
        if (v == 0.0) v = 0.0;
        memcpy(&u, &v, 8);
        for (int b = 0; b < 8; b++) {
            h ^= (u >> (8 * b)) & 0xff;
            h *= 1099511628211ULL;
        }
    };
    for (int size = 8; size <= 2048; size *= 2) {
        for (int j = 0; j < size / 2; j++) tw[tw_idx(size, j)] = {oct_cos(j, size), oct_sin(j, size)};
        for (int j = 0; j < size / 2; j++) mix(tw[tw_idx(size, j)].re);
        for (int j = 0; j < size / 2; j++) mix(tw[tw_idx(size, j)].im);
    }
    return h == kTwFingerprint;
}

// ---- static scratch of the entry point that allocates nothing (tfhe_hip_gate_batch_dev_noalloc) --------------------
// One copy per GPU the library runs on, held until the process ends.  Regions (256-byte aligned): extracted samples of
// the first and (MUX) second blind rotation, the keyswitch's transposed input, its partial sums.  Every call on a GPU
// shares the array; nothing in it has to start at zero (each region is fully written before it is read: the blind
// rotation writes every extracted row it later reads, k_ks_transpose every abarT column incl. the padding ciphertexts,
// k_keyswitch_mfma every partial sum that k_ks_reduce adds), but calls that overlap on one GPU race.
constexpr size_t na_align(size_t b) { return (b + 255) / 256 * 256; }
constexpr size_t kNaMaxGates = TFHE_HIP_NOALLOC_MAX_GATES;
constexpr size_t kNaMaxBp = (kNaMaxGates + 32 * kKsMT - 1) / (32 * kKsMT) * (32 * kKsMT);
constexpr size_t kNaExtBytes = na_align(kNaMaxGates * (kN + 1) * sizeof(int32_t));
constexpr size_t kNaAbartBytes = na_align((size_t) kN * kNaMaxBp * sizeof(uint32_t));
constexpr size_t kNaPartBytes = (size_t) 8 << 20;
constexpr size_t kNaExt1Off = 0, kNaExt2Off = kNaExtBytes, kNaAbartOff = 2 * kNaExtBytes, kNaPartOff = kNaAbartOff + kNaAbartBytes;
constexpr size_t kNaTotalBytes = kNaPartOff + kNaPartBytes;
static __device__ __align__(256) unsigned char g_noalloc_scratch[kNaTotalBytes];
// g_noalloc_scratch is a host-side stand-in here, not a device address: host code only ever uses the per-device pointer
constexpr int kNaMaxDevices = 64;
unsigned char *g_na_ptr[kNaMaxDevices];
std::mutex g_na_mu;
// with `device` current: resolve this device's copy of the array once
hipError_t noalloc_scratch_init(int device) {
    if (device < 0 || device >= kNaMaxDevices) return hipSuccess;  // beyond the table: the no-alloc entry point reports it
    std::lock_guard<std::mutex> lock(g_na_mu);
    if (g_na_ptr[device]) return hipSuccess;
    void *p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(g_noalloc_scratch));
    if (e == hipSuccess) g_na_ptr[device] = (unsigned char *) p;
    return e;
}
unsigned char *noalloc_scratch_ptr(int device) {
    if (device < 0 || device >= kNaMaxDevices) return nullptr;
    std::lock_guard<std::mutex> lock(g_na_mu);
    return g_na_ptr[device];
}

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap) return hipSuccess;
        if (p) (void) hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() {
        if (p) (void) hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

}  // namespace

struct tfhe_hip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;      // host<->device copies of chunked host-buffer calls
    std::vector<hipEvent_t> chunk_ev;       // per chunk: inputs landed, kernels done
    c64 *tw = nullptr;
    PhaseAConsts ka{};
    int arith = TFHE_HIP_ARITH_EXACT;  // tfhe_hip_set_arithmetic
    c64 *ntab[3] = {nullptr, nullptr, nullptr};  // transform tables for N = 512 / 1024 / 2048 (nk_ops.h), built on first use
    uint64_t *cdt = nullptr; // noise table of the client-side kernels (gauss_cdt.h), uploaded on first use
    c64 *twf = nullptr;      // per-lane twiddle image of the FAST mode (fast_core.h)
    c64 *bkf = nullptr;      // FAST mode: [n][6][2][512] in Z_k order, built from bk when the mode is selected
    c64 *bk = nullptr;       // [n][6][2][512] physical order
    int32_t *ks = nullptr;   // [N][t][3][632]   (VALU keyswitch, kept for the ragged tail / A-B checks)
    int8_t *kl = nullptr;    // [N][4][640][t][4] byte limbs (MFMA keyswitch)
    int ks_mfma = 1;
    bool have_keys = false;
    DevBuf in_a, in_b, in_c, ext1, ext2, outb, misc, abart, ks_part, jobs_br, jobs_ks, jobs_lin, fp, grp_stage, grp_in, grp_rows;
    std::vector<BrJob> h_br;
    std::vector<KsJob> h_ks;
    std::vector<int32_t> h_lin;
    std::vector<uint8_t> h_mark;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};  // br start/stop, ks start/stop
    bool ev_br = false, ev_ks = false;
    bool in_mul2048 = false;  // the N = 2048 product is running its four N = 1024 products
    long long fault_chunk = -1;  // test hook: fail the host pipeline at this chunk once (TFHE_HIP_TEST_FAIL_CHUNK)
    uint64_t key_print[3] = {0, 0, 0};  // group member 0: fingerprints of its key buffers, taken before replication
    int key_path = 0;  // how the keys got here: 0 = uploaded / adopted, 1 = peer copy with peer access, 2 = peer copy staged by the runtime, 3 = same device
    int br_waves = 0;  // 0 = by batch size; TFHE_HIP_BR_WAVES=1|4|8 forces a shape, 6 the one-workgroup-per-ciphertext kernel
};

namespace {

constexpr size_t kBkBytes = (size_t) kLwe * kKpl * 2 * kNh * sizeof(c64);
constexpr size_t kKsDevBytes = (size_t) kN * kKsT * 3 * kKsRow * sizeof(int32_t);
constexpr size_t kKsHostElems = (size_t) kN * kKsT * 4 * kCt;
constexpr int kFftLds = kTwLaneEntries * (int) sizeof(c64) + 4 * kNh * (int) sizeof(c64);
constexpr long long kPolyMulMaxBlocks = 256 * 2 * 4;  // product kernels: workgroups walk over the batch (two resident per CU, four rounds)

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, device): set once, not per launch
int set_max_lds_once(const void *kernel, int lds, int device) {
    static std::mutex mu;
    static std::vector<std::pair<const void *, int>> done;
    std::lock_guard<std::mutex> lock(mu);
    for (auto &d : done)
        if (d.first == kernel && d.second == device) return TFHE_HIP_OK;
    HIPCHK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    done.emplace_back(kernel, device);
    return TFHE_HIP_OK;
}

template <int MODE>
int launch_br(tfhe_hip_ctx *c, const BrArgs &A0, const c64 *custom_bk = nullptr) {
    if (A0.B <= 0) return TFHE_HIP_OK;
    BrArgs A = A0;
#ifdef TFHE_HIP_CLOCK_DIAG
    static unsigned long long *d_diag = nullptr;
    static long long diag_cap = 0;
    if (A.B > diag_cap) {
        if (d_diag) (void) hipFree(d_diag);
        HIPCHK(hipMalloc((void **) &d_diag, (size_t) A.B * 16));
        diag_cap = A.B;
    }
    HIPCHK(hipMemsetAsync(d_diag, 0, (size_t) A.B * 16, c->stream));
    A.diag = d_diag;
    struct DiagPrint {
        tfhe_hip_ctx *c; long long B; unsigned long long *d;
        ~DiagPrint() {
            std::vector<unsigned long long> h(2 * B);
            (void) hipStreamSynchronize(c->stream);
            (void) hipMemcpy(h.data(), d, (size_t) B * 16, hipMemcpyDeviceToHost);
            std::vector<double> ghz, cyc;
            for (long long i = 0; i < B; i++)
                if (h[2 * i + 1]) { ghz.push_back(0.1 * (double) h[2 * i] / (double) h[2 * i + 1]); cyc.push_back((double) h[2 * i]); }
            if (ghz.empty()) return;
            std::sort(ghz.begin(), ghz.end()); std::sort(cyc.begin(), cyc.end());
            fprintf(stderr, "[clock diag] B=%lld in-kernel clock median %.3f GHz (min %.3f max %.3f); loop cycles per wave median %.0f\n",
                    B, ghz[ghz.size() / 2], ghz.front(), ghz.back(), cyc[cyc.size() / 2]);
        }
    } diag_print{c, A.B, d_diag};
#endif
    // A ragged last round: the chip holds 2,048 ciphertexts of the throughput shape at a time, and a round takes the same
    // time however few of them it holds (37 gates beyond 16,384: + 9.4 ms).  A small remainder is launched on its own, in
    // the shape that suits its size (up to 512: the latency kernel, 2.2 - 4.4 ms; up to 1,024: four waves, 6.5 ms).
    constexpr long long kRound = 256 * 8;
    if (c->br_waves == 0 && MODE != MODE_EXTMUL && A0.B > kRound && A0.B % kRound != 0 && A0.B % kRound <= 1024) {
        BrArgs head = A0, tail = A0;
        head.B = A0.B - A0.B % kRound;
        tail.B = A0.B % kRound;
        if (MODE == MODE_GATE) {
            if (tail.jobs) {
                tail.jobs += head.B;
            } else {
                tail.a += head.B * kCt;
                if (tail.b) tail.b += head.B * kCt;
            }
            tail.out += head.B * kExt;
        } else {
            tail.acc_io += head.B * 2 * kN;
            tail.bara += head.B * kLwe;
        }
        if (int r = launch_br<MODE>(c, head, custom_bk)) return r;
        return launch_br<MODE>(c, tail, custom_bk);
    }
    // waves per workgroup: 8 (two per SIMD) is the throughput shape.  Batches that cannot fill the chip that way run
    // with fewer waves per CU, which finish sooner (measured per wave: 8 per CU 16.4 ms, 4 per CU 10.5 ms, alone 8.8 ms)
    int waves = c->br_waves;
    // (measured, default set, ms for 256 / 512 / 768 / 1024 / 1536 gates: latency kernel 3.3 / 6.1 / 9.0 / 12.0 / 18.0,
    //  4 waves per workgroup 9.5 / 9.6 / 9.7 / 9.8 / 19.6, 8 waves 13.9 / 14.0 / 14.1 / 14.2 / 14.2; tools/shape_bench.py)
    constexpr long long kCoopMax = kCoopWaves > 4 ? 768 : 512;  // rounds of 256 workgroups that still beat 4 waves
    // FAST mode (round 3, ms for 256 / 512 / 768 / 1024 gates, tools/lat_time.py: latency kernel 2.15 / 3.9 / 5.8 / 7.7, one
    // wave per workgroup 5.4 / 5.7 / 6.1 / 6.1, four waves 6.1 / 6.2 / 6.2 / 6.2, eight 8.7): up to three rounds of
    // workgroups with the default set (the 80-bit set's throughput shapes are faster from two rounds on)
    const long long coop_max = c->arith == TFHE_HIP_ARITH_FAST ? (kCoopWaves > 4 ? 768 : 512) : kCoopMax;
    // 769..1,024 gates: FAST with the default set runs one wave per workgroup (6.06 against 6.23 ms with four; EXACT at
    // one wave per workgroup keeps only 768 ciphertexts resident: 18.5 against 9.9 ms)
    const int mid_shape = c->arith == TFHE_HIP_ARITH_FAST && kCoopWaves > 4 ? 1 : 4;
    if (waves == 0)
        waves = MODE == MODE_EXTMUL ? (A.B <= 256 ? 1 : A.B <= 1024 ? 4 : 8)
                                    : (A.B <= coop_max ? kShapeCoop : A.B <= 1024 ? mid_shape : 8);
    if (waves == kShapeCoop && MODE != MODE_EXTMUL && c->arith == TFHE_HIP_ARITH_FAST) {  // latency shape, FAST arithmetic
        A.tw = c->twf;
        A.bk = custom_bk ? custom_bk : c->bkf;
        if (int r = set_max_lds_once((const void *) k_blind_rotate_coop_fast<MODE == MODE_EXTMUL ? MODE_RAW : MODE>, kCoopFastLds,
                                     c->device))
            return r;
        hipLaunchKernelGGL((k_blind_rotate_coop_fast<MODE == MODE_EXTMUL ? MODE_RAW : MODE>), dim3((unsigned) A.B),
                           dim3(64 * kCoopFastWaves), kCoopFastLds, c->stream, A);
        HIPCHK(hipGetLastError());
        return TFHE_HIP_OK;
    }
    if (waves == kShapeCoop && MODE != MODE_EXTMUL) {  // one workgroup per ciphertext (latency shape)
        if (int r = set_max_lds_once((const void *) k_blind_rotate_coop<MODE == MODE_EXTMUL ? MODE_RAW : MODE>, kCoopLds, c->device))
            return r;
        hipLaunchKernelGGL((k_blind_rotate_coop<MODE == MODE_EXTMUL ? MODE_RAW : MODE>), dim3((unsigned) A.B),
                           dim3(64 * kCoopWaves), kCoopLds, c->stream, A);
        HIPCHK(hipGetLastError());
        return TFHE_HIP_OK;
    }
    if (waves == kShapeCoop) waves = 1;
    const bool fast = c->arith == TFHE_HIP_ARITH_FAST;
    if (fast) {  // the image and the key of the mode
        A.tw = c->twf;
        A.bk = c->bkf;
    }
    if (custom_bk) A.bk = custom_bk;  // caller-supplied TGSW operands, already in the layout of the mode
    A.bk_bytes = custom_bk ? 0x7ffffff0 : (int) kBkBytes;  // operand indices of a caller's array are validated on the host
    const size_t lds = (fast ? brf::kFastLaneEntries : kTwLaneEntries) * sizeof(c64) + (size_t) waves * kBrLdsPerWave;
    const unsigned grid = (unsigned) ((A.B + waves - 1) / waves);
    auto go = [&](auto kernel) -> int {
        if (int r = set_max_lds_once((const void *) kernel, (int) lds, c->device)) return r;
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(64 * waves), lds, c->stream, A);
        return TFHE_HIP_OK;
    };
    int r;
    if (fast) {
        if (waves == 8) r = go(k_blind_rotate<MODE, 8, true>);
        else if (waves == 4) r = go(k_blind_rotate<MODE, 4, true>);
        else r = go(k_blind_rotate<MODE, 1, true>);
    } else if (waves == 8) r = go(k_blind_rotate<MODE, 8>);
    else if (waves == 4) r = go(k_blind_rotate<MODE, 4>);
    else r = go(k_blind_rotate<MODE, 1>);
    if (r) return r;
    HIPCHK(hipGetLastError());
    return TFHE_HIP_OK;
}

// workspace of one keyswitch launch: the transposed input and, for small batches, the partial sums.  Null = the
// context's own growable buffers; the entry point that allocates nothing brings slices of its static array.
struct KsWork {
    uint32_t *abart;
    size_t abart_bytes;
    uint32_t *part;
    size_t part_bytes;
};
// split of the input coefficients over block groups for a batch of Bp (padded) ciphertexts: fill the 256 CUs
long long ks_xblocks(long long Bp) { return (Bp / (32 * kKsMT) + 3) / 4; }
// ... and in whole rounds: the chip runs 512 blocks at a time (two per CU), so e.g. the 1,280 blocks of 16,384 ciphertexts
// take three rounds for the work of 2.5; split two ways they are five rounds of half the length
int ks_split(long long Bp) {
    const long long nb = kKsYB * ks_xblocks(Bp);
    static const int forced = getenv("TFHE_HIP_KS_SPLIT") ? atoi(getenv("TFHE_HIP_KS_SPLIT")) : 0;  // diagnostic
    if (forced == 1 || forced == 2 || forced == 4 || forced == 8 || forced == 16) return forced;
    int S = 1;
    while (S < 16 && nb * S < 256) S *= 2;
    auto fill = [&](int s) { return (double) (nb * s) / (double) ((nb * s + 511) / 512 * 512); };
    while (S < 16 && nb * S < 8192 && fill(S) < 0.9 && fill(2 * S) > fill(S)) S *= 2;
    return S;
}
int launch_ks(tfhe_hip_ctx *c, int32_t *d_out, const int32_t *d_in, long long B, const KsJob *kj = nullptr,
              const KsWork *ws = nullptr) {
    if (B <= 0) return TFHE_HIP_OK;
    if (c->ks_mfma || kj || ws) {
        const long long Bp = (B + 32 * kKsMT - 1) / (32 * kKsMT) * (32 * kKsMT);
        const size_t abart_bytes = (size_t) kN * Bp * sizeof(uint32_t);
        uint32_t *abart;
        if (ws) {
            if (abart_bytes > ws->abart_bytes) return fail(TFHE_HIP_ERR_ARG, "keyswitch workspace too small");
            abart = ws->abart;
        } else {
            HIPCHK(c->abart.reserve(abart_bytes));
            abart = (uint32_t *) c->abart.p;
        }
        hipLaunchKernelGGL(k_ks_transpose, dim3((unsigned) (Bp / 32), kN / 32), dim3(256), 0, c->stream, abart, d_in, B, Bp, kj);
        const long long xblocks = ks_xblocks(Bp);
        int S = ks_split(Bp);
        if (ws)  // a fixed workspace: as many parts as fit (any split gives the same bits: integer sums)
            while (S > 1 && (size_t) S * Bp * kCt * sizeof(uint32_t) > ws->part_bytes) S /= 2;
        if (S == 1) {
            hipLaunchKernelGGL(k_keyswitch_mfma, dim3((unsigned) xblocks, kKsYB), dim3(256), 0, c->stream,
                               (const uint32_t *) abart, (const int8_t *) c->kl, d_in, d_out, B, Bp, kj,
                               (uint32_t *) nullptr);
        } else {
            uint32_t *part;
            if (ws) {
                part = ws->part;
            } else {
                HIPCHK(c->ks_part.reserve((size_t) S * Bp * kCt * sizeof(uint32_t)));
                part = (uint32_t *) c->ks_part.p;
            }
            hipLaunchKernelGGL(k_keyswitch_mfma, dim3((unsigned) xblocks, kKsYB, (unsigned) S), dim3(256), 0, c->stream,
                               (const uint32_t *) abart, (const int8_t *) c->kl, d_in, d_out, B, Bp, kj, part);
            hipLaunchKernelGGL(k_ks_reduce, dim3((unsigned) ((B * kCt + 255) / 256)), dim3(256), 0, c->stream,
                               (const uint32_t *) part, S, d_in, d_out, B, Bp, kj);
        }
    } else {
        dim3 grid((unsigned) ((B + kKsG - 1) / kKsG));
        hipLaunchKernelGGL(k_keyswitch, grid, dim3(192), 0, c->stream, d_in, c->ks, d_out, B);
    }
    HIPCHK(hipGetLastError());
    return TFHE_HIP_OK;
}

struct GateLin {
    int32_t cst, sa, sb;
};
// constants of boot-gates.cpp; modSwitchToTorus32(+-1,8) = +-2^29, (+-1,4) = +-2^30 (numeric-functions.cpp:71-76)
const int32_t kMU = 1 << 29;
bool gate_lin(int gate, GateLin &g) {
    switch (gate) {
        case TFHE_HIP_NAND: g = {kMU, -1, -1}; return true;
        case TFHE_HIP_OR: g = {kMU, 1, 1}; return true;
        case TFHE_HIP_AND: g = {-kMU, 1, 1}; return true;
        case TFHE_HIP_XOR: g = {2 * kMU, 2, 2}; return true;
        case TFHE_HIP_XNOR: g = {-2 * kMU, -2, -2}; return true;
        case TFHE_HIP_NOR: g = {-kMU, -1, -1}; return true;
        case TFHE_HIP_ANDNY: g = {-kMU, -1, 1}; return true;
        case TFHE_HIP_ANDYN: g = {-kMU, 1, -1}; return true;
        case TFHE_HIP_ORNY: g = {kMU, -1, 1}; return true;
        case TFHE_HIP_ORYN: g = {kMU, 1, -1}; return true;
        default: return false;
    }
}

// FAST mode: (re)derive its key from the exact mode's; called whenever keys were loaded or the mode was selected
int build_fast_key(tfhe_hip_ctx *c) {
    if (c->arith != TFHE_HIP_ARITH_FAST || !c->bk) return TFHE_HIP_OK;
    const long long total = (long long) kLwe * kKpl * 2 * kNh;
    if (!c->bkf) HIPCHK(hipMalloc((void **) &c->bkf, (size_t) total * sizeof(c64)));
    hipLaunchKernelGGL(k_bk_to_fast, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, c->stream, c->bkf,
                       (const c64 *) c->bk, total);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    return TFHE_HIP_OK;
}

int check_ctx(tfhe_hip_ctx *c, bool need_keys) {
    if (!c) return fail(TFHE_HIP_ERR_ARG, "null context");
    HIPCHK(hipSetDevice(c->device));
    if (need_keys && !c->have_keys) return fail(TFHE_HIP_ERR_NOKEY, "keys not loaded");
    return TFHE_HIP_OK;
}

// the keyswitch key in the host layout [N][t][4][n+1] sits in c->misc: pack it into the two device layouts
int pack_ks_from_misc(tfhe_hip_ctx *c);
int upload_ks(tfhe_hip_ctx *c, const int32_t *ks) {
    HIPCHK(c->misc.reserve(kKsHostElems * sizeof(int32_t)));
    HIPCHK(hipMemcpyAsync(c->misc.p, ks, kKsHostElems * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    return pack_ks_from_misc(c);
}
int pack_ks_from_misc(tfhe_hip_ctx *c) {
    if (!c->ks) HIPCHK(hipMalloc((void **) &c->ks, kKsDevBytes));
    const long long rows = (long long) kN * kKsT * 3;
    hipLaunchKernelGGL(k_ks_pack, dim3((unsigned) rows), dim3(256), 0, c->stream, c->ks, (const int32_t *) c->misc.p, rows);
    HIPCHK(hipGetLastError());
    if (!c->kl) HIPCHK(hipMalloc((void **) &c->kl, (size_t) kN * 4 * kKsQ * 32));
    hipLaunchKernelGGL(k_ks_pack_mfma, dim3(kN), dim3(128), 0, c->stream, c->kl, (const int32_t *) c->misc.p);
    HIPCHK(hipGetLastError());
    return TFHE_HIP_OK;
}

}  // namespace

int tfhe_hip_internal_fail(int code, const char *msg) { return fail(code, msg ? msg : ""); }

extern "C" {

const char *tfhe_hip_last_error(void) { return g_err.c_str(); }
void tfhe_hip_param_set(int32_t out[8]) {
    const int32_t v[8] = {TFHE_HIP_LAMBDA, kLwe, kN, 1, kL, kBgbit, kKsT, kKsBasebit};
    for (int i = 0; i < 8; i++) out[i] = v[i];
}

int tfhe_hip_create(tfhe_hip_ctx **out, int device) {
    if (!out) return fail(TFHE_HIP_ERR_ARG, "null out pointer");
    *out = nullptr;
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(TFHE_HIP_ERR_ARG, "no such device");
    std::vector<c64> tw;
    if (!build_twiddles(tw))
        return fail(TFHE_HIP_ERR_TABLE, "twiddle table fingerprint mismatch: host libm cos/sin round differently");
    HIPCHK(hipSetDevice(device));
    HIPCHK(noalloc_scratch_init(device));
    tfhe_hip_ctx *c = new tfhe_hip_ctx();
    c->device = device;
    c->ka = make_phaseA_consts([&](int i) { return tw[i]; });
    const char *w = getenv("TFHE_HIP_BR_WAVES");
    if (w && (atoi(w) == 8 || atoi(w) == 4 || atoi(w) == 1 || atoi(w) == kShapeCoop)) c->br_waves = atoi(w);
    const char *fc = getenv("TFHE_HIP_TEST_FAIL_CHUNK");
    if (fc) c->fault_chunk = atoll(fc);
    const char *km = getenv("TFHE_HIP_KS_MFMA");
    if (km) c->ks_mfma = atoi(km) != 0;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
    std::vector<c64> image(kTwLaneEntries);
    build_lane_twiddles(image.data(), [&](int i) { return tw[i]; });
    image.resize(kTwLaneEntries + kTwHalfEntries);  // followed by the image of the split inverse (latency kernel)
    build_half_twiddles(image.data() + kTwLaneEntries, [&](int i) { return tw[i]; });
    if (e == hipSuccess) e = hipMalloc((void **) &c->tw, image.size() * sizeof(c64));
    if (e == hipSuccess) e = hipMemcpy(c->tw, image.data(), image.size() * sizeof(c64), hipMemcpyHostToDevice);
    std::vector<c64> fimage(brf::kFastLaneEntries);
    brf::build_fast_twiddles(fimage.data());
    if (e == hipSuccess) e = hipMalloc((void **) &c->twf, fimage.size() * sizeof(c64));
    if (e == hipSuccess) e = hipMemcpy(c->twf, fimage.data(), fimage.size() * sizeof(c64), hipMemcpyHostToDevice);
    const char *ar = getenv("TFHE_HIP_ARITH");  // default arithmetic of new contexts: "fast" or "exact"
    if (ar && (!strcmp(ar, "fast") || !strcmp(ar, "FAST") || !strcmp(ar, "1"))) c->arith = TFHE_HIP_ARITH_FAST;
    for (int i = 0; i < 4 && e == hipSuccess; i++) e = hipEventCreate(&c->ev[i]);
    if (e != hipSuccess) {
        tfhe_hip_destroy(c);
        return fail(TFHE_HIP_ERR_HIP, std::string("context creation: ") + hipGetErrorString(e));
    }
    *out = c;
    return TFHE_HIP_OK;
}

void tfhe_hip_destroy(tfhe_hip_ctx *c) {
    if (!c) return;
    (void) hipSetDevice(c->device);
    if (c->stream) (void) hipStreamSynchronize(c->stream);
    for (DevBuf *b : {&c->in_a, &c->in_b, &c->in_c, &c->ext1, &c->ext2, &c->outb, &c->misc, &c->abart, &c->ks_part, &c->jobs_br, &c->jobs_ks,
                      &c->jobs_lin, &c->fp, &c->grp_stage, &c->grp_in, &c->grp_rows}) b->release();
    if (c->tw) (void) hipFree(c->tw);
    if (c->twf) (void) hipFree(c->twf);
    if (c->cdt) (void) hipFree(c->cdt);
    for (c64 *t : c->ntab)
        if (t) (void) hipFree(t);
    if (c->bkf) (void) hipFree(c->bkf);
    if (c->bk) (void) hipFree(c->bk);
    if (c->ks) (void) hipFree(c->ks);
    if (c->kl) (void) hipFree(c->kl);
    for (int i = 0; i < 4; i++)
        if (c->ev[i]) (void) hipEventDestroy(c->ev[i]);
    for (hipEvent_t ev : c->chunk_ev) (void) hipEventDestroy(ev);
    if (c->copy_stream) (void) hipStreamDestroy(c->copy_stream);
    if (c->stream) (void) hipStreamDestroy(c->stream);
    delete c;
}

void *tfhe_hip_stream(tfhe_hip_ctx *c) { return c ? (void *) c->stream : nullptr; }

int tfhe_hip_sync(tfhe_hip_ctx *c) {
    if (int r = check_ctx(c, false)) return r;
    HIPCHK(hipStreamSynchronize(c->stream));
    return TFHE_HIP_OK;
}

int tfhe_hip_set_arithmetic(tfhe_hip_ctx *c, int mode) {
    if (int r = check_ctx(c, false)) return r;
    if (mode != TFHE_HIP_ARITH_EXACT && mode != TFHE_HIP_ARITH_FAST) return fail(TFHE_HIP_ERR_ARG, "unknown arithmetic mode");
    HIPCHK(hipStreamSynchronize(c->stream));
    c->arith = mode;
    if (c->have_keys) return build_fast_key(c);
    return TFHE_HIP_OK;
}
int tfhe_hip_get_arithmetic(tfhe_hip_ctx *c) { return c ? c->arith : TFHE_HIP_ERR_ARG; }

// the coefficient-form bootstrapping key sits in c->misc: transform it, then upload the keyswitch key
static int keys_from_coef_in_misc(tfhe_hip_ctx *c, const int32_t *ks) {
    const int npoly = kLwe * kKpl * 2;
    HIPCHK(hipFuncSetAttribute((const void *) k_bk_convert, hipFuncAttributeMaxDynamicSharedMemorySize, kFftLds));
    hipLaunchKernelGGL(k_bk_convert, dim3((npoly + 3) / 4), dim3(256), kFftLds, c->stream, (const int32_t *) c->misc.p,
                       c->bk, (const c64 *) c->tw, c->ka, npoly);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));  // misc is reused by upload_ks
    if (ks) {  // null: the keyswitch key has been packed on the device already (tfhe_hip_keygen_cloud)
        if (int r = upload_ks(c, ks)) return r;
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    c->misc.release();
    if (int r = build_fast_key(c)) return r;
    c->have_keys = true;
    return TFHE_HIP_OK;
}

int tfhe_hip_load_keys_coef(tfhe_hip_ctx *c, const int32_t *bk_coef, const int32_t *ks) {
    if (int r = check_ctx(c, false)) return r;
    if (!bk_coef || !ks) return fail(TFHE_HIP_ERR_ARG, "null key pointer");
    c->have_keys = false;
    const int npoly = kLwe * kKpl * 2;
    const size_t coef_bytes = (size_t) npoly * kN * sizeof(int32_t);
    if (!c->bk) HIPCHK(hipMalloc((void **) &c->bk, kBkBytes));
    HIPCHK(c->misc.reserve(coef_bytes));
    HIPCHK(hipMemcpyAsync(c->misc.p, bk_coef, coef_bytes, hipMemcpyHostToDevice, c->stream));
    return keys_from_coef_in_misc(c, ks);
}

// Keys replicated by the caller's own transport (an RCCL broadcast between one-process-per-GPU ranks): the three
// device-resident key buffers of a context, allocated here if it has none yet, and the call that declares them filled.
int tfhe_hip_keys_device_buffers(tfhe_hip_ctx *c, void *ptrs[3], size_t bytes[3]) {
    if (int r = check_ctx(c, false)) return r;
    if (!ptrs || !bytes) return fail(TFHE_HIP_ERR_ARG, "null output array");
    const size_t kl_bytes = (size_t) kN * 4 * kKsQ * 32;
    if (!c->bk) HIPCHK(hipMalloc((void **) &c->bk, kBkBytes));
    if (!c->ks) HIPCHK(hipMalloc((void **) &c->ks, kKsDevBytes));
    if (!c->kl) HIPCHK(hipMalloc((void **) &c->kl, kl_bytes));
    ptrs[0] = c->bk, ptrs[1] = c->ks, ptrs[2] = c->kl;
    bytes[0] = kBkBytes, bytes[1] = kKsDevBytes, bytes[2] = kl_bytes;
    return TFHE_HIP_OK;
}
int tfhe_hip_keys_adopt(tfhe_hip_ctx *c) {
    if (int r = check_ctx(c, false)) return r;
    if (!c->bk || !c->ks || !c->kl) return fail(TFHE_HIP_ERR_NOKEY, "no key buffers: call tfhe_hip_keys_device_buffers first");
    c->have_keys = false;
    HIPCHK(hipStreamSynchronize(c->stream));
    if (int r = build_fast_key(c)) return r;
    c->have_keys = true;
    return TFHE_HIP_OK;
}
int tfhe_hip_keys_fingerprint(tfhe_hip_ctx *c, uint64_t out[3]) {
    if (int r = check_ctx(c, false)) return r;
    if (!out) return fail(TFHE_HIP_ERR_ARG, "null output array");
    if (!c->bk || !c->ks || !c->kl) return fail(TFHE_HIP_ERR_NOKEY, "no key buffers");
    const void *ptrs[3] = {c->bk, c->ks, c->kl};
    const size_t bytes[3] = {kBkBytes, kKsDevBytes, (size_t) kN * 4 * kKsQ * 32};
    HIPCHK(c->fp.reserve(3 * sizeof(unsigned long long)));
    HIPCHK(hipMemsetAsync(c->fp.p, 0, 3 * sizeof(unsigned long long), c->stream));
    for (int k = 0; k < 3; k++)
        hipLaunchKernelGGL(k_fingerprint, dim3(2048), dim3(256), 0, c->stream, (const unsigned long long *) ptrs[k],
                           (long long) (bytes[k] / 8), (unsigned long long *) c->fp.p + k);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, c->fp.p, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return TFHE_HIP_OK;
}

}  // extern "C"
namespace {
// xoshiro256** state transition as a 256 x 256 matrix over GF(2), rows as 4 words: (M v)_i = parity(M[i] & v)
struct BitMat {
    uint64_t r[256][4];
};
void xoshiro_step(uint64_t s[4]) {
    const uint64_t t = s[1] << 17;
    s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = (s[3] << 45) | (s[3] >> 19);
}
void bitmat_mul(BitMat &out, const BitMat &a, const BitMat &b) {  // out = a * b: row i = XOR of the rows k of b with a[i][k] set
    for (int i = 0; i < 256; i++) {
        uint64_t acc[4] = {0, 0, 0, 0};
        for (int k = 0; k < 256; k++)
            if ((a.r[i][k >> 6] >> (k & 63)) & 1)
                for (int w = 0; w < 4; w++) acc[w] ^= b.r[k][w];
        for (int w = 0; w < 4; w++) out.r[i][w] = acc[w];
    }
}
// jump[k] = T^(draws * 2^k), k < nbits
void build_jump_matrices(std::vector<BitMat> &jump, int draws, int nbits) {
    auto T = std::make_unique<BitMat>(), acc = std::make_unique<BitMat>(), tmp = std::make_unique<BitMat>();
    std::memset(T.get(), 0, sizeof(BitMat));
    for (int bcol = 0; bcol < 256; bcol++) {  // column bcol = image of the basis vector e_bcol
        uint64_t s[4] = {0, 0, 0, 0};
        s[bcol >> 6] = 1ull << (bcol & 63);
        xoshiro_step(s);
        for (int i = 0; i < 256; i++)
            if ((s[i >> 6] >> (i & 63)) & 1) T->r[i][bcol >> 6] |= 1ull << (bcol & 63);
    }
    std::memset(acc.get(), 0, sizeof(BitMat));
    for (int i = 0; i < 256; i++) acc->r[i][i >> 6] = 1ull << (i & 63);  // identity
    auto base = std::make_unique<BitMat>(*T);
    for (int e = draws; e; e >>= 1) {  // acc = T^draws
        if (e & 1) { bitmat_mul(*tmp, *base, *acc); std::swap(acc, tmp); }
        bitmat_mul(*tmp, *base, *base);
        std::swap(base, tmp);
    }
    jump.resize(nbits);
    jump[0] = *acc;
    for (int k = 1; k < nbits; k++) bitmat_mul(jump[k], jump[k - 1], jump[k - 1]);
}
uint64_t host_splitmix(uint64_t &x) {
    uint64_t z = (x += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
}  // namespace
extern "C" {

}  // extern "C"
namespace {
// the noise table of the client-side kernels: built on the host (fingerprint-checked) and uploaded on first use
int ensure_cdt(tfhe_hip_ctx *c) {
    if (c->cdt) return TFHE_HIP_OK;
    static std::vector<uint64_t> table;
    static int ok = 0;
    static std::once_flag once;
    std::call_once(once, [] {
        table.resize(GAUSS_CDT_BINS);
        ok = gauss_cdt_build(table.data(), nullptr);
    });
    if (!ok) return fail(TFHE_HIP_ERR_TABLE, "noise table self check failed (host libm rounds erfcl differently)");
    HIPCHK(hipMalloc((void **) &c->cdt, GAUSS_CDT_BINS * sizeof(uint64_t)));
    HIPCHK(hipMemcpyAsync(c->cdt, table.data(), GAUSS_CDT_BINS * sizeof(uint64_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return TFHE_HIP_OK;
}
// zeroes a device region that held secret key material when the enclosing function leaves, whichever way it leaves
struct DeviceWipe {
    tfhe_hip_ctx *c;
    void *p = nullptr;
    size_t bytes = 0;
    bool failed = false;
    void arm(void *ptr, size_t n) { p = ptr; bytes = n; }
    void now() {
        if (!p) return;
        const hipError_t w1 = hipMemsetAsync(p, 0, bytes, c->stream);
        const hipError_t w2 = hipStreamSynchronize(c->stream);
        failed = failed || w1 != hipSuccess || w2 != hipSuccess;
        p = nullptr;
    }
    ~DeviceWipe() { now(); }
};
}  // namespace
extern "C" {

int tfhe_hip_keygen_cloud(tfhe_hip_ctx *c, uint64_t seed, const int32_t *lwe_key, const int32_t *tlwe_key,
                          int32_t *bk_coef_out, int32_t *ks_out) {
    if (int r = check_ctx(c, false)) return r;
    if (!lwe_key || !tlwe_key) return fail(TFHE_HIP_ERR_ARG, "null key pointer");
    if (int r = ensure_cdt(c)) return r;
    c->have_keys = false;
    static std::vector<BitMat> jump;  // depends on the parameter set only
    static std::once_flag once;
    std::call_once(once, [] { build_jump_matrices(jump, kKsDraws, kKsJumpBits); });
    ClientRng start;  // stream 3 of the seed: tfhe_client_keygen (client_keys.c)
    uint64_t x = seed ^ (3ULL * 0xD1342543DE82EF95ULL + 0x632BE59BD9B4E019ULL);
    for (int i = 0; i < 4; i++) start.s[i] = host_splitmix(x);
    const size_t jump_bytes = (size_t) kKsJumpBits * sizeof(BitMat);
    HIPCHK(c->misc.reserve(kKsHostElems * sizeof(int32_t)));
    HIPCHK(c->abart.reserve(jump_bytes + (kLwe + kN) * sizeof(int32_t)));
    uint64_t *d_jump = (uint64_t *) c->abart.p;
    int32_t *d_lwe = (int32_t *) ((unsigned char *) c->abart.p + jump_bytes), *d_tlwe = d_lwe + kLwe;
    DeviceWipe wipe{c};
    wipe.arm(d_lwe, (kLwe + kN) * sizeof(int32_t));
    HIPCHK(hipMemcpyAsync(d_jump, jump.data(), jump_bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_lwe, lwe_key, kLwe * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_tlwe, tlwe_key, kN * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(c->misc.p, 0, kKsHostElems * sizeof(int32_t), c->stream));  // the h = 0 rows
    static_assert(kN * kKsT * 3 % 64 == 0, "whole workgroups of samples");
    hipLaunchKernelGGL(k_client_ks_draw, dim3(kN * kKsT * 3 / 64), dim3(64), 0, c->stream, start, (const uint64_t *) d_jump,
                       (const int32_t *) d_lwe, (const int32_t *) d_tlwe, (int32_t *) c->misc.p, (const uint64_t *) c->cdt);
    HIPCHK(hipGetLastError());
    if (ks_out) HIPCHK(hipMemcpyAsync(ks_out, c->misc.p, kKsHostElems * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    int r = pack_ks_from_misc(c);
    wipe.now();
    if (r == TFHE_HIP_OK && wipe.failed) r = fail(TFHE_HIP_ERR_HIP, "could not wipe the secret keys from device memory");
    if (r) return r;
    return tfhe_hip_keygen_bk(c, seed, lwe_key, tlwe_key, nullptr, bk_coef_out);
}

int tfhe_hip_keygen_bk(tfhe_hip_ctx *c, uint64_t seed, const int32_t *lwe_key, const int32_t *tlwe_key, const int32_t *ks,
                       int32_t *bk_coef_out) {
    if (int r = check_ctx(c, false)) return r;
    if (!lwe_key || !tlwe_key) return fail(TFHE_HIP_ERR_ARG, "null key pointer");
    if (!ks && !c->kl) return fail(TFHE_HIP_ERR_ARG, "no keyswitch key given and none on the device");
    if (int r = ensure_cdt(c)) return r;
    c->have_keys = false;
    const size_t coef_bytes = (size_t) kLwe * kKpl * 2 * kN * sizeof(int32_t);
    if (!c->bk) HIPCHK(hipMalloc((void **) &c->bk, kBkBytes));
    HIPCHK(c->misc.reserve(coef_bytes));
    HIPCHK(c->abart.reserve((kLwe + kN) * sizeof(int32_t)));
    int32_t *d_lwe = (int32_t *) c->abart.p, *d_tlwe = d_lwe + kLwe;
    DeviceWipe wipe{c};  // the secret keys do not stay on the device, whatever happens below
    wipe.arm(d_lwe, (kLwe + kN) * sizeof(int32_t));
    HIPCHK(hipMemcpyAsync(d_lwe, lwe_key, kLwe * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_tlwe, tlwe_key, kN * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_client_bk_draw, dim3((kLwe + 63) / 64), dim3(64), 0, c->stream, seed, (int32_t *) c->misc.p,
                       (const uint64_t *) c->cdt);
    hipLaunchKernelGGL(k_client_bk_mask, dim3(kLwe * kKpl), dim3(256), 0, c->stream, (const int32_t *) d_lwe,
                       (const int32_t *) d_tlwe, (int32_t *) c->misc.p);
    HIPCHK(hipGetLastError());
    if (bk_coef_out) HIPCHK(hipMemcpyAsync(bk_coef_out, c->misc.p, coef_bytes, hipMemcpyDeviceToHost, c->stream));
    const int r = keys_from_coef_in_misc(c, ks);
    wipe.now();
    if (r == TFHE_HIP_OK && wipe.failed) {
        c->have_keys = false;
        return fail(TFHE_HIP_ERR_HIP, "could not wipe the secret keys from device memory");
    }
    return r;
}

int tfhe_hip_load_keys_fft(tfhe_hip_ctx *c, const double *bkfft, const int32_t *ks) {
    if (int r = check_ctx(c, false)) return r;
    if (!bkfft || !ks) return fail(TFHE_HIP_ERR_ARG, "null key pointer");
    c->have_keys = false;
    if (!c->bk) HIPCHK(hipMalloc((void **) &c->bk, kBkBytes));
    HIPCHK(c->misc.reserve(kBkBytes));
    HIPCHK(hipMemcpyAsync(c->misc.p, bkfft, kBkBytes, hipMemcpyHostToDevice, c->stream));
    const long long total = (long long) kLwe * kKpl * 2 * kNh;
    hipLaunchKernelGGL(k_bk_permute, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, c->stream, c->bk,
                       (const c64 *) c->misc.p, total, 1);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    if (int r = upload_ks(c, ks)) return r;
    HIPCHK(hipStreamSynchronize(c->stream));
    c->misc.release();
    if (int r = build_fast_key(c)) return r;
    c->have_keys = true;
    return TFHE_HIP_OK;
}

int tfhe_hip_get_bkfft(tfhe_hip_ctx *c, double *out) {
    if (int r = check_ctx(c, true)) return r;
    if (!out) return fail(TFHE_HIP_ERR_ARG, "null output");
    HIPCHK(c->misc.reserve(kBkBytes));
    const long long total = (long long) kLwe * kKpl * 2 * kNh;
    hipLaunchKernelGGL(k_bk_permute, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, c->stream, (c64 *) c->misc.p,
                       (const c64 *) c->bk, total, 0);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, c->misc.p, kBkBytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->misc.release();
    return TFHE_HIP_OK;
}

// ---- device-resident entry points -----------------------------------------------------------------------------
int tfhe_hip_bootstrap_batch_dev(tfhe_hip_ctx *c, int32_t *d_out, const int32_t *d_in, int32_t mu, int64_t count,
                                 int with_ks) {
    if (int r = check_ctx(c, true)) return r;
    if (count < 0 || (count > 0 && (!d_out || !d_in))) return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    c->ev_br = c->ev_ks = false;
    if (count == 0) return TFHE_HIP_OK;
    int32_t *ext = d_out;
    if (with_ks) {
        HIPCHK(c->ext1.reserve((size_t) count * kExt * sizeof(int32_t)));
        ext = (int32_t *) c->ext1.p;
    }
    BrArgs A{};
    A.a = d_in; A.b = nullptr; A.sa = 1; A.sb = 0; A.cst = 0; A.mu = mu; A.out = ext;
    A.bk = c->bk; A.tw = c->tw; A.ka = c->ka; A.B = count;
    HIPCHK(hipEventRecord(c->ev[0], c->stream));
    if (int r = launch_br<MODE_GATE>(c, A)) return r;
    HIPCHK(hipEventRecord(c->ev[1], c->stream));
    c->ev_br = true;
    if (with_ks) {
        HIPCHK(hipEventRecord(c->ev[2], c->stream));
        if (int r = launch_ks(c, d_out, ext, count)) return r;
        HIPCHK(hipEventRecord(c->ev[3], c->stream));
        c->ev_ks = true;
    }
    return TFHE_HIP_OK;
}

int tfhe_hip_keyswitch_batch_dev(tfhe_hip_ctx *c, int32_t *d_out, const int32_t *d_in, int64_t count) {
    if (int r = check_ctx(c, true)) return r;
    if (count < 0 || (count > 0 && (!d_out || !d_in))) return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    c->ev_br = c->ev_ks = false;
    if (count == 0) return TFHE_HIP_OK;
    HIPCHK(hipEventRecord(c->ev[2], c->stream));
    if (int r = launch_ks(c, d_out, d_in, count)) return r;
    HIPCHK(hipEventRecord(c->ev[3], c->stream));
    c->ev_ks = true;
    return TFHE_HIP_OK;
}

// work buffers of one gate batch: the extracted samples of the one or (MUX) two blind rotations, and the keyswitch's
struct GateWork {
    int32_t *ext1, *ext2;
    KsWork ks;
};
static int gate_batch_dev_impl(tfhe_hip_ctx *c, int gate, int32_t *d_out, const int32_t *d_a, const int32_t *d_b,
                               const int32_t *d_c, int64_t count, const GateWork *gw) {
    if (int r = check_ctx(c, gate <= TFHE_HIP_MUX)) return r;
    if (gate < TFHE_HIP_NAND || gate > TFHE_HIP_CONST0) return fail(TFHE_HIP_ERR_ARG, "unknown gate id");
    if (count < 0 || (count > 0 && !d_out)) return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    c->ev_br = c->ev_ks = false;
    if (count == 0) return TFHE_HIP_OK;
    const bool unary = gate == TFHE_HIP_NOT || gate == TFHE_HIP_COPY;
    const bool nullary = gate == TFHE_HIP_CONST0 || gate == TFHE_HIP_CONST1;
    if (!nullary && !d_a) return fail(TFHE_HIP_ERR_ARG, "null input a");
    if (!nullary && !unary && !d_b) return fail(TFHE_HIP_ERR_ARG, "null input b");
    if (gate == TFHE_HIP_MUX && !d_c) return fail(TFHE_HIP_ERR_ARG, "null input c");
    if (unary || nullary) {
        const long long total = (long long) count * kCt;
        hipLaunchKernelGGL(k_linear_gate, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, c->stream, d_out, d_a,
                           gate, kMU, total);
        HIPCHK(hipGetLastError());
        return TFHE_HIP_OK;
    }
    int32_t *ext1, *ext2 = nullptr;
    if (gw) {
        ext1 = gw->ext1;
        ext2 = gw->ext2;
    } else {
        HIPCHK(c->ext1.reserve((size_t) count * kExt * sizeof(int32_t)));
        ext1 = (int32_t *) c->ext1.p;
        if (gate == TFHE_HIP_MUX) {
            HIPCHK(c->ext2.reserve((size_t) count * kExt * sizeof(int32_t)));
            ext2 = (int32_t *) c->ext2.p;
        }
    }
    BrArgs A{};
    A.mu = kMU; A.bk = c->bk; A.tw = c->tw; A.ka = c->ka; A.B = count;
    HIPCHK(hipEventRecord(c->ev[0], c->stream));
    if (gate == TFHE_HIP_MUX) {  // boot-gates.cpp:329-370
        A.a = d_a; A.b = d_b; A.sa = 1; A.sb = 1; A.cst = -kMU; A.out = ext1;
        if (int r = launch_br<MODE_GATE>(c, A)) return r;
        A.a = d_a; A.b = d_c; A.sa = -1; A.sb = 1; A.cst = -kMU; A.out = ext2;
        if (int r = launch_br<MODE_GATE>(c, A)) return r;
        const long long total = (long long) count * kExt;
        hipLaunchKernelGGL(k_mux_combine, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, c->stream, ext1,
                           (const int32_t *) ext2, kMU, total);
        HIPCHK(hipGetLastError());
    } else {
        GateLin g;
        gate_lin(gate, g);
        A.a = d_a; A.b = d_b; A.sa = g.sa; A.sb = g.sb; A.cst = g.cst; A.out = ext1;
        if (int r = launch_br<MODE_GATE>(c, A)) return r;
    }
    HIPCHK(hipEventRecord(c->ev[1], c->stream));
    c->ev_br = true;
    HIPCHK(hipEventRecord(c->ev[2], c->stream));
    if (int r = launch_ks(c, d_out, (const int32_t *) ext1, count, nullptr, gw ? &gw->ks : nullptr)) return r;
    HIPCHK(hipEventRecord(c->ev[3], c->stream));
    c->ev_ks = true;
    return TFHE_HIP_OK;
}

int tfhe_hip_gate_batch_dev(tfhe_hip_ctx *c, int gate, int32_t *d_out, const int32_t *d_a, const int32_t *d_b,
                            const int32_t *d_c, int64_t count) {
    return gate_batch_dev_impl(c, gate, d_out, d_a, d_b, d_c, count, nullptr);
}

// ---- the entry point that allocates nothing (include/tfhe_hip.h) ---------------------------------------------------
// Work buffers = slices of the file-scope device array g_noalloc_scratch; its per-device address is looked up once per
// device in tfhe_hip_create (noalloc_scratch_init) and only that pointer is ever handed to kernels.
int tfhe_hip_gate_batch_dev_noalloc(tfhe_hip_ctx *c, int gate, int32_t *d_out, const int32_t *d_a, const int32_t *d_b,
                                    const int32_t *d_c, int64_t count) {
    if (!c) return fail(TFHE_HIP_ERR_ARG, "null context");
    if (count > TFHE_HIP_NOALLOC_MAX_GATES)
        return fail(TFHE_HIP_ERR_ARG, "tfhe_hip_gate_batch_dev_noalloc: count exceeds TFHE_HIP_NOALLOC_MAX_GATES (" +
                                          std::to_string(TFHE_HIP_NOALLOC_MAX_GATES) + ")");
    unsigned char *base = noalloc_scratch_ptr(c->device);
    if (!base) return fail(TFHE_HIP_ERR_HIP, "the static scratch array of this device was not resolved at tfhe_hip_create");
    GateWork gw;
    gw.ext1 = (int32_t *) (base + kNaExt1Off);
    gw.ext2 = (int32_t *) (base + kNaExt2Off);
    gw.ks = KsWork{(uint32_t *) (base + kNaAbartOff), kNaAbartBytes, (uint32_t *) (base + kNaPartOff), kNaPartBytes};
    return gate_batch_dev_impl(c, gate, d_out, d_a, d_b, d_c, count, &gw);
}

int tfhe_hip_last_kernel_ms(tfhe_hip_ctx *c, float *br_ms, float *ks_ms) {
    if (int r = check_ctx(c, false)) return r;
    HIPCHK(hipStreamSynchronize(c->stream));
    float a = 0.f, b = 0.f;
    if (c->ev_br) HIPCHK(hipEventElapsedTime(&a, c->ev[0], c->ev[1]));
    if (c->ev_ks) HIPCHK(hipEventElapsedTime(&b, c->ev[2], c->ev[3]));
    if (br_ms) *br_ms = a;
    if (ks_ms) *ks_ms = b;
    return TFHE_HIP_OK;
}

// ---- host-buffer entry points -----------------------------------------------------------------------------------
static int h2d(tfhe_hip_ctx *c, DevBuf &buf, const void *src, size_t bytes) {
    HIPCHK(buf.reserve(bytes));
    HIPCHK(hipMemcpyAsync(buf.p, src, bytes, hipMemcpyHostToDevice, c->stream));
    return TFHE_HIP_OK;
}
static int d2h(tfhe_hip_ctx *c, void *dst, const void *src, size_t bytes) {
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return TFHE_HIP_OK;
}

// Host buffers.  Batches above one chunk are pipelined: the copies of chunk k+1 (in) and chunk k-1 (out) run on a
// second stream while the kernels of chunk k run, so that PCIe time hides behind the blind rotation.
int tfhe_hip_gate_batch(tfhe_hip_ctx *c, int gate, int32_t *out, const int32_t *a, const int32_t *b, const int32_t *cc,
                        int64_t count) {
    return tfhe_hip_internal_gate_batch(c, gate, out, a, b, cc, count, nullptr, nullptr, nullptr);
}
}  // extern "C"

// Host buffers: chunks of 4,096 gates (two full rounds of the blind-rotation kernel; the first chunk is one round,
// so that less copying - and, with hooks, packing - stands before the first kernel).  The copies of chunk k+1 and
// k-1 run on a second stream, and the hooks on the calling thread, while the kernels of chunk k run.
int tfhe_hip_internal_gate_batch(tfhe_hip_ctx *c, int gate, int *out, const int *a, const int *b, const int *cc,
                                 long long count, tfhe_hip_chunk_hook fill, tfhe_hip_chunk_hook drain, void *user) {
    if (int r = check_ctx(c, false)) return r;
    if (count < 0 || (count > 0 && !out)) return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    if (count == 0) return TFHE_HIP_OK;
    const size_t bytes = (size_t) count * kCt * sizeof(int32_t);
    constexpr int64_t kChunk = 4096, kFirst = 2048;
    if (count <= kChunk) {
        if (fill) fill(user, 0, count);
        if (a) if (int r = h2d(c, c->in_a, a, bytes)) return r;
        if (b) if (int r = h2d(c, c->in_b, b, bytes)) return r;
        if (cc) if (int r = h2d(c, c->in_c, cc, bytes)) return r;
        HIPCHK(c->outb.reserve(bytes));
        if (int r = tfhe_hip_gate_batch_dev(c, gate, (int32_t *) c->outb.p, a ? (const int32_t *) c->in_a.p : nullptr,
                                            b ? (const int32_t *) c->in_b.p : nullptr,
                                            cc ? (const int32_t *) c->in_c.p : nullptr, count))
            return r;
        if (int r = d2h(c, out, c->outb.p, bytes)) return r;
        if (drain) drain(user, 0, count);
        return TFHE_HIP_OK;
    }
    if (a) HIPCHK(c->in_a.reserve(bytes));
    if (b) HIPCHK(c->in_b.reserve(bytes));
    if (cc) HIPCHK(c->in_c.reserve(bytes));
    HIPCHK(c->outb.reserve(bytes));
    // every work buffer at the size of the largest chunk BEFORE the first launch: growing one later would free memory
    // that the kernels of the previous chunk may still be using (and stall the overlap on hipFree's implicit sync)
    if (gate <= TFHE_HIP_MUX) {
        HIPCHK(c->ext1.reserve((size_t) kChunk * kExt * sizeof(int32_t)));
        if (gate == TFHE_HIP_MUX) HIPCHK(c->ext2.reserve((size_t) kChunk * kExt * sizeof(int32_t)));
        const long long Bp = (kChunk + 32 * kKsMT - 1) / (32 * kKsMT) * (32 * kKsMT);
        HIPCHK(c->abart.reserve((size_t) kN * Bp * sizeof(uint32_t)));
        size_t part = 0;  // chunks are 2,048 or 4,096 gates or a ragged tail: the largest partial-sum buffer any needs
        for (long long bp = 32 * kKsMT; bp <= Bp; bp += 32 * kKsMT) {
            const int S = ks_split(bp);
            if (S > 1) part = std::max<size_t>(part, (size_t) S * (size_t) bp * kCt * sizeof(uint32_t));
        }
        if (part) HIPCHK(c->ks_part.reserve(part));
    }
    const int64_t nchunks = 1 + (count - kFirst + kChunk - 1) / kChunk;
    while ((int64_t) c->chunk_ev.size() < 3 * nchunks) {
        hipEvent_t ev;
        HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        c->chunk_ev.push_back(ev);
    }
    // From here on asynchronous copies target the caller's buffers: no early return.  A failing HIP call records the
    // error (hipfail), the loop stops, and both streams are drained before the function reports it.
    auto hipfail = [&](hipError_t e, const char *what) -> int {
        return e == hipSuccess ? TFHE_HIP_OK : fail(TFHE_HIP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    };
#define STEP(x)                                                \
    do {                                                       \
        if (int r_ = hipfail((x), #x)) return r_;              \
    } while (0)
    auto span = [&](int64_t k, int64_t &off, int64_t &n) {
        off = k == 0 ? 0 : kFirst + (k - 1) * kChunk;
        const int64_t want = k == 0 ? kFirst : kChunk;
        n = count - off < want ? count - off : want;
    };
    auto copy_in = [&](int64_t k) -> int {  // events of chunk k: [3k] inputs on the device, [3k+1] kernels done, [3k+2] output on the host
        int64_t off, n;
        span(k, off, n);
        if (fill) fill(user, off, n);
        const size_t o = (size_t) off * kCt, nb = (size_t) n * kCt * sizeof(int32_t);
        if (a) STEP(hipMemcpyAsync((int32_t *) c->in_a.p + o, a + o, nb, hipMemcpyHostToDevice, c->copy_stream));
        if (b) STEP(hipMemcpyAsync((int32_t *) c->in_b.p + o, b + o, nb, hipMemcpyHostToDevice, c->copy_stream));
        if (cc) STEP(hipMemcpyAsync((int32_t *) c->in_c.p + o, cc + o, nb, hipMemcpyHostToDevice, c->copy_stream));
        STEP(hipEventRecord(c->chunk_ev[3 * k], c->copy_stream));
        return TFHE_HIP_OK;
    };
    auto copy_out = [&](int64_t k) -> int {
        int64_t off, n;
        span(k, off, n);
        const size_t o = (size_t) off * kCt, nb = (size_t) n * kCt * sizeof(int32_t);
        STEP(hipStreamWaitEvent(c->copy_stream, c->chunk_ev[3 * k + 1], 0));
        STEP(hipMemcpyAsync(out + o, (int32_t *) c->outb.p + o, nb, hipMemcpyDeviceToHost, c->copy_stream));
        STEP(hipEventRecord(c->chunk_ev[3 * k + 2], c->copy_stream));
        return TFHE_HIP_OK;
    };
    auto drain_chunk = [&](int64_t k) -> int {
        if (!drain) return TFHE_HIP_OK;
        int64_t off, n;
        span(k, off, n);
        STEP(hipEventSynchronize(c->chunk_ev[3 * k + 2]));
        drain(user, off, n);
        return TFHE_HIP_OK;
    };
    auto run_chunk = [&](int64_t k) -> int {
        int64_t off, n;
        span(k, off, n);
        const size_t o = (size_t) off * kCt;
        STEP(hipStreamWaitEvent(c->stream, c->chunk_ev[3 * k], 0));
        if (c->fault_chunk >= 0 && k == c->fault_chunk) {  // test hook (TFHE_HIP_TEST_FAIL_CHUNK): a failure mid-pipeline
            c->fault_chunk = -1;
            return fail(TFHE_HIP_ERR_HIP, "injected failure (TFHE_HIP_TEST_FAIL_CHUNK)");
        }
        if (int r = tfhe_hip_gate_batch_dev(c, gate, (int32_t *) c->outb.p + o, a ? (const int32_t *) c->in_a.p + o : nullptr,
                                            b ? (const int32_t *) c->in_b.p + o : nullptr,
                                            cc ? (const int32_t *) c->in_c.p + o : nullptr, n))
            return r;
        STEP(hipEventRecord(c->chunk_ev[3 * k + 1], c->stream));
        return TFHE_HIP_OK;
    };
#undef STEP
    int rc = copy_in(0);
    for (int64_t k = 0; k < nchunks && rc == TFHE_HIP_OK; k++) {
        rc = run_chunk(k);
        if (rc == TFHE_HIP_OK && k + 1 < nchunks) rc = copy_in(k + 1);
        if (rc == TFHE_HIP_OK && k > 0) rc = copy_out(k - 1);
        if (rc == TFHE_HIP_OK && k > 0) rc = drain_chunk(k - 1);
    }
    if (rc == TFHE_HIP_OK) rc = copy_out(nchunks - 1);
    const std::string first_error = rc == TFHE_HIP_OK ? std::string() : g_err;
    (void) hipStreamSynchronize(c->copy_stream);  // nothing targets the caller's buffers once these two return
    (void) hipStreamSynchronize(c->stream);
    if (rc == TFHE_HIP_OK) rc = drain_chunk(nchunks - 1);
    else g_err = first_error;
    return rc;
}

extern "C" {

int tfhe_hip_bootstrap_batch(tfhe_hip_ctx *c, int32_t *out, const int32_t *in, int32_t mu, int64_t count, int with_ks) {
    if (int r = check_ctx(c, true)) return r;
    if (count < 0 || (count > 0 && (!out || !in))) return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    if (count == 0) return TFHE_HIP_OK;
    const size_t ib = (size_t) count * kCt * sizeof(int32_t);
    const size_t ob = (size_t) count * (with_ks ? kCt : kExt) * sizeof(int32_t);
    if (int r = h2d(c, c->in_a, in, ib)) return r;
    HIPCHK(c->outb.reserve(ob));
    if (int r = tfhe_hip_bootstrap_batch_dev(c, (int32_t *) c->outb.p, (const int32_t *) c->in_a.p, mu, count, with_ks))
        return r;
    return d2h(c, out, c->outb.p, ob);
}

int tfhe_hip_keyswitch_batch(tfhe_hip_ctx *c, int32_t *out, const int32_t *in, int64_t count) {
    if (int r = check_ctx(c, true)) return r;
    if (count < 0 || (count > 0 && (!out || !in))) return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    if (count == 0) return TFHE_HIP_OK;
    const size_t ib = (size_t) count * kExt * sizeof(int32_t), ob = (size_t) count * kCt * sizeof(int32_t);
    if (int r = h2d(c, c->in_a, in, ib)) return r;
    HIPCHK(c->outb.reserve(ob));
    if (int r = tfhe_hip_keyswitch_batch_dev(c, (int32_t *) c->outb.p, (const int32_t *) c->in_a.p, count)) return r;
    return d2h(c, out, c->outb.p, ob);
}

int tfhe_hip_tgsw_decomp_batch(tfhe_hip_ctx *c, int32_t *out, const int32_t *polys, int64_t count) {
    if (int r = check_ctx(c, false)) return r;
    if (count < 0 || (count > 0 && (!out || !polys))) return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    if (count == 0) return TFHE_HIP_OK;
    const size_t ib = (size_t) count * kN * sizeof(int32_t), ob = ib * kL;
    if (int r = h2d(c, c->in_a, polys, ib)) return r;
    HIPCHK(c->outb.reserve(ob));
    const long long total = (long long) count * kN;
    hipLaunchKernelGGL(k_decomp, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, c->stream, (int32_t *) c->outb.p,
                       (const int32_t *) c->in_a.p, total);
    HIPCHK(hipGetLastError());
    return d2h(c, out, c->outb.p, ob);
}

static int raw_br(tfhe_hip_ctx *c, int mode, int32_t *acc, const int32_t *aux, size_t aux_elems, int64_t count) {
    if (int r = check_ctx(c, true)) return r;
    if (count < 0 || (count > 0 && (!acc || !aux))) return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    if (count == 0) return TFHE_HIP_OK;
    if (mode == MODE_EXTMUL)
        for (int64_t i = 0; i < count; i++)
            if (aux[i] < 0 || aux[i] >= kLwe) return fail(TFHE_HIP_ERR_ARG, "key index out of range");
    const size_t ab = (size_t) count * 2 * kN * sizeof(int32_t);
    if (int r = h2d(c, c->in_a, acc, ab)) return r;
    if (int r = h2d(c, c->in_b, aux, aux_elems * sizeof(int32_t))) return r;
    BrArgs A{};
    A.acc_io = (int32_t *) c->in_a.p; A.bara = (const int32_t *) c->in_b.p; A.bk = c->bk; A.tw = c->tw; A.ka = c->ka; A.B = count;
    c->ev_br = c->ev_ks = false;
    HIPCHK(hipEventRecord(c->ev[0], c->stream));
    if (int r = (mode == MODE_RAW ? launch_br<MODE_RAW>(c, A) : launch_br<MODE_EXTMUL>(c, A))) return r;
    HIPCHK(hipEventRecord(c->ev[1], c->stream));
    c->ev_br = true;
    return d2h(c, acc, c->in_a.p, ab);
}
int tfhe_hip_blind_rotate_batch(tfhe_hip_ctx *c, int32_t *acc, const int32_t *bara, int64_t count) {
    return raw_br(c, MODE_RAW, acc, bara, (size_t) (count > 0 ? count : 0) * kLwe, count);
}
int tfhe_hip_extern_mul_batch(tfhe_hip_ctx *c, int32_t *acc, const int32_t *key_index, int64_t count) {
    return raw_br(c, MODE_EXTMUL, acc, key_index, (size_t) (count > 0 ? count : 0), count);
}

static int fwd_batch(tfhe_hip_ctx *c, double *spectra, const int32_t *polys, int64_t count, int torus) {
    if (int r = check_ctx(c, false)) return r;
    if (count < 0 || (count > 0 && (!spectra || !polys))) return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    if (count == 0) return TFHE_HIP_OK;
    const size_t pb = (size_t) count * kN * sizeof(int32_t), sb = (size_t) count * kNh * sizeof(c64);
    if (int r = h2d(c, c->in_a, polys, pb)) return r;
    HIPCHK(c->outb.reserve(sb));
    if (c->arith == TFHE_HIP_ARITH_FAST) {
        HIPCHK(hipFuncSetAttribute((const void *) k_poly_fwd_fast, hipFuncAttributeMaxDynamicSharedMemorySize, kFftLdsFast));
        hipLaunchKernelGGL(k_poly_fwd_fast, dim3((unsigned) ((count + 3) / 4)), dim3(256), kFftLdsFast, c->stream,
                           (const int32_t *) c->in_a.p, (c64 *) c->outb.p, (const c64 *) c->twf, c->ka, (long long) count, torus);
    } else {
        HIPCHK(hipFuncSetAttribute((const void *) k_poly_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, kFftLds));
        hipLaunchKernelGGL(k_poly_fwd, dim3((unsigned) ((count + 3) / 4)), dim3(256), kFftLds, c->stream,
                           (const int32_t *) c->in_a.p, (c64 *) c->outb.p, (const c64 *) c->tw, c->ka, (long long) count, torus);
    }
    HIPCHK(hipGetLastError());
    return d2h(c, spectra, c->outb.p, sb);
}
int tfhe_hip_int_ifft_batch(tfhe_hip_ctx *c, double *spectra, const int32_t *polys, int64_t count) {
    return fwd_batch(c, spectra, polys, count, 0);
}
int tfhe_hip_torus_ifft_batch(tfhe_hip_ctx *c, double *spectra, const int32_t *polys, int64_t count) {
    return fwd_batch(c, spectra, polys, count, 1);
}
int tfhe_hip_torus_fft_batch(tfhe_hip_ctx *c, int32_t *polys, const double *spectra, int64_t count) {
    if (int r = check_ctx(c, false)) return r;
    if (count < 0 || (count > 0 && (!spectra || !polys))) return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    if (count == 0) return TFHE_HIP_OK;
    const size_t pb = (size_t) count * kN * sizeof(int32_t), sb = (size_t) count * kNh * sizeof(c64);
    if (int r = h2d(c, c->in_a, spectra, sb)) return r;
    HIPCHK(c->outb.reserve(pb));
    if (c->arith == TFHE_HIP_ARITH_FAST) {
        HIPCHK(hipFuncSetAttribute((const void *) k_poly_inv_fast, hipFuncAttributeMaxDynamicSharedMemorySize, kFftLdsFast));
        hipLaunchKernelGGL(k_poly_inv_fast, dim3((unsigned) ((count + 3) / 4)), dim3(256), kFftLdsFast, c->stream,
                           (const c64 *) c->in_a.p, (int32_t *) c->outb.p, (const c64 *) c->twf, c->ka, (long long) count);
    } else {
        HIPCHK(hipFuncSetAttribute((const void *) k_poly_inv, hipFuncAttributeMaxDynamicSharedMemorySize, kFftLds));
        hipLaunchKernelGGL(k_poly_inv, dim3((unsigned) ((count + 3) / 4)), dim3(256), kFftLds, c->stream,
                           (const c64 *) c->in_a.p, (int32_t *) c->outb.p, (const c64 *) c->tw, c->ka, (long long) count);
    }
    HIPCHK(hipGetLastError());
    return d2h(c, polys, c->outb.p, pb);
}
int tfhe_hip_negacyclic_mul_batch_dev(tfhe_hip_ctx *c, int32_t *d_out, const int32_t *d_i, const int32_t *d_t,
                                      int64_t count) {
    if (int r = check_ctx(c, false)) return r;
    if (count < 0 || (count > 0 && (!d_out || !d_i || !d_t))) return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    if (count == 0) return TFHE_HIP_OK;
    const bool timed = !c->in_mul2048;  // tfhe_hip_last_kernel_ms: events around this call's kernel (the N = 2048 call brackets its own)
    if (timed) {
        c->ev_br = c->ev_ks = false;
        HIPCHK(hipEventRecord(c->ev[0], c->stream));
    }
    if (c->arith == TFHE_HIP_ARITH_FAST) {
        HIPCHK(hipFuncSetAttribute((const void *) k_poly_mul_fast, hipFuncAttributeMaxDynamicSharedMemorySize, kFftLdsFast));
        hipLaunchKernelGGL(k_poly_mul_fast, dim3((unsigned) std::min<long long>((count + 3) / 4, kPolyMulMaxBlocks)), dim3(256), kFftLdsFast, c->stream, d_i, d_t,
                           d_out, (const c64 *) c->twf, c->ka, (long long) count);
    } else {
        HIPCHK(hipFuncSetAttribute((const void *) k_poly_mul, hipFuncAttributeMaxDynamicSharedMemorySize, kFftLds));
        hipLaunchKernelGGL(k_poly_mul, dim3((unsigned) std::min<long long>((count + 3) / 4, kPolyMulMaxBlocks)), dim3(256), kFftLds, c->stream, d_i, d_t, d_out,
                           (const c64 *) c->tw, c->ka, (long long) count);
    }
    HIPCHK(hipGetLastError());
    if (timed) {
        HIPCHK(hipEventRecord(c->ev[1], c->stream));
        c->ev_br = true;
    }
    return TFHE_HIP_OK;
}
int tfhe_hip_negacyclic_mul_batch(tfhe_hip_ctx *c, int32_t *out, const int32_t *ip, const int32_t *tp, int64_t count) {
    if (int r = check_ctx(c, false)) return r;
    if (count < 0 || (count > 0 && (!out || !ip || !tp))) return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    if (count == 0) return TFHE_HIP_OK;
    const size_t pb = (size_t) count * kN * sizeof(int32_t);
    if (int r = h2d(c, c->in_a, ip, pb)) return r;
    if (int r = h2d(c, c->in_b, tp, pb)) return r;
    HIPCHK(c->outb.reserve(pb));
    if (int r = tfhe_hip_negacyclic_mul_batch_dev(c, (int32_t *) c->outb.p, (const int32_t *) c->in_a.p,
                                                  (const int32_t *) c->in_b.p, count))
        return r;
    return d2h(c, out, c->outb.p, pb);
}

int tfhe_hip_negacyclic_mul2048_batch_dev(tfhe_hip_ctx *c, int32_t *d_out, const int32_t *d_i, const int32_t *d_t,
                                          int64_t count) {
    if (int r = check_ctx(c, false)) return r;
    if (count < 0 || (count > 0 && (!d_out || !d_i || !d_t))) return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    if (c->arith == TFHE_HIP_ARITH_FAST) {  // one fused kernel: four forward, two inverse transforms per product
        if (count == 0) return TFHE_HIP_OK;
        if ((((uintptr_t) d_out) | ((uintptr_t) d_i) | ((uintptr_t) d_t)) & 7)
            return fail(TFHE_HIP_ERR_ARG, "N = 2048 polynomials must be 8-byte aligned");
        c->ev_br = c->ev_ks = false;
        HIPCHK(hipEventRecord(c->ev[0], c->stream));
        HIPCHK(hipFuncSetAttribute((const void *) k_poly_mul2048_fast, hipFuncAttributeMaxDynamicSharedMemorySize, kFftLdsFast));
        hipLaunchKernelGGL(k_poly_mul2048_fast, dim3((unsigned) std::min<long long>((count + 3) / 4, kPolyMulMaxBlocks)), dim3(256),
                           kFftLdsFast, c->stream, d_i, d_t, d_out, (const c64 *) c->twf, c->ka, (long long) count);
        HIPCHK(hipGetLastError());
        HIPCHK(hipEventRecord(c->ev[1], c->stream));
        c->ev_br = true;
        return TFHE_HIP_OK;
    }
    constexpr int64_t kChunk = 8192;  // workspace: 3 x 4 x 4 KB per product
    const int64_t first = count < kChunk ? count : kChunk;
    if (count > 0) HIPCHK(c->misc.reserve((size_t) first * 12 * kN * sizeof(int32_t)));
    c->ev_br = c->ev_ks = false;
    if (count > 0) HIPCHK(hipEventRecord(c->ev[0], c->stream));
    struct Flag {
        bool &f;
        ~Flag() { f = false; }
    } flag{c->in_mul2048};
    c->in_mul2048 = true;
    for (int64_t off = 0; off < count; off += kChunk) {
        const int64_t n = count - off < kChunk ? count - off : kChunk;
        int32_t *I = (int32_t *) c->misc.p, *T = I + (size_t) first * 4 * kN, *P = T + (size_t) first * 4 * kN;
        const unsigned blocks = (unsigned) ((n * kN + 255) / 256);
        hipLaunchKernelGGL(k_split2048, dim3(blocks), dim3(256), 0, c->stream, d_i + off * 2 * kN, d_t + off * 2 * kN, I, T,
                           (long long) n);
        HIPCHK(hipGetLastError());
        if (int r = tfhe_hip_negacyclic_mul_batch_dev(c, P, I, T, 4 * n)) return r;
        hipLaunchKernelGGL(k_combine2048, dim3(blocks), dim3(256), 0, c->stream, (const int32_t *) P, d_out + off * 2 * kN,
                           (long long) n);
        HIPCHK(hipGetLastError());
    }
    if (count > 0) {
        HIPCHK(hipEventRecord(c->ev[1], c->stream));
        c->ev_br = true;
    }
    return TFHE_HIP_OK;
}
int tfhe_hip_negacyclic_mul2048_batch(tfhe_hip_ctx *c, int32_t *out, const int32_t *ip, const int32_t *tp, int64_t count) {
    if (int r = check_ctx(c, false)) return r;
    if (count < 0 || (count > 0 && (!out || !ip || !tp))) return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    if (count == 0) return TFHE_HIP_OK;
    const size_t pb = (size_t) count * 2 * kN * sizeof(int32_t);
    if (int r = h2d(c, c->in_a, ip, pb)) return r;
    if (int r = h2d(c, c->in_b, tp, pb)) return r;
    HIPCHK(c->outb.reserve(pb));
    if (int r = tfhe_hip_negacyclic_mul2048_batch_dev(c, (int32_t *) c->outb.p, (const int32_t *) c->in_a.p,
                                                      (const int32_t *) c->in_b.p, count))
        return r;
    return d2h(c, out, c->outb.p, pb);
}

// ---- leveled operations on caller-supplied operands -------------------------------------------------
static int check_index(const int32_t *index, int64_t count, int64_t n) {
    for (int64_t i = 0; i < count; i++)
        if (index[i] < 0 || index[i] >= n) return fail(TFHE_HIP_ERR_ARG, "operand index out of range");
    return TFHE_HIP_OK;
}
// tGswFFTExternMulToTLwe (tgsw-fft-operations.cpp:92-120) with TGswSampleFFT operands of the caller's
int tfhe_hip_tgsw_fft_extern_mul_batch(tfhe_hip_ctx *c, int32_t *acc, const double *gsw_fft, const int32_t *gsw_index,
                                       int64_t ngsw, int64_t count) {
    if (int r = check_ctx(c, false)) return r;
    if (count < 0 || ngsw < 0 || (count > 0 && (!acc || !gsw_fft || !gsw_index || ngsw == 0)))
        return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    if (count == 0) return TFHE_HIP_OK;
    if (int r = check_index(gsw_index, count, ngsw)) return r;
    // the kernel addresses the operand array through a buffer resource with 32-bit byte offsets
    if ((unsigned long long) ngsw * kKpl * 2 * kNh * sizeof(c64) > 0x7ffffff0ull)
        return fail(TFHE_HIP_ERR_ARG, "more than 2 GiB of TGSW operands in one call: split it");
    const size_t op_bytes = (size_t) ngsw * kKpl * 2 * kNh * sizeof(c64), ab = (size_t) count * 2 * kN * sizeof(int32_t);
    const long long total = (long long) ngsw * kKpl * 2 * kNh;
    const bool fast = c->arith == TFHE_HIP_ARITH_FAST;
    // operand spectra: reference order -> physical slot order (in_c) -> (FAST) Z_k order (outb)
    if (int r = h2d(c, c->misc, gsw_fft, op_bytes)) return r;
    HIPCHK(c->in_c.reserve(op_bytes));
    hipLaunchKernelGGL(k_bk_permute, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, c->stream, (c64 *) c->in_c.p,
                       (const c64 *) c->misc.p, total, 1);
    const c64 *op = (const c64 *) c->in_c.p;
    if (fast) {
        HIPCHK(c->outb.reserve(op_bytes));
        hipLaunchKernelGGL(k_bk_to_fast, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, c->stream, (c64 *) c->outb.p,
                           (const c64 *) c->in_c.p, total);
        op = (const c64 *) c->outb.p;
    }
    HIPCHK(hipGetLastError());
    if (int r = h2d(c, c->in_a, acc, ab)) return r;
    if (int r = h2d(c, c->in_b, gsw_index, (size_t) count * sizeof(int32_t))) return r;
    BrArgs A{};
    A.acc_io = (int32_t *) c->in_a.p; A.bara = (const int32_t *) c->in_b.p; A.tw = c->tw; A.ka = c->ka; A.B = count;
    c->ev_br = c->ev_ks = false;
    HIPCHK(hipEventRecord(c->ev[0], c->stream));
    if (int r = launch_br<MODE_EXTMUL>(c, A, op)) return r;
    HIPCHK(hipEventRecord(c->ev[1], c->stream));
    c->ev_br = true;
    return d2h(c, acc, c->in_a.p, ab);
}
// products of the operand lists I, T (count2 = number of products) into P, in the arithmetic of the context
static int mul_lists(tfhe_hip_ctx *c, int32_t *P, const int32_t *I, const int32_t *T, int64_t count2) {
    return tfhe_hip_negacyclic_mul_batch_dev(c, P, I, T, count2);
}
// tGswExternMulToTLwe (tgsw-functions.cpp:157-170): acc[g] <- tgsw[index[g]] (*) acc[g], coefficient-domain operands
int tfhe_hip_tgsw_extern_mul_batch(tfhe_hip_ctx *c, int32_t *acc, const int32_t *tgsw, const int32_t *tgsw_index,
                                   int64_t ntgsw, int64_t count) {
    if (int r = check_ctx(c, false)) return r;
    if (count < 0 || ntgsw < 0 || (count > 0 && (!acc || !tgsw || !tgsw_index || ntgsw == 0)))
        return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    if (count == 0) return TFHE_HIP_OK;
    if (int r = check_index(tgsw_index, count, ntgsw)) return r;
    const size_t ab = (size_t) count * 2 * kN * sizeof(int32_t), list = (size_t) count * kKpl * 2 * kN * sizeof(int32_t);
    if (int r = h2d(c, c->in_a, acc, ab)) return r;
    if (int r = h2d(c, c->in_b, tgsw, (size_t) ntgsw * kKpl * 2 * kN * sizeof(int32_t))) return r;
    if (int r = h2d(c, c->in_c, tgsw_index, (size_t) count * sizeof(int32_t))) return r;
    HIPCHK(c->misc.reserve(3 * list));
    int32_t *I = (int32_t *) c->misc.p, *T = I + list / 4, *P = T + list / 4;
    const long long n1 = (long long) count * kKpl * kN;
    hipLaunchKernelGGL(k_tgsw_prepare, dim3((unsigned) ((n1 + 255) / 256)), dim3(256), 0, c->stream, (const int32_t *) c->in_a.p,
                       (const int32_t *) c->in_b.p, (const int32_t *) c->in_c.p, I, T, (long long) count);
    HIPCHK(hipGetLastError());
    if (int r = mul_lists(c, P, I, T, count * kKpl * 2)) return r;
    const long long n2 = (long long) count * 2 * kN;
    hipLaunchKernelGGL(k_rows_sum<false>, dim3((unsigned) ((n2 + 255) / 256)), dim3(256), 0, c->stream, (int32_t *) c->in_a.p,
                       (const int32_t *) P, kKpl, (long long) count);  // tLweClear, then the rows in order
    HIPCHK(hipGetLastError());
    return d2h(c, acc, c->in_a.p, ab);
}
// tLweAddMulRTo (tlwe-functions.cpp:191-197): result[g] += p[g] * sample[g]
int tfhe_hip_tlwe_addmul_r_batch(tfhe_hip_ctx *c, int32_t *result, const int32_t *poly, const int32_t *sample, int64_t count) {
    if (int r = check_ctx(c, false)) return r;
    if (count < 0 || (count > 0 && (!result || !poly || !sample))) return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    if (count == 0) return TFHE_HIP_OK;
    const size_t ab = (size_t) count * 2 * kN * sizeof(int32_t);
    if (int r = h2d(c, c->in_a, result, ab)) return r;
    if (int r = h2d(c, c->in_b, sample, ab)) return r;
    if (int r = h2d(c, c->in_c, poly, ab / 2)) return r;
    HIPCHK(c->misc.reserve(2 * ab));
    int32_t *I = (int32_t *) c->misc.p, *P = I + ab / 4;
    const long long n1 = (long long) count * kN, n2 = (long long) count * 2 * kN;
    hipLaunchKernelGGL(k_addmul_prepare, dim3((unsigned) ((n1 + 255) / 256)), dim3(256), 0, c->stream, (const int32_t *) c->in_c.p, I,
                       (long long) count);
    HIPCHK(hipGetLastError());
    if (int r = mul_lists(c, P, I, (const int32_t *) c->in_b.p, count * 2)) return r;
    hipLaunchKernelGGL(k_rows_sum<true>, dim3((unsigned) ((n2 + 255) / 256)), dim3(256), 0, c->stream, (int32_t *) c->in_a.p,
                       (const int32_t *) P, 1, (long long) count);
    HIPCHK(hipGetLastError());
    return d2h(c, result, c->in_a.p, ab);
}

// ---- client-side bulk operations --------------------------------------------------------------
int tfhe_hip_encrypt_batch_dev(tfhe_hip_ctx *c, int32_t *d_out, uint64_t seed, int64_t first_index,
                               const int32_t *lwe_key, const uint8_t *bits, int64_t count) {
    if (int r = check_ctx(c, false)) return r;
    if (count < 0 || first_index < 0 || (count > 0 && (!d_out || !lwe_key || !bits)))
        return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    if (count == 0) return TFHE_HIP_OK;
    if (int r = ensure_cdt(c)) return r;
    HIPCHK(c->misc.reserve(kLwe * sizeof(int32_t) + (size_t) count));
    int32_t *d_key = (int32_t *) c->misc.p;
    uint8_t *d_bits = (uint8_t *) (d_key + kLwe);
    DeviceWipe wipe{c};  // the secret key (and the plaintext bits) leave device memory when this call returns
    wipe.arm(d_key, kLwe * sizeof(int32_t) + (size_t) count);
    HIPCHK(hipMemcpyAsync(d_key, lwe_key, kLwe * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_bits, bits, (size_t) count, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_client_encrypt, dim3((unsigned) ((count + 63) / 64)), dim3(64), 0, c->stream, seed,
                       (const int32_t *) d_key, (const uint8_t *) d_bits, (long long) first_index, (long long) count, d_out,
                       (const uint64_t *) c->cdt);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));  // misc is reused by other calls
    return TFHE_HIP_OK;
}
int tfhe_hip_encrypt_batch(tfhe_hip_ctx *c, int32_t *out, uint64_t seed, int64_t first_index, const int32_t *lwe_key,
                           const uint8_t *bits, int64_t count) {
    if (int r = check_ctx(c, false)) return r;
    if (count < 0 || (count > 0 && !out)) return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    if (count == 0) return TFHE_HIP_OK;
    const size_t ob = (size_t) count * kCt * sizeof(int32_t);
    HIPCHK(c->outb.reserve(ob));
    if (int r = tfhe_hip_encrypt_batch_dev(c, (int32_t *) c->outb.p, seed, first_index, lwe_key, bits, count)) return r;
    return d2h(c, out, c->outb.p, ob);
}
int tfhe_hip_phase_batch(tfhe_hip_ctx *c, int32_t *phases, const int32_t *lwe_key, const int32_t *samples, int64_t count) {
    if (int r = check_ctx(c, false)) return r;
    if (count < 0 || (count > 0 && (!phases || !lwe_key || !samples))) return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    if (count == 0) return TFHE_HIP_OK;
    const size_t ib = (size_t) count * kCt * sizeof(int32_t);
    if (int r = h2d(c, c->in_a, samples, ib)) return r;
    HIPCHK(c->misc.reserve(kLwe * sizeof(int32_t)));
    DeviceWipe wipe{c};
    wipe.arm(c->misc.p, kLwe * sizeof(int32_t));
    HIPCHK(hipMemcpyAsync(c->misc.p, lwe_key, kLwe * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c->outb.reserve((size_t) count * sizeof(int32_t)));
    hipLaunchKernelGGL(k_client_phase, dim3((unsigned) ((count + 3) / 4)), dim3(256), 0, c->stream,
                       (const int32_t *) c->misc.p, (const int32_t *) c->in_a.p, (long long) count, (int32_t *) c->outb.p);
    HIPCHK(hipGetLastError());
    return d2h(c, phases, c->outb.p, (size_t) count * sizeof(int32_t));
}

// ---- netlist level on a device-resident ciphertext pool ---------------------------------------
// validates one level against the level contract and expands it into bootstrap / keyswitch / linear jobs
// (c->h_br, c->h_ks, c->h_lin)
static int expand_level(tfhe_hip_ctx *c, int64_t pool_rows, const int32_t *gates, int64_t count) {
    // validation: indices in range; a level reads the pool as it was before the call, so no row may be both
    // written and read (or written twice) within one level
    c->h_mark.assign((size_t) pool_rows, 0);
    for (int64_t g = 0; g < count; g++) {
        const int32_t *d = gates + 5 * g;
        const int gate = d[0];
        if (gate < TFHE_HIP_NAND || gate > TFHE_HIP_CONST0) return fail(TFHE_HIP_ERR_ARG, "unknown gate id in level");
        const int nin = gate <= TFHE_HIP_ORYN ? 2 : gate == TFHE_HIP_MUX ? 3 : (gate <= TFHE_HIP_COPY ? 1 : 0);
        if (d[1] < 0 || d[1] >= pool_rows) return fail(TFHE_HIP_ERR_ARG, "output row out of range");
        for (int k = 0; k < nin; k++)
            if (d[2 + k] < 0 || d[2 + k] >= pool_rows) return fail(TFHE_HIP_ERR_ARG, "input row out of range");
        if (c->h_mark[d[1]] & 1) return fail(TFHE_HIP_ERR_ARG, "row written twice in one level");
        c->h_mark[d[1]] |= 1;
    }
    for (int64_t g = 0; g < count; g++) {
        const int32_t *d = gates + 5 * g;
        const int nin = d[0] <= TFHE_HIP_ORYN ? 2 : d[0] == TFHE_HIP_MUX ? 3 : (d[0] <= TFHE_HIP_COPY ? 1 : 0);
        for (int k = 0; k < nin; k++)
            if (c->h_mark[d[2 + k]] & 1) return fail(TFHE_HIP_ERR_ARG, "row read and written in the same level");
    }
    c->h_br.clear(); c->h_ks.clear(); c->h_lin.clear();
    for (int64_t g = 0; g < count; g++) {
        const int32_t *d = gates + 5 * g;
        GateLin gl;
        if (gate_lin(d[0], gl)) {
            c->h_ks.push_back(KsJob{(int32_t) c->h_br.size(), -1, 0, d[1]});
            c->h_br.push_back(BrJob{d[2], d[3], gl.sa, gl.sb, gl.cst, 0, 0, 0});
        } else if (d[0] == TFHE_HIP_MUX) {  // boot-gates.cpp:329-370
            c->h_ks.push_back(KsJob{(int32_t) c->h_br.size(), (int32_t) c->h_br.size() + 1, kMU, d[1]});
            c->h_br.push_back(BrJob{d[2], d[3], 1, 1, -kMU, 0, 0, 0});
            c->h_br.push_back(BrJob{d[2], d[4], -1, 1, -kMU, 0, 0, 0});
        } else {
            c->h_lin.push_back(d[0]); c->h_lin.push_back(d[1]); c->h_lin.push_back(d[2]);
        }
    }
    return TFHE_HIP_OK;
}

int tfhe_hip_gate_level_dev(tfhe_hip_ctx *c, int32_t *d_pool, int64_t pool_rows, const int32_t *gates, int64_t count) {
    if (int r = check_ctx(c, true)) return r;
    if (count < 0 || pool_rows < 0 || (count > 0 && (!d_pool || !gates))) return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    c->ev_br = c->ev_ks = false;
    if (count == 0) return TFHE_HIP_OK;
    if (int r = expand_level(c, pool_rows, gates, count)) return r;
    const long long nbr = (long long) c->h_br.size(), nks = (long long) c->h_ks.size(), nlin = (long long) c->h_lin.size() / 3;
    if (nlin) {
        HIPCHK(c->jobs_lin.reserve(c->h_lin.size() * sizeof(int32_t)));
        HIPCHK(hipMemcpyAsync(c->jobs_lin.p, c->h_lin.data(), c->h_lin.size() * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(k_linear_level, dim3((unsigned) nlin), dim3(256), 0, c->stream, d_pool,
                           (const int32_t *) c->jobs_lin.p, kMU, nlin);
        HIPCHK(hipGetLastError());
    }
    if (nbr) {
        HIPCHK(c->jobs_br.reserve((size_t) nbr * sizeof(BrJob)));
        HIPCHK(c->jobs_ks.reserve((size_t) nks * sizeof(KsJob)));
        HIPCHK(hipMemcpyAsync(c->jobs_br.p, c->h_br.data(), (size_t) nbr * sizeof(BrJob), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipMemcpyAsync(c->jobs_ks.p, c->h_ks.data(), (size_t) nks * sizeof(KsJob), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c->ext1.reserve((size_t) nbr * kExt * sizeof(int32_t)));
        BrArgs A{};
        A.jobs = (const BrJob *) c->jobs_br.p; A.a = d_pool; A.mu = kMU; A.out = (int32_t *) c->ext1.p;
        A.bk = c->bk; A.tw = c->tw; A.ka = c->ka; A.B = nbr;
        HIPCHK(hipEventRecord(c->ev[0], c->stream));
        if (int r = launch_br<MODE_GATE>(c, A)) return r;
        HIPCHK(hipEventRecord(c->ev[1], c->stream));
        c->ev_br = true;
        HIPCHK(hipEventRecord(c->ev[2], c->stream));
        if (int r = launch_ks(c, d_pool, (const int32_t *) c->ext1.p, nks, (const KsJob *) c->jobs_ks.p)) return r;
        HIPCHK(hipEventRecord(c->ev[3], c->stream));
        c->ev_ks = true;
    }
    // the job lists live in pageable host vectors that the next call overwrites: wait for the copies (and the level)
    HIPCHK(hipStreamSynchronize(c->stream));
    return TFHE_HIP_OK;
}

// ---- compiled netlist: all levels of a circuit resident on the device, replayed as one hipGraph -------------------
struct tfhe_hip_netlist {
    tfhe_hip_ctx *c = nullptr;
    int32_t *d_pool = nullptr;
    struct Level {
        long long nbr, nks, nlin;
        size_t obr, oks, olin;  // element offsets into the job arrays
    };
    std::vector<Level> levels;
    DevBuf jobs_br, jobs_ks, jobs_lin, ext, abart, ks_part;  // job arrays and the workspace the graph's kernels use
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

// every launch of the circuit, in level order, on the context's stream; no host synchronisation in between
static int netlist_enqueue(tfhe_hip_netlist *nl) {
    tfhe_hip_ctx *c = nl->c;
    std::swap(c->abart, nl->abart);  // launch_ks takes its workspace from the context: lend it the netlist's
    std::swap(c->ks_part, nl->ks_part);
    int r = TFHE_HIP_OK;
    for (const tfhe_hip_netlist::Level &lv : nl->levels) {
        if (lv.nlin) {
            hipLaunchKernelGGL(k_linear_level, dim3((unsigned) lv.nlin), dim3(256), 0, c->stream, nl->d_pool,
                               (const int32_t *) nl->jobs_lin.p + lv.olin, kMU, lv.nlin);
        }
        if (lv.nbr) {
            BrArgs A{};
            A.jobs = (const BrJob *) nl->jobs_br.p + lv.obr; A.a = nl->d_pool; A.mu = kMU; A.out = (int32_t *) nl->ext.p;
            A.bk = c->bk; A.tw = c->tw; A.ka = c->ka; A.B = lv.nbr;
            if ((r = launch_br<MODE_GATE>(c, A))) break;
            if ((r = launch_ks(c, nl->d_pool, (const int32_t *) nl->ext.p, lv.nks, (const KsJob *) nl->jobs_ks.p + lv.oks))) break;
        }
    }
    std::swap(c->abart, nl->abart);
    std::swap(c->ks_part, nl->ks_part);
    return r;
}

void tfhe_hip_netlist_destroy(tfhe_hip_netlist *nl) {
    if (!nl) return;
    if (nl->c) {
        (void) hipSetDevice(nl->c->device);
        (void) hipStreamSynchronize(nl->c->stream);
    }
    if (nl->exec) (void) hipGraphExecDestroy(nl->exec);
    if (nl->graph) (void) hipGraphDestroy(nl->graph);
    nl->jobs_br.release(); nl->jobs_ks.release(); nl->jobs_lin.release();
    nl->ext.release(); nl->abart.release(); nl->ks_part.release();
    delete nl;
}

int tfhe_hip_netlist_create(tfhe_hip_ctx *c, tfhe_hip_netlist **out, int32_t *d_pool, int64_t pool_rows,
                            const int32_t *gates, const int64_t *level_counts, int64_t nlevels) {
    if (int r = check_ctx(c, true)) return r;
    if (!out) return fail(TFHE_HIP_ERR_ARG, "null output");
    *out = nullptr;
    if (nlevels < 0 || pool_rows < 0 || (nlevels > 0 && (!d_pool || !gates || !level_counts)))
        return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    std::unique_ptr<tfhe_hip_netlist, void (*)(tfhe_hip_netlist *)> nl(new tfhe_hip_netlist, tfhe_hip_netlist_destroy);
    nl->c = c;
    nl->d_pool = d_pool;
    std::vector<BrJob> br;
    std::vector<KsJob> ks;
    std::vector<int32_t> lin;
    size_t ext_b = 0, abart_b = 0, part_b = 0;
    const int32_t *g = gates;
    for (int64_t l = 0; l < nlevels; l++) {
        if (level_counts[l] < 0) return fail(TFHE_HIP_ERR_ARG, "negative level size");
        if (int r = expand_level(c, pool_rows, g, level_counts[l])) return r;
        g += 5 * level_counts[l];
        tfhe_hip_netlist::Level lv{(long long) c->h_br.size(), (long long) c->h_ks.size(), (long long) c->h_lin.size() / 3,
                                   br.size(), ks.size(), lin.size()};
        br.insert(br.end(), c->h_br.begin(), c->h_br.end());
        ks.insert(ks.end(), c->h_ks.begin(), c->h_ks.end());
        lin.insert(lin.end(), c->h_lin.begin(), c->h_lin.end());
        if (lv.nbr) {  // workspace of this level, as launch_ks sizes it
            const long long Bp = (lv.nks + 32 * kKsMT - 1) / (32 * kKsMT) * (32 * kKsMT);
            const int S = ks_split(Bp);
            ext_b = std::max(ext_b, (size_t) lv.nbr * kExt * sizeof(int32_t));
            abart_b = std::max(abart_b, (size_t) ((size_t) kN * (size_t) Bp * sizeof(uint32_t)));
            if (S > 1) part_b = std::max(part_b, (size_t) ((size_t) S * (size_t) Bp * kCt * sizeof(uint32_t)));
        }
        nl->levels.push_back(lv);
    }
    auto upload = [&](DevBuf &d, const void *src, size_t bytes) -> hipError_t {
        if (!bytes) return hipSuccess;
        hipError_t e = d.reserve(bytes);
        return e == hipSuccess ? hipMemcpy(d.p, src, bytes, hipMemcpyHostToDevice) : e;
    };
    HIPCHK(upload(nl->jobs_br, br.data(), br.size() * sizeof(BrJob)));
    HIPCHK(upload(nl->jobs_ks, ks.data(), ks.size() * sizeof(KsJob)));
    HIPCHK(upload(nl->jobs_lin, lin.data(), lin.size() * sizeof(int32_t)));
    if (ext_b) HIPCHK(nl->ext.reserve(ext_b));
    if (abart_b) HIPCHK(nl->abart.reserve(abart_b));
    if (part_b) HIPCHK(nl->ks_part.reserve(part_b));
    // record the launches once; if this runtime cannot capture them, tfhe_hip_netlist_run enqueues them one by one
    HIPCHK(hipStreamSynchronize(c->stream));
    if (!nl->levels.empty() && hipStreamBeginCapture(c->stream, hipStreamCaptureModeRelaxed) == hipSuccess) {
        const int r = netlist_enqueue(nl.get());
        hipGraph_t graph = nullptr;
        const hipError_t e = hipStreamEndCapture(c->stream, &graph);
        if (r == TFHE_HIP_OK && e == hipSuccess && graph &&
            hipGraphInstantiate(&nl->exec, graph, nullptr, nullptr, 0) == hipSuccess) {
            nl->graph = graph;
        } else {
            if (graph) (void) hipGraphDestroy(graph);
            nl->exec = nullptr;
            (void) hipGetLastError();
        }
    }
    *out = nl.release();
    return TFHE_HIP_OK;
}

int tfhe_hip_netlist_run(tfhe_hip_netlist *nl) {
    if (!nl || !nl->c) return fail(TFHE_HIP_ERR_ARG, "null netlist");
    if (int r = check_ctx(nl->c, true)) return r;
    nl->c->ev_br = nl->c->ev_ks = false;
    if (nl->levels.empty()) return TFHE_HIP_OK;
    if (nl->exec) {
        HIPCHK(hipGraphLaunch(nl->exec, nl->c->stream));
        return TFHE_HIP_OK;
    }
    return netlist_enqueue(nl);
}

int tfhe_hip_netlist_is_graph(const tfhe_hip_netlist *nl) { return nl && nl->exec ? 1 : 0; }

// ---- device group: one context, stream and host thread per GPU (SURVEY.md 8(b) seam row, 8(e)) --------------------------
// The reference has no multi-device code.  Gates of a batch are independent, so a batch is split into
// contiguous shards, one per member, with no collective on the data path; keys are uploaded once to the first member
// and replicated device to device (hipMemcpyPeerAsync: xGMI where the GPUs are linked, staged by the runtime otherwise).
struct tfhe_hip_group {
    struct Worker {  // one persistent host thread per member: its thread-local staging buffers and error text live on
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::function<int()> job;
        bool has_job = false, done = false, quit = false;
        int rc = 0;
        std::string err;
    };
    std::vector<tfhe_hip_ctx *> ctx;
    std::vector<std::unique_ptr<Worker>> workers;
};
namespace {
void group_worker_main(tfhe_hip_group::Worker *w) {
    for (;;) {
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv.wait(lk, [&] { return w->has_job || w->quit; });
        if (w->quit) return;
        std::function<int()> job = std::move(w->job);
        w->has_job = false;
        lk.unlock();
        const int rc = job();
        const std::string err = rc ? g_err : std::string();
        lk.lock();
        w->rc = rc;
        w->err = err;
        w->done = true;
        w->cv.notify_all();
    }
}
// runs fn(member index) on every member's thread; first failure (lowest member) is reported to the caller's thread
int group_run(tfhe_hip_group *g, const std::function<int(int)> &fn) {
    const int n = (int) g->ctx.size();
    for (int i = 0; i < n; i++) {
        tfhe_hip_group::Worker *w = g->workers[i].get();
        std::lock_guard<std::mutex> lk(w->mu);
        w->job = [fn, i] { return fn(i); };
        w->has_job = true;
        w->done = false;
        w->cv.notify_all();
    }
    int rc = TFHE_HIP_OK;
    for (int i = 0; i < n; i++) {  // every member finishes before the call returns, whatever the others reported
        tfhe_hip_group::Worker *w = g->workers[i].get();
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv.wait(lk, [&] { return w->done; });
        if (w->rc && rc == TFHE_HIP_OK) rc = fail(w->rc, "group member " + std::to_string(i) + ": " + w->err);
    }
    return rc;
}
void shard(long long count, int i, int n, long long &lo, long long &hi) {  // contiguous blocks, sizes differ by at most one
    const long long base = count / n, rem = count % n;
    lo = i * base + std::min<long long>(i, rem);
    hi = lo + base + (i < rem ? 1 : 0);
}
// device-to-device replica of member 0's keys (after tfhe_hip_load_keys_* on member 0).  Between two devices the copy
// goes over xGMI only with peer access enabled: asked for and switched on here when the runtime offers it (once per
// pair; "already enabled" is fine); otherwise hipMemcpyPeerAsync is staged through the host by the runtime.  Which of the
// two happened is kept per member (tfhe_hip_group_key_path), and the replica is checked against the source's fingerprints.
int replicate_keys(tfhe_hip_ctx *dst, tfhe_hip_ctx *src) {
    HIPCHK(hipSetDevice(dst->device));
    dst->have_keys = false;
    dst->key_path = 3;
    if (dst->device != src->device) {
        int can = 0;
        HIPCHK(hipDeviceCanAccessPeer(&can, dst->device, src->device));
        dst->key_path = 2;
        if (can) {
            const hipError_t e = hipDeviceEnablePeerAccess(src->device, 0);
            if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) dst->key_path = 1;
            (void) hipGetLastError();  // "already enabled" is not an error of this call
        }
    }
    if (!dst->bk) HIPCHK(hipMalloc((void **) &dst->bk, kBkBytes));
    if (!dst->ks) HIPCHK(hipMalloc((void **) &dst->ks, kKsDevBytes));
    const size_t kl_bytes = (size_t) kN * 4 * kKsQ * 32;
    if (!dst->kl) HIPCHK(hipMalloc((void **) &dst->kl, kl_bytes));
    HIPCHK(hipMemcpyPeerAsync(dst->bk, dst->device, src->bk, src->device, kBkBytes, dst->stream));
    HIPCHK(hipMemcpyPeerAsync(dst->ks, dst->device, src->ks, src->device, kKsDevBytes, dst->stream));
    HIPCHK(hipMemcpyPeerAsync(dst->kl, dst->device, src->kl, src->device, kl_bytes, dst->stream));
    HIPCHK(hipStreamSynchronize(dst->stream));
    uint64_t got[3];
    if (int r = tfhe_hip_keys_fingerprint(dst, got)) return r;
    if (got[0] != src->key_print[0] || got[1] != src->key_print[1] || got[2] != src->key_print[2])
        return fail(TFHE_HIP_ERR_HIP, "replicated keys differ from the source's (device " + std::to_string(src->device) + " -> " +
                                          std::to_string(dst->device) + ")");
    if (int r = build_fast_key(dst)) return r;
    dst->have_keys = true;
    return TFHE_HIP_OK;
}
}  // namespace

int tfhe_hip_group_create(tfhe_hip_group **out, const int *devices, int ndev) {
    if (!out) return fail(TFHE_HIP_ERR_ARG, "null out pointer");
    *out = nullptr;
    if (!devices || ndev < 1 || ndev > 64) return fail(TFHE_HIP_ERR_ARG, "bad device list");
    std::unique_ptr<tfhe_hip_group, void (*)(tfhe_hip_group *)> g(new tfhe_hip_group, tfhe_hip_group_destroy);
    for (int i = 0; i < ndev; i++) {
        tfhe_hip_ctx *c = nullptr;
        if (int r = tfhe_hip_create(&c, devices[i])) return r;
        g->ctx.push_back(c);
    }
    for (int i = 0; i < ndev; i++) {
        g->workers.emplace_back(new tfhe_hip_group::Worker);
        g->workers.back()->th = std::thread(group_worker_main, g->workers.back().get());
    }
    *out = g.release();
    return TFHE_HIP_OK;
}
void tfhe_hip_group_destroy(tfhe_hip_group *g) {
    if (!g) return;
    for (auto &w : g->workers) {
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->quit = true;
            w->cv.notify_all();
        }
        if (w->th.joinable()) w->th.join();
    }
    for (tfhe_hip_ctx *c : g->ctx) tfhe_hip_destroy(c);
    delete g;
}
int tfhe_hip_group_size(const tfhe_hip_group *g) { return g ? (int) g->ctx.size() : 0; }
tfhe_hip_ctx *tfhe_hip_group_ctx(tfhe_hip_group *g, int i) {
    return g && i >= 0 && i < (int) g->ctx.size() ? g->ctx[i] : nullptr;
}
static int group_replicate(tfhe_hip_group *g) {
    if (g->ctx.size() > 1)
        if (int r = tfhe_hip_keys_fingerprint(g->ctx[0], g->ctx[0]->key_print)) return r;
    return group_run(g, [g](int i) { return i == 0 ? TFHE_HIP_OK : replicate_keys(g->ctx[i], g->ctx[0]); });
}
int tfhe_hip_group_key_path(const tfhe_hip_group *g, int member) {
    if (!g || member < 0 || member >= (int) g->ctx.size()) return TFHE_HIP_ERR_ARG;
    return g->ctx[member]->key_path;
}
int tfhe_hip_group_load_keys_coef(tfhe_hip_group *g, const int32_t *bk_coef, const int32_t *ks) {
    if (!g) return fail(TFHE_HIP_ERR_ARG, "null group");
    if (int r = tfhe_hip_load_keys_coef(g->ctx[0], bk_coef, ks)) return r;
    return group_replicate(g);
}
int tfhe_hip_group_load_keys_fft(tfhe_hip_group *g, const double *bkfft, const int32_t *ks) {
    if (!g) return fail(TFHE_HIP_ERR_ARG, "null group");
    if (int r = tfhe_hip_load_keys_fft(g->ctx[0], bkfft, ks)) return r;
    return group_replicate(g);
}
int tfhe_hip_group_set_arithmetic(tfhe_hip_group *g, int mode) {
    if (!g) return fail(TFHE_HIP_ERR_ARG, "null group");
    return group_run(g, [g, mode](int i) { return tfhe_hip_set_arithmetic(g->ctx[i], mode); });
}
int tfhe_hip_group_gate_batch(tfhe_hip_group *g, int gate, int32_t *out, const int32_t *a, const int32_t *b,
                              const int32_t *c, int64_t count) {
    if (!g) return fail(TFHE_HIP_ERR_ARG, "null group");
    if (count < 0 || (count > 0 && !out)) return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    const int n = (int) g->ctx.size();
    return group_run(g, [=](int i) -> int {
        long long lo, hi;
        shard(count, i, n, lo, hi);
        if (hi == lo) return TFHE_HIP_OK;
        const size_t o = (size_t) lo * kCt;
        return tfhe_hip_gate_batch(g->ctx[i], gate, out + o, a ? a + o : nullptr, b ? b + o : nullptr, c ? c + o : nullptr,
                                   hi - lo);
    });
}
// One level of a netlist over the group (BASELINE configs[3] on several GPUs without torch.distributed): every member
// holds a replica of the ciphertext pool; the level's gates are split into contiguous blocks, member i evaluates its
// block on its replica (tfhe_hip_gate_level_dev), gathers the rows it wrote into a staging buffer, and every other member
// pulls that buffer with ONE peer copy and scatters it into its own replica - 2,524 B per gate and peer, no host copy.
int tfhe_hip_group_gate_level(tfhe_hip_group *g, int32_t *const *d_pools, int64_t pool_rows, const int32_t *gates, int64_t count) {
    if (!g) return fail(TFHE_HIP_ERR_ARG, "null group");
    if (count < 0 || pool_rows < 0 || !d_pools || (count > 0 && !gates)) return fail(TFHE_HIP_ERR_ARG, "bad buffer or count");
    const int n = (int) g->ctx.size();
    for (int i = 0; i < n; i++)
        if (!d_pools[i]) return fail(TFHE_HIP_ERR_ARG, "null pool replica");
    if (count == 0) return TFHE_HIP_OK;
    if (int r = check_ctx(g->ctx[0], true)) return r;
    if (int r = expand_level(g->ctx[0], pool_rows, gates, count)) return r;  // the level contract over the WHOLE level
    std::vector<int32_t> out_rows((size_t) count);
    for (int64_t q = 0; q < count; q++) out_rows[(size_t) q] = gates[5 * q + 1];
    // phase 1: every member evaluates its block and stages the rows it wrote
    int rc = group_run(g, [&, n](int i) -> int {
        tfhe_hip_ctx *c = g->ctx[i];
        long long lo, hi;
        shard(count, i, n, lo, hi);
        if (int r = check_ctx(c, true)) return r;
        HIPCHK(c->grp_rows.reserve((size_t) count * sizeof(int32_t)));
        HIPCHK(hipMemcpyAsync(c->grp_rows.p, out_rows.data(), (size_t) count * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
        if (hi == lo) {
            HIPCHK(hipStreamSynchronize(c->stream));
            return TFHE_HIP_OK;
        }
        if (int r = tfhe_hip_gate_level_dev(c, d_pools[i], pool_rows, gates + 5 * lo, hi - lo)) return r;
        if (n == 1) return TFHE_HIP_OK;
        const long long total = (hi - lo) * kCt;
        HIPCHK(c->grp_stage.reserve((size_t) total * sizeof(int32_t)));
        hipLaunchKernelGGL(k_rows_gather, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, c->stream, (int32_t *) c->grp_stage.p,
                           (const int32_t *) d_pools[i], (const int32_t *) c->grp_rows.p + lo, total);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(c->stream));
        return TFHE_HIP_OK;
    });
    if (rc || n == 1) return rc;
    // phase 2: every member pulls the other members' staged rows into its own replica
    return group_run(g, [&, n](int j) -> int {
        tfhe_hip_ctx *c = g->ctx[j];
        HIPCHK(hipSetDevice(c->device));
        for (int i = 0; i < n; i++) {
            long long lo, hi;
            shard(count, i, n, lo, hi);
            if (i == j || hi == lo) continue;
            const long long total = (hi - lo) * kCt;
            HIPCHK(c->grp_in.reserve((size_t) total * sizeof(int32_t)));
            HIPCHK(hipMemcpyPeerAsync(c->grp_in.p, c->device, g->ctx[i]->grp_stage.p, g->ctx[i]->device, (size_t) total * sizeof(int32_t),
                                      c->stream));
            hipLaunchKernelGGL(k_rows_scatter, dim3((unsigned) ((total + 255) / 256)), dim3(256), 0, c->stream, d_pools[j],
                               (const int32_t *) c->grp_in.p, (const int32_t *) c->grp_rows.p + lo, total);
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(c->stream));  // grp_in is reused by the next peer
        }
        return TFHE_HIP_OK;
    });
}
}  // extern "C"
int tfhe_hip_internal_group_run(tfhe_hip_group *g, int (*fn)(void *user, tfhe_hip_ctx *ctx, int member, int members),
                                void *user) {
    if (!g || !fn) return fail(TFHE_HIP_ERR_ARG, "null group");
    const int n = (int) g->ctx.size();
    return group_run(g, [=](int i) { return fn(user, g->ctx[i], i, n); });
}
extern "C" {

// device-memory helpers so that C callers of the *_dev entry points need no HIP of their own
int tfhe_hip_dev_alloc(tfhe_hip_ctx *c, size_t bytes, void **out) {
    if (int r = check_ctx(c, false)) return r;
    if (!out) return fail(TFHE_HIP_ERR_ARG, "null out pointer");
    HIPCHK(hipMalloc(out, bytes ? bytes : 1));
    return TFHE_HIP_OK;
}
int tfhe_hip_dev_free(tfhe_hip_ctx *c, void *p) {
    if (int r = check_ctx(c, false)) return r;
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipFree(p));
    return TFHE_HIP_OK;
}
int tfhe_hip_copy_to_dev(tfhe_hip_ctx *c, void *dst, const void *src, size_t bytes) {
    if (int r = check_ctx(c, false)) return r;
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return TFHE_HIP_OK;
}
int tfhe_hip_copy_from_dev(tfhe_hip_ctx *c, void *dst, const void *src, size_t bytes) {
    if (int r = check_ctx(c, false)) return r;
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return TFHE_HIP_OK;
}

}  // extern "C"

#include "nk_ops.h"

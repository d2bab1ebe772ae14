// Negacyclic transforms with the ring dimension as a parameter: N = 512, 1024, 2048 (NH = N/2 = 2^LH complex points,
// LH = 8, 9, 10), one wavefront per polynomial, 2^(LH-6) = 4 / 8 / 16 complex values per lane.
//
// The reference's TGSW / TLWE layer takes N, k, l, Bgbit from its parameter structs (src/libtfhe/tgsw-functions.cpp:157-170,
// 277-293, 338-352; src/test/tgsw_fft_test.cpp:28-35 runs N = 512 / 1024 / 2048 with k = 1 / 2), but this fork compiles its
// transform for N = 1024 only (src/include/tfhe_core.h:14, polynomials.cpp:13-175), so at other N there is no reference
// transform to be bit-identical to: these are the folded FMA transforms of fast_core.h, written once for every N -
//   omega = exp(i pi / N), z_j = p_j + i p_(j+NH):   Z_k = P(omega^(4k+1)) = sum_j z_j omega^(j(4k+1)),  k < NH
//   forward   radix-2 DIT over positions x (input z_(bitrev x), output Z_x), level L pairs x and x + 2^L,
//             twiddle t(L, k') = exp(i pi (4k'+1) / 2^(L+2)), k' = x mod 2^L;  t(L, k' + 2^(L-1)) = i t(L, k')
//   inverse   z_j = (1/NH) omega^(-j) sum_k Y_k exp(-2 pi i jk / NH): plain DIT, w(L, k') = exp(-2 pi i k' / 2^(L+1)),
//             quarter turn -i w; then the untwist omega^(-j) 2^32 / NH and truncation toward zero (polynomials.cpp:392)
//   reference slot i of a spectrum (exponent 2i+1):  C[i] = Z_(i/2) for even i, conj(Z_((N-1-i)/2)) for odd i
// and checked against the reference's exact products (torusPolynomialMultNaive_aux, multiplication.cpp:31-43, whose N is
// a run-time argument) within the tolerance of its own test (src/test/test-multiplication.cpp:66-77).
//
// Data layout.  R = LH - 6 index bits live in the register number, six in the lane.  Levels run in phases of up to R:
// phase p keeps the position bits [b0, b0 + R), b0 = min(pR, 6), in the register number; between phases the values pass
// through a wave-private LDS buffer of NH values (swizzled: conflict-free 16-byte stores and loads under the gfx950
// lane-group rules for LH = 9, 10, within 11 % of it for LH = 8; tools/lds_sim.py).  Phase 0 holds position
// r | bitrev6(lane) << R, i.e. input coefficient bitrev_R(r) << 6 | lane: coalesced loads; the last phase holds
// r << 6 | lane: coalesced stores, and - as in fast_core.h - the pointwise product of two spectra enters the inverse
// transform by register renaming.  Twiddles: one table per direction (tw_index: 2^(L-1) entries for level L) and the untwist
// factors, built on the host in long double and copied to LDS by each workgroup.
#pragma once
#include "fast_core.h"

namespace nfft {
using brc::c64;

template <int LH>
struct Dim {
    static constexpr int NH = 1 << LH, N = 2 << LH, R = LH - 6, VPL = 1 << R, PHASES = (LH + R - 1) / R;
    static constexpr int kTableEntries = NH / 2 + NH / 2 + NH;  // forward | inverse | untwist
    static_assert(LH >= 8 && LH <= 10, "N = 512, 1024, 2048");
};

// entry of level L, position k' in a direction's table; only k' < 2^(L-1) is ever read (the other half of a level are
// quarter turns of it and are folded into the butterflies), so a level holds 2^(L-1) entries (level 0: one)
BRC_HD constexpr int tw_index(int L, int kp) { return (L ? 1 << (L - 1) : 0) + kp; }
// host: twiddle image of dimension LH (long double trigonometry, rounded once)
template <int LH>
inline void build_tables(c64 *out /*[Dim<LH>::kTableEntries]*/) {
    using D = Dim<LH>;
    const long double pi = 3.14159265358979323846264338327950288L;
    for (int L = 0; L < LH; L++)
        for (int kp = 0; kp < (L ? 1 << (L - 1) : 1); kp++) {
            const long double af = pi * (4 * kp + 1) / (long double) (1 << (L + 2)), ai = -2 * pi * kp / (long double) (1 << (L + 1));
            out[tw_index(L, kp)] = {(double) cosl(af), (double) sinl(af)};
            out[D::NH / 2 + tw_index(L, kp)] = {(double) cosl(ai), (double) sinl(ai)};
        }
    const long double scale = 4294967296.0L / D::NH;
    for (int j = 0; j < D::NH; j++) {
        const long double a = -pi * j / D::N;
        out[D::NH + j] = {(double) (scale * cosl(a)), (double) (scale * sinl(a))};
    }
}

BRC_HD constexpr int bitrev(int v, int bits) {
    int r = 0;
    for (int i = 0; i < bits; i++) r |= ((v >> i) & 1) << (bits - 1 - i);
    return r;
}
template <int LH>
BRC_HD constexpr int swz(int x) {
    return LH == 8 ? x ^ ((x >> 2) & 15) ^ ((x >> 5) & 7) : LH == 9 ? x ^ ((x >> 3) & 15) ^ ((x >> 6) & 7) : x ^ ((x >> 3) & 7) ^ ((x >> 7) & 7);
}
template <int LH>
BRC_HD constexpr int phase_base(int p) {
    return p * Dim<LH>::R < 6 ? p * Dim<LH>::R : 6;
}
// position held by (lane, register r) in phase p; lane6 = bitrev6(lane), needed by phase 0 only
template <int LH>
BRC_HD int position(int p, int lane, int lane6, int r) {
    constexpr int R = Dim<LH>::R;
    if (p == 0) return r | (lane6 << R);
    const int b0 = phase_base<LH>(p);
    return (lane & ((1 << b0) - 1)) | (r << b0) | ((lane >> b0) << (b0 + R));
}

// levels [p R, min((p+1) R, LH)) on the register bits of phase p; tw = this direction's table (LDS)
template <int LH, bool INV>
BRC_HD void phase(c64 (&v)[Dim<LH>::VPL], int p, int lane, const c64 *tw) {
    constexpr int R = Dim<LH>::R, VPL = Dim<LH>::VPL;
    const int b0 = phase_base<LH>(p);
    const int lane_low = lane & ((1 << b0) - 1);
    const int L_lo = p * R, L_hi = (p + 1) * R < LH ? (p + 1) * R : LH;
#pragma unroll
    for (int L = L_lo; L < L_hi; L++) {
        const int j = L - b0;  // register bit of this level
        c64 t[VPL / 2];        // twiddles of the register patterns below bit j, quarter-turn partners folded
        if (j == 0 && L > 0) {  // the top bit of k' is a lane bit here: the quarter turn is applied to the twiddle itself
            const c64 e = tw[tw_index(L, lane_low & ((1 << (L - 1)) - 1))];
            const bool top = (lane_low >> (L - 1)) & 1;
            t[0] = !top ? e : INV ? c64{e.im, -e.re} : c64{-e.im, e.re};
        } else {
#pragma unroll
            for (int q = 0; q < VPL / 2; q++)
                if (q < (j ? 1 << (j - 1) : 1)) t[q] = tw[tw_index(L, lane_low | (q << b0))];
        }
#pragma unroll
        for (int r = 0; r < VPL; r++) {
            if (r & (1 << j)) continue;
            const int below = r & ((1 << j) - 1);
            const bool turn = j > 0 && (below >> (j - 1)) & 1;
            const c64 tt = t[j > 0 ? below & ((1 << (j - 1)) - 1) : 0];
            if (!turn) brf::bf(v[r], v[r | (1 << j)], tt);
            else if (INV) brf::bf_mi(v[r], v[r | (1 << j)], tt);
            else brf::bf_pi(v[r], v[r | (1 << j)], tt);
        }
    }
}

#if defined(__HIPCC__)
// one transform of the wave: v in phase-0 layout on entry (register r = element bitrev_R(r) << 6 | lane of the input
// sequence), in last-phase layout (register r = index r << 6 | lane) on exit; scratch = NH values of wave-private LDS
template <int LH, bool INV>
__device__ __forceinline__ void transform(c64 (&v)[Dim<LH>::VPL], int lane, c64 *scratch, const c64 *tables) {
    using D = Dim<LH>;
    const c64 *tw = tables + (INV ? D::NH / 2 : 0);
    const int lane6 = bitrev(lane, 6);
#pragma unroll
    for (int p = 0; p < D::PHASES; p++) {
        phase<LH, INV>(v, p, lane, tw);
        if (p + 1 < D::PHASES) {
#pragma unroll
            for (int r = 0; r < D::VPL; r++) scratch[swz<LH>(position<LH>(p, lane, lane6, r))] = v[r];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < D::VPL; r++) v[r] = scratch[swz<LH>(position<LH>(p + 1, lane, lane6, r))];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront", "local");
            __builtin_amdgcn_wave_barrier();
        }
    }
}
// untwist, scaling and truncation toward zero of an inverse transform's output: coefficient j = r << 6 | lane in lo[r],
// j + NH in hi[r]
template <int LH>
__device__ __forceinline__ void finish_inverse(int32_t (&lo)[Dim<LH>::VPL], int32_t (&hi)[Dim<LH>::VPL],
                                               const c64 (&v)[Dim<LH>::VPL], int lane, const c64 *tables) {
    using D = Dim<LH>;
#pragma unroll
    for (int r = 0; r < D::VPL; r++) {
        const c64 u = tables[D::NH + ((r << 6) | lane)];
        lo[r] = brf::to_torus32<false>(brf::fmad(v[r].re, u.re, -(v[r].im * u.im)));
        hi[r] = brf::to_torus32<false>(brf::fmad(v[r].re, u.im, v[r].im * u.re));
    }
}
#endif  // __HIPCC__
// index of spectrum value Z_k in the reference's coefsC order, and whether it is stored conjugated there
template <int LH>
BRC_HD int ref_slot(int k, bool &conj) {
    conj = k >= Dim<LH>::NH / 2;
    return conj ? Dim<LH>::N - 1 - 2 * k : 2 * k;
}

}  // namespace nfft

// Per-lane building blocks of the FAST arithmetic mode (tfhe_hip_set_arithmetic(ctx, TFHE_HIP_ARITH_FAST)).
//
// The EXACT mode (br_core.h) repeats the reference's 2048-point radix-2 transform operation by operation.  This mode
// computes the same negacyclic transforms (reference src/libtfhe/polynomials.cpp:347-393) in
// their minimal folded form - N/2 = 512 complex points in both directions, fused multiply-adds everywhere - and
// agrees with the reference within the tolerances of its own tests (src/test/test-MulToTLwe.cpp:365-405,
// src/test/test-multiplication.cpp:66-77), not bit for bit.  Integer stages (rotation, decomposition, extraction,
// keyswitch) are shared with the exact mode and stay exact.
//
//   omega = exp(i pi / N), z_j = p_j + i p_{j+N/2} (j < N/2)
//   forward   Z_k = P(omega^(4k+1)) = sum_j z_j omega^(j (4k+1)),  k < N/2
//             as a radix-2 DIT whose twiddles carry the twist:  level L (blocks of 2^L -> 2^(L+1)), position k' < 2^L:
//                 t = exp(i pi (4k'+1) / 2^(L+2)),   lo = E + t O,  hi = E - t O = 2E - lo       (6 FMA per butterfly)
//             and t(k' + 2^(L-1)) = i t(k'): half of each level's twiddles are quarter turns of the other half.
//             Input in bit-reversed positions, output natural: the layouts A/B/C of br_core.h (fwd_xA/xB/xC).
//   inverse   z_j = (1/512) omega^(-j) sum_k Y_k exp(-2 pi i jk/512): plain radix-2 DIT (twiddle w = exp(-2 pi i k'/2^(L+1)),
//             quarter turn = -i w), then the untwist, which also carries the scaling 2^32 / 512.  The forward output of lane
//             l, register r is Z_(r<<6|l); the inverse wants Y_(br3(r')<<6|l) in register r': the same lane, so the
//             summed spectrum feeds the inverse transform by register renaming, with no exchange.
//   Reference slot i (exponent 2i+1):  C[i] = Z_(i/2) for even i, conj(Z_((N-1-i)/2)) for odd i.
#pragma once
#include <math.h>
#include "br_core.h"

namespace brf {
using brc::c64;
using brc::PhaseAConsts;

enum { FS_FB = 0, FS_FC = 4, FS_IB = 8, FS_IC = 12, FS_IF = 16 };
constexpr int kFastSlots = 24;
constexpr int kFastLaneEntries = kFastSlots * 64;  // 1536 c64 = 24,576 bytes

// value of slot `slot` of lane `lane` of the per-lane twiddle image (host; long double trigonometry, rounded once)
inline c64 fast_slot_value(int slot, int lane) {
    const long double pi = 3.14159265358979323846264338327950288L;
    const int l3 = lane & 7;
    long double ang;
    if (slot < 8) {  // forward: exp(i pi (4k'+1) / 2^(L+2))
        static const int den[8] = {32, 64, 128, 128, 256, 512, 1024, 1024};
        const int kp = slot < 4 ? l3 + (slot == 3 ? 8 : 0) : lane + (slot == 7 ? 64 : 0);
        ang = pi * (4 * kp + 1) / den[slot];
        return {(double) cosl(ang), (double) sinl(ang)};
    }
    if (slot < 16) {  // inverse: exp(-2 pi i k' / 2^(L+1))
        static const int den[8] = {16, 32, 64, 64, 128, 256, 512, 512};
        const int s = slot - 8;
        const int kp = s < 4 ? l3 + (s == 3 ? 8 : 0) : lane + (s == 7 ? 64 : 0);
        ang = -2 * pi * kp / den[s];
        return {(double) cosl(ang), (double) sinl(ang)};
    }
    const int j = ((slot - 16) << 6) | lane;  // untwist of coefficient j, scaled by 2^32 / 512
    ang = -pi * j / 1024;
    return {(double) (8388608.0L * cosl(ang)), (double) (8388608.0L * sinl(ang))};
}
inline void build_fast_twiddles(c64 *out /*[kFastLaneEntries]*/) {
    for (int s = 0; s < kFastSlots; s++)
        for (int l = 0; l < 64; l++) out[s * 64 + l] = fast_slot_value(s, l);
}

BRC_HD double fmad(double a, double b, double c) { return __builtin_fma(a, b, c); }

// lo = E + t O, hi = E - t O
BRC_HD void bf(c64 &E, c64 &O, const c64 t) {
    const double lr = fmad(t.re, O.re, fmad(-t.im, O.im, E.re));
    const double li = fmad(t.re, O.im, fmad(t.im, O.re, E.im));
    O = {fmad(2.0, E.re, -lr), fmad(2.0, E.im, -li)};
    E = {lr, li};
}
// the same with the twiddle i t (forward quarter turn)
BRC_HD void bf_pi(c64 &E, c64 &O, const c64 t) {
    const double lr = fmad(-t.im, O.re, fmad(-t.re, O.im, E.re));
    const double li = fmad(-t.im, O.im, fmad(t.re, O.re, E.im));
    O = {fmad(2.0, E.re, -lr), fmad(2.0, E.im, -li)};
    E = {lr, li};
}
// the same with the twiddle -i t (inverse quarter turn)
BRC_HD void bf_mi(c64 &E, c64 &O, const c64 t) {
    const double lr = fmad(t.im, O.re, fmad(t.re, O.im, E.re));
    const double li = fmad(t.im, O.im, fmad(-t.re, O.re, E.im));
    O = {fmad(2.0, E.re, -lr), fmad(2.0, E.im, -li)};
    E = {lr, li};
}
BRC_HD void bf_one(c64 &E, c64 &O) {  // t = 1
    const c64 lo = {E.re + O.re, E.im + O.im}, hi = {E.re - O.re, E.im - O.im};
    E = lo;
    O = hi;
}
BRC_HD void bf_negi(c64 &E, c64 &O) {  // t = -i: t O = (O.im, -O.re)
    const c64 lo = {E.re + O.im, E.im - O.re}, hi = {E.re - O.im, E.im + O.re};
    E = lo;
    O = hi;
}
BRC_HD void cmac(c64 &acc, const c64 a, const c64 b) {
    acc.re = fmad(a.re, b.re, fmad(-a.im, b.im, acc.re));
    acc.im = fmad(a.re, b.im, fmad(a.im, b.re, acc.im));
}
BRC_HD c64 cmul(const c64 a, const c64 b) { return {fmad(a.re, b.re, -(a.im * b.im)), fmad(a.re, b.im, a.im * b.re)}; }

// ---- forward, levels 0..2 (register bits = x0 x1 x2); constants: cos/sin of pi/4, pi/8, pi/16, 3pi/16 -----------------
BRC_HD void fwd_phaseA(c64 (&v)[8], const PhaseAConsts &k) {
    const c64 t0 = {k.c4, k.c4};  // exp(i pi/4)
#pragma unroll
    for (int r = 0; r < 8; r += 2) bf(v[r], v[r + 1], t0);
    const c64 t1 = {k.c8, k.s8};  // exp(i pi/8); k' = 1: i t1
#pragma unroll
    for (int g = 0; g < 8; g += 4) {
        bf(v[g + 0], v[g + 2], t1);
        bf_pi(v[g + 1], v[g + 3], t1);
    }
    const c64 t2a = {k.c16, k.s16}, t2b = {k.s316, k.c316};  // exp(i pi/16), exp(i 5pi/16); k' = 2, 3: i times these
    bf(v[0], v[4], t2a);
    bf(v[1], v[5], t2b);
    bf_pi(v[2], v[6], t2a);
    bf_pi(v[3], v[7], t2b);
}
// levels 3..5 (BASE = FS_FB) and 6..8 (BASE = FS_FC): register bits are the three position bits of the phase
template <int BASE, class TW>
BRC_HD void fwd_phaseBC(c64 (&v)[8], const TW &tw) {
    {
        const c64 t = tw(BASE + 0);
#pragma unroll
        for (int r = 0; r < 8; r += 2) bf(v[r], v[r + 1], t);
    }
    {
        const c64 t = tw(BASE + 1);
#pragma unroll
        for (int g = 0; g < 8; g += 4) {
            bf(v[g + 0], v[g + 2], t);
            bf_pi(v[g + 1], v[g + 3], t);
        }
    }
    {
        const c64 a = tw(BASE + 2), b = tw(BASE + 3);
        bf(v[0], v[4], a);
        bf(v[1], v[5], b);
        bf_pi(v[2], v[6], a);
        bf_pi(v[3], v[7], b);
    }
}

// ---- inverse -----------------------------------------------------------------------------------------------------
// input register r' = Y_(br3(r')<<6 | lane); levels 0..2: twiddles 1 | 1, -i | 1, exp(-i pi/4), -i, -i exp(-i pi/4)
BRC_HD void inv_phaseA(c64 (&v)[8], const PhaseAConsts &k) {
#pragma unroll
    for (int r = 0; r < 8; r += 2) bf_one(v[r], v[r + 1]);
#pragma unroll
    for (int g = 0; g < 8; g += 4) {
        bf_one(v[g + 0], v[g + 2]);
        bf_negi(v[g + 1], v[g + 3]);
    }
    const c64 w = {k.c4, -k.c4};
    bf_one(v[0], v[4]);
    bf(v[1], v[5], w);
    bf_negi(v[2], v[6]);
    bf_mi(v[3], v[7], w);
}
template <int BASE, class TW>
BRC_HD void inv_phaseBC(c64 (&v)[8], const TW &tw) {
    {
        const c64 t = tw(BASE + 0);
#pragma unroll
        for (int r = 0; r < 8; r += 2) bf(v[r], v[r + 1], t);
    }
    {
        const c64 t = tw(BASE + 1);
#pragma unroll
        for (int g = 0; g < 8; g += 4) {
            bf(v[g + 0], v[g + 2], t);
            bf_mi(v[g + 1], v[g + 3], t);
        }
    }
    {
        const c64 a = tw(BASE + 2), b = tw(BASE + 3);
        bf(v[0], v[4], a);
        bf(v[1], v[5], b);
        bf_mi(v[2], v[6], a);
        bf_mi(v[3], v[7], b);
    }
}
// untwist + scaling + truncation toward zero (polynomials.cpp:392): register r of lane l gives the coefficients
// j = r<<6|l (real part) and j + 512 (imaginary part).  BOUNDED as brc::scale_to_torus32.
template <bool BOUNDED>
BRC_HD int32_t to_torus32(double x) {
    return BOUNDED ? brc::trunc_to_torus32_bounded(x) : (int32_t) (uint32_t) (uint64_t) (int64_t) x;
}
template <bool BOUNDED, class TW>
BRC_HD void inv_final(int32_t (&lo)[8], int32_t (&hi)[8], const c64 (&v)[8], const TW &tw) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
        const c64 u = tw(FS_IF + r);
        lo[r] = to_torus32<BOUNDED>(fmad(v[r].re, u.re, -(v[r].im * u.im)));
        hi[r] = to_torus32<BOUNDED>(fmad(v[r].re, u.im, v[r].im * u.re));
    }
}

// reference slot i of a spectrum held as Z_k
BRC_HD int ref_slot_source(int i) { return (i & 1) ? (brc::kN - 1 - i) >> 1 : i >> 1; }

}  // namespace brf

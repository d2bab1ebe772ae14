// Per-lane building blocks of the blind-rotation kernel (one wavefront = one ciphertext, 64 lanes).
//
// Everything here is written as "what lane `lane` of the wave does", with the cross-lane exchanges going through a
// wave-private LDS scratch area.  The same code is compiled for the GPU (tfhe_hip_kernels.hip) and, for
// pre-validation without a GPU, for the host (tests/emu/emu_wave.cpp runs the 64 lanes in a loop, phase by phase).
//
// ARITHMETIC CONTRACT.  The reference computes its negacyclic transforms with a 2048-point radix-2 FFT
// (reference src/libtfhe/polynomials.cpp:13-175) on an antisymmetric real extension (:347-362) / on a Hermitian
// spectrum (:378-393).  In that computation most operands are exact zeros or exact complex conjugates of other
// operands.  The routines below perform exactly the remaining floating-point operations - the same IEEE-754
// binary64 multiplications and additions, on the same operands, each rounded on its own (no FMA contraction:
// this file must be compiled with -ffp-contract=off) - so their results are BIT-IDENTICAL to the reference's:
//
//  forward  (IntPolynomial_ifft / TorusPolynomial_ifft): after the size-2 and size-4 merges the 2048 array holds
//           z_k = p_k + i p_{k+N/2} (k bit-reversed) and conjugates; every later block keeps only odd bins and
//           is Hermitian, so 512 complex values per level suffice: 9 levels x 256 butterflies
//               E' = E + w O ,   O' = conj(E - w O)
//           done in place; the conjugate-mirrored placement of the reference turns into a Gray-code relabelling:
//           physical slot x holds logical bin igray(x), twiddle index q = 2*igray_k(x mod 2^k) + 1.
//  inverse  (TorusPolynomial_fft): the even half of the bit-reversed array stays zero, the odd half is a plain
//           1024-point DIT FFT of c~[m] = C[m] (m<512), conj(C[1023-m]) (m>=512), followed by the last merge
//           of which only  Re = Wr*cos + Wi*sin  is used, scaled by 2^22 and truncated toward zero (:392).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define BRC_HD __host__ __device__ __forceinline__
#else
#define BRC_HD inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

namespace brc {

constexpr int kN = 1024;       // polynomial degree
constexpr int kNh = 512;       // N/2
// The parameter set is a compile-time choice: the library is built once per set (tfhe_amd/build.py), the default
// 128-bit set of tfhe_gate_bootstrapping.cpp:51-75 as libtfhe_hip.so and, with -DTFHE_HIP_LAMBDA=80, the 80-bit set
// of :23-48 as libtfhe_hip_80.so.  N, k and the keyswitch decomposition (t = 8, base 4) are the same in both.
#ifndef TFHE_HIP_LAMBDA
#define TFHE_HIP_LAMBDA 128
#endif
#if TFHE_HIP_LAMBDA == 128
constexpr int kLwe = 630;      // n
constexpr int kL = 3;          // decomposition length
constexpr int kBgbit = 7;
#elif TFHE_HIP_LAMBDA == 80
constexpr int kLwe = 500;
constexpr int kL = 2;
constexpr int kBgbit = 10;
#else
#error "TFHE_HIP_LAMBDA must be 128 or 80"
#endif
constexpr int kKpl = 2 * kL;   // (k+1)*l rows
// sum over the digits of Bg/2 at the digit's position (tgsw.cpp:21-27; 2164391936 = src/include/tfhe_core.h:24 for the
// 128-bit set): one bit per digit, 31 - p*Bgbit.  The same word is the mask of the digits' top bits.
constexpr uint32_t decomp_offset_bits() {
    uint32_t m = 0;
    for (int p = 0; p < kL; p++) m |= 1u << (31 - p * kBgbit);
    return m;
}
constexpr uint32_t kDecompOffset = decomp_offset_bits();
static_assert(TFHE_HIP_LAMBDA != 128 || kDecompOffset == 2164391936u, "offset of the default set");
constexpr int kTwEntries = 2044;                 // base tables, sizes 8..2048: sum of size/2
// Per-lane twiddle image kept in LDS: 31 slots x 64 lanes; slot s of lane l sits at [s*64 + l], so every table read of
// the lane-dependent phases is `lane base + constant` (one address register, immediate offsets) and the 64 lanes of
// a wave read 64 consecutive 16-byte entries (no bank conflicts).  Which base-table entry goes where: tw_slot_index.
constexpr int kTwSlots = 31;
constexpr int kTwLaneEntries = kTwSlots * 64;    // 1984 c64 = 31,744 bytes
enum { TWS_FB = 0, TWS_FC = 4, TWS_IB = 8, TWS_IC = 15, TWS_FIN = 23 };

struct alignas(16) c64 {
    double re, im;
};

// index of twiddle j of the table for merge size `size` (8..2048): w = cos(2 pi j/size) + i sin(2 pi j/size)
BRC_HD constexpr int tw_idx(int size, int j) { return size / 2 - 4 + j; }

BRC_HD constexpr int br3(int v) { return ((v & 1) << 2) | (v & 2) | ((v >> 2) & 1); }
BRC_HD constexpr int br4(int v) { return ((v & 1) << 3) | ((v & 2) << 1) | ((v >> 1) & 2) | ((v >> 3) & 1); }
BRC_HD int br6(int v) { return (br3(v & 7) << 3) | br3((v >> 3) & 7); }
// inverse Gray code (prefix xor from the top) of a value of up to 16 bits
BRC_HD constexpr int igray(int u) {
    u ^= u >> 1;
    u ^= u >> 2;
    u ^= u >> 4;
    u ^= u >> 8;
    return u;
}
BRC_HD constexpr int gray(int u) { return u ^ (u >> 1); }

// Base-table index held in slot `slot` of lane `lane` (layouts of the phases: see below).  Only entries of the first
// quarter turn are stored; the others follow from two exact symmetries of the reference table (octant folding,
// src/include/tfhe_core.h:26,93):   table(n, n/4 - j) = (sin_j, cos_j)      ("mirror", used by the forward phases)
//                                   table(n, n/4 + j) = (-sin_j, cos_j)     ("quarter", used by the inverse phases)
BRC_HD int tw_slot_index(int slot, int lane) {
    const int l3 = lane & 7, g3 = igray(l3), g6 = igray(lane), low = ((lane >> 3) << 4) | l3;
    if (slot == 0) return tw_idx(64, 2 * g3 + 1);                      // forward level 3
    if (slot == 1) return tw_idx(128, 2 * g3 + 1);                     // forward level 4
    if (slot == 2) return tw_idx(256, 2 * g3 + 1);                     // forward level 5, x4x3 = 00
    if (slot == 3) return tw_idx(256, 2 * (16 | g3) + 1);              //                   x4x3 = 11
    if (slot == 4) return tw_idx(512, 2 * g6 + 1);                     // forward level 6
    if (slot == 5) return tw_idx(1024, 2 * g6 + 1);                    // forward level 7
    if (slot == 6) return tw_idx(2048, 2 * g6 + 1);                    // forward level 8, x7x6 = 00
    if (slot == 7) return tw_idx(2048, 2 * (128 | g6) + 1);            //                   x7x6 = 11
    if (slot == 8) return tw_idx(32, l3);                              // inverse level 4, y3 = 0
    if (slot < 11) return tw_idx(64, ((slot - 9) << 3) | l3);          // inverse level 5, y4 = 0, y3 = slot-9
    if (slot < 15) return tw_idx(128, ((slot - 11) << 3) | l3);        // inverse level 6, y5 = 0, (y4 y3) = slot-11
    if (slot < 17) return tw_idx(256, low | ((slot - 15) << 3));       // inverse level 7, y3 = slot-15
    if (slot < 19) return tw_idx(512, low | ((slot - 17) << 3));       // inverse level 8, y7 = 0, y3 = slot-17
    if (slot < 23) return tw_idx(1024, (((slot - 19) >> 1) << 7) | low | (((slot - 19) & 1) << 3));  // level 9, y8 = 0
    const int rho = slot - 23;                                         // last merge, y9 = 0 (rho < 8)
    return tw_idx(2048, ((rho >> 1) << 7) | low | ((rho & 1) << 3));
}
// lane image from the base tables; BASE: (index) -> c64
template <class BASE>
inline void build_lane_twiddles(c64 *out /*[kTwLaneEntries]*/, const BASE &base) {
    for (int s = 0; s < kTwSlots; s++)
        for (int l = 0; l < 64; l++) out[s * 64 + l] = base(tw_slot_index(s, l));
}

// ---- LDS scratch swizzle (slot = 16-byte element index inside the wave's 8 KB scratch) ------------------------
// XORs slot bits 6..8 into bits 0..2 and bit 7 into bit 3.  With it every ds_write_b128 / ds_read_b128 of the three
// layouts used below (A: x = br6(l)<<3|r, B: x = (l>>3)<<6|r<<3|(l&7), C: x = r<<6|l) and of the Gray-ordered read
// that feeds the inverse transform is bank-conflict free under the gfx950 lane-group rules (tools/lds_sim.py).
BRC_HD int swz(int x) { return x ^ ((x >> 6) & 7) ^ (((x >> 7) & 1) << 3); }

// ---- butterflies ---------------------------------------------------------------------------------------------
// forward ("reverse" table of the reference, polynomials.cpp:78-85 with s = -sin):
//   t = (Or*c + Oi*(-s), Oi*c - Or*(-s));  x*(-s) == -(x*s) and a+(-b) == a-b exactly, hence the form below
BRC_HD void fwd_bf(c64 &E, c64 &O, const c64 w) {
    const double tr = O.re * w.re - O.im * w.im;
    const double ti = O.im * w.re + O.re * w.im;
    const c64 lo = {E.re + tr, E.im + ti};
    const c64 hi = {E.re - tr, ti - E.im};  // conj(E - t)
    E = lo;
    O = hi;
}
// inverse ("direct" table, polynomials.cpp:160-167)
BRC_HD void inv_bf(c64 &E, c64 &O, const c64 w) {
    const double tr = O.re * w.re + O.im * w.im;
    const double ti = O.im * w.re - O.re * w.im;
    const c64 lo = {E.re + tr, E.im + ti};
    const c64 hi = {E.re - tr, E.im - ti};
    E = lo;
    O = hi;
}

// inverse butterfly with the twiddle a quarter turn after w: table(n, j + n/4) = (-s, c) for w = table(n, j) = (c, s).
// Value-identical to inv_bf with that entry: Or*(-s) + Oi*c = Oi*c - Or*s and Oi*(-s) - Or*c = -(Oi*s) - Or*c exactly.
BRC_HD void inv_bf_q(c64 &E, c64 &O, const c64 w) {
    const double tr = O.im * w.re - O.re * w.im;
    const double ti = -(O.im * w.im) - O.re * w.re;
    const c64 lo = {E.re + tr, E.im + ti};
    const c64 hi = {E.re - tr, E.im - ti};
    E = lo;
    O = hi;
}
BRC_HD c64 mirror(const c64 w) { return {w.im, w.re}; }  // table(n, n/4 - j) from table(n, j)

// =================================================================================================================
// FORWARD TRANSFORM, 512 complex values, 8 per lane.
//   physical slot x (9 bits).  phase A: lane l, reg r  <-> x = br6(l)<<3 | r      (levels 0..2 on x0..x2)
//                              phase B: lane l, reg r  <-> x = (l>>3)<<6 | r<<3 | (l&7)   (levels 3..5 on x3..x5)
//                              phase C: lane l, reg r  <-> x = r<<6 | l            (levels 6..8 on x6..x8)
//   input of phase A:  v[r] = p[j] + i p[j+512],  j = br9(x) = br3(r)<<6 | l
//   output of phase C: v[r] = P(omega^(2i+1)),    i = igray(x)         (reference LagrangeHalfC index i)
// =================================================================================================================
// The in-register phases (levels whose twiddles do not depend on the lane) need only seven distinct table values,
// cos/sin of pi/4, pi/8, pi/16 and 3pi/16 (the table's octant folding makes cos(pi/2 - x) and sin(x) the same
// double); they travel as kernel arguments, i.e. in scalar registers.
struct PhaseAConsts {
    double c4;        // table(8,1).re = table(8,1).im
    double c8, s8;    // table(16,1)
    double c16, s16;  // table(32,1)
    double c316, s316;  // table(32,3)
};
template <class TW>
BRC_HD PhaseAConsts make_phaseA_consts(const TW &tw) {
    PhaseAConsts k;
    k.c4 = tw(tw_idx(8, 1)).re;
    k.c8 = tw(tw_idx(16, 1)).re;
    k.s8 = tw(tw_idx(16, 1)).im;
    k.c16 = tw(tw_idx(32, 1)).re;
    k.s16 = tw(tw_idx(32, 1)).im;
    k.c316 = tw(tw_idx(32, 3)).re;
    k.s316 = tw(tw_idx(32, 3)).im;
    return k;
}

BRC_HD void fwd_phaseA(c64 (&v)[8], const PhaseAConsts &k) {
    const c64 w8 = {k.c4, k.c4};
#pragma unroll
    for (int r = 0; r < 8; r += 2) fwd_bf(v[r], v[r + 1], w8);
    const c64 w16a = {k.c8, k.s8}, w16b = {k.s8, k.c8};  // table(16,1), table(16,3)
#pragma unroll
    for (int g = 0; g < 8; g += 4) {
        fwd_bf(v[g + 0], v[g + 2], w16a);
        fwd_bf(v[g + 1], v[g + 3], w16b);
    }
    // level 2 (size 32): u = r&3 -> igray = 0,1,3,2 -> q = 1,3,7,5
    const c64 w32a = {k.c16, k.s16}, w32b = {k.c316, k.s316}, w32c = {k.s16, k.c16}, w32d = {k.s316, k.c316};
    fwd_bf(v[0], v[4], w32a);
    fwd_bf(v[1], v[5], w32b);
    fwd_bf(v[2], v[6], w32c);
    fwd_bf(v[3], v[7], w32d);
}

// slot written by lane/reg in phase A layout, read in phase B layout
BRC_HD int fwd_xA(int lane, int r) { return (br6(lane) << 3) | r; }
BRC_HD int fwd_xB(int lane, int r) { return ((lane >> 3) << 6) | (r << 3) | (lane & 7); }
BRC_HD int fwd_xC(int lane, int r) { return (r << 6) | lane; }

template <class TW>  // TW: (slot) -> this lane's entry of the per-lane twiddle image
BRC_HD void fwd_phaseB(c64 (&v)[8], const TW &tw) {
    {   // level 3, size 64: q = 2*g3+1, g3 = igray(x2x1x0)
        const c64 w = tw(TWS_FB + 0);
#pragma unroll
        for (int r = 0; r < 8; r += 2) fwd_bf(v[r], v[r + 1], w);
    }
    {   // level 4, size 128: m = x3<<3 | (g3 ^ (x3?7:0)) = g3 or 15-g3: the second is the mirror of the first
        const c64 w0 = tw(TWS_FB + 1);
        const c64 w1 = mirror(w0);
#pragma unroll
        for (int g = 0; g < 8; g += 4) {
            fwd_bf(v[g + 0], v[g + 2], w0);
            fwd_bf(v[g + 1], v[g + 3], w1);
        }
    }
    {   // level 5, size 256: m = g3, 15-g3, 31-g3, 16+g3 for x4x3 = 00, 01, 10, 11; 31-m mirrors m
        const c64 a = tw(TWS_FB + 2), d = tw(TWS_FB + 3);
        fwd_bf(v[0], v[4], a);
        fwd_bf(v[1], v[5], mirror(d));
        fwd_bf(v[2], v[6], mirror(a));
        fwd_bf(v[3], v[7], d);
    }
}

template <class TW>
BRC_HD void fwd_phaseC(c64 (&v)[8], const TW &tw) {
    {   // level 6, size 512: q = 2*g6+1, g6 = igray(x5..x0)
        const c64 w = tw(TWS_FC + 0);
#pragma unroll
        for (int r = 0; r < 8; r += 2) fwd_bf(v[r], v[r + 1], w);
    }
    {   // level 7, size 1024: m = g6 or 127-g6
        const c64 w0 = tw(TWS_FC + 1);
        const c64 w1 = mirror(w0);
#pragma unroll
        for (int g = 0; g < 8; g += 4) {
            fwd_bf(v[g + 0], v[g + 2], w0);
            fwd_bf(v[g + 1], v[g + 3], w1);
        }
    }
    {   // level 8, size 2048: m = g6, 127-g6, 255-g6, 128+g6 for x7x6 = 00, 01, 10, 11; 255-m mirrors m
        const c64 a = tw(TWS_FC + 2), d = tw(TWS_FC + 3);
        fwd_bf(v[0], v[4], a);
        fwd_bf(v[1], v[5], mirror(d));
        fwd_bf(v[2], v[6], mirror(a));
        fwd_bf(v[3], v[7], d);
    }
}

// =================================================================================================================
// INVERSE TRANSFORM, 1024 complex values, 16 per lane (reg rho = 4 bits).
//   working index y (10 bits), W[y] initially c~[br10(y)].
//   phase A: lane l, reg rho <-> y = br6(l)<<4 | rho                               (levels 0..3 on y0..y3)
//   phase B: lane l, reg rho <-> y = (l>>3)<<7 | (rho>>1)<<4 | (rho&1)<<3 | (l&7)   (levels 4..6 on y4..y6 = rho bits 1..3)
//   phase C: lane l, reg rho <-> y = (rho>>1)<<7 | (l>>3)<<4 | (rho&1)<<3 | (l&7)   (levels 7..9 on y7..y9 = rho bits 1..3)
//   output: coefficient j = y of the polynomial.
//   input of phase A: c~[m], m = br10(y) = br4(rho)<<6 | l
// =================================================================================================================
BRC_HD int inv_yA(int lane, int rho) { return (br6(lane) << 4) | rho; }
BRC_HD int inv_yB(int lane, int rho) { return ((lane >> 3) << 7) | ((rho >> 1) << 4) | ((rho & 1) << 3) | (lane & 7); }
BRC_HD int inv_yC(int lane, int rho) { return ((rho >> 1) << 7) | ((lane >> 3) << 4) | ((rho & 1) << 3) | (lane & 7); }

// inverse butterflies with the special twiddles of the in-register levels.  Each is value-identical to inv_bf with
// the table entry (products with 1, 0 and a negated cosine are exact rewrites: x*1 = x, x + y*0 = x up to the sign of
// a zero, x*(-c) = -(x*c), -a + b = b - a).
BRC_HD void inv_bf_one(c64 &E, c64 &O) {  // w = (1, 0)
    const c64 lo = {E.re + O.re, E.im + O.im}, hi = {E.re - O.re, E.im - O.im};
    E = lo;
    O = hi;
}
BRC_HD void inv_bf_mi(c64 &E, c64 &O) {  // w = (-0, 1): t = (Oi, -Or)
    const c64 lo = {E.re + O.im, E.im - O.re}, hi = {E.re - O.im, E.im + O.re};
    E = lo;
    O = hi;
}
BRC_HD void inv_bf_neg(c64 &E, c64 &O, const double c, const double s) {  // w = (-c, s)
    const double tr = O.im * s - O.re * c;     // Or*(-c) + Oi*s
    const double ti = -(O.im * c) - O.re * s;  // Oi*(-c) - Or*s
    const c64 lo = {E.re + tr, E.im + ti}, hi = {E.re - tr, E.im - ti};
    E = lo;
    O = hi;
}

BRC_HD void inv_phaseA_lo(c64 (&v)[16]) {
    // level 0 (size 2, polynomials.cpp:113-122): plain sum / difference
#pragma unroll
    for (int r = 0; r < 16; r += 2) inv_bf_one(v[r], v[r + 1]);
    // level 1 (size 4, :125-145): even index plain, odd index twiddle -i
#pragma unroll
    for (int g = 0; g < 16; g += 4) {
        inv_bf_one(v[g], v[g + 2]);
        inv_bf_mi(v[g + 1], v[g + 3]);
    }
}
BRC_HD void inv_phaseA_hi(c64 (&v)[16], const PhaseAConsts &k) {
    // level 2 (size 8): table(8, j) = (1,0), (c4,c4), (-0,1), (-c4,c4)
#pragma unroll
    for (int g = 0; g < 16; g += 8) {
        inv_bf_one(v[g + 0], v[g + 4]);
        inv_bf(v[g + 1], v[g + 5], c64{k.c4, k.c4});
        inv_bf_mi(v[g + 2], v[g + 6]);
        inv_bf_neg(v[g + 3], v[g + 7], k.c4, k.c4);
    }
    // level 3 (size 16): table(16, j) = (1,0), (c8,s8), (c4,c4), (s8,c8), (-0,1), (-s8,c8), (-c4,c4), (-c8,s8)
    inv_bf_one(v[0], v[8]);
    inv_bf(v[1], v[9], c64{k.c8, k.s8});
    inv_bf(v[2], v[10], c64{k.c4, k.c4});
    inv_bf(v[3], v[11], c64{k.s8, k.c8});
    inv_bf_mi(v[4], v[12]);
    inv_bf_neg(v[5], v[13], k.s8, k.c8);
    inv_bf_neg(v[6], v[14], k.c4, k.c4);
    inv_bf_neg(v[7], v[15], k.c8, k.s8);
}
BRC_HD void inv_phaseA(c64 (&v)[16], const PhaseAConsts &k) {
    inv_phaseA_lo(v);
    inv_phaseA_hi(v, k);
}

// rho = (y6 y5 y4 | y3), l3 = lane & 7.
template <class TW>
BRC_HD void inv_level4(c64 (&v)[16], const TW &tw) {
    // size 32: pairs differ in y4 = rho bit1; j = y3<<3 | l3, and y3 = 1 is the quarter turn of y3 = 0
    const c64 w = tw(TWS_IB + 0);
#pragma unroll
    for (int hi = 0; hi < 4; hi++) {  // y6 y5
        inv_bf(v[hi << 2], v[(hi << 2) | 2], w);
        inv_bf_q(v[(hi << 2) | 1], v[(hi << 2) | 3], w);
    }
}
template <class TW>
BRC_HD void inv_level5(c64 (&v)[16], const TW &tw) {
    // size 64: pairs differ in y5 = rho bit2; j = y4<<4 | y3<<3 | l3; y4 = 1 is the quarter turn
#pragma unroll
    for (int y3 = 0; y3 < 2; y3++) {
        const c64 w = tw(TWS_IB + 1 + y3);
#pragma unroll
        for (int y6 = 0; y6 < 2; y6++) {
            inv_bf(v[(y6 << 3) | y3], v[(y6 << 3) | y3 | 4], w);
            inv_bf_q(v[(y6 << 3) | 2 | y3], v[(y6 << 3) | 2 | y3 | 4], w);
        }
    }
}
template <class TW>
BRC_HD void inv_level6(c64 (&v)[16], const TW &tw) {
    // size 128: pairs differ in y6 = rho bit3; j = (y5 y4 y3)<<3 | l3; y5 = 1 is the quarter turn
#pragma unroll
    for (int lo = 0; lo < 4; lo++) {  // y4 y3
        const c64 w = tw(TWS_IB + 3 + lo);
        inv_bf(v[lo], v[lo | 8], w);
        inv_bf_q(v[lo | 4], v[lo | 12], w);
    }
}
template <class TW>
BRC_HD void inv_phaseB(c64 (&v)[16], const TW &tw) {
    inv_level4(v, tw);
    inv_level5(v, tw);
    inv_level6(v, tw);
}

// rho = (y9 y8 y7 | y3); lane = (y6 y5 y4 | y2 y1 y0); low 7 bits of y: (lane>>3)<<4 | y3<<3 | (lane&7)
template <class TW>
BRC_HD void inv_level7(c64 (&v)[16], const TW &tw) {
    // size 256: pairs differ in y7 = rho bit1; j = y6..y0 (the quarter-turn bit y6 is a lane bit: no sharing)
#pragma unroll
    for (int y3 = 0; y3 < 2; y3++) {
        const c64 w = tw(TWS_IC + y3);
#pragma unroll
        for (int hi = 0; hi < 4; hi++) {
            const int r0 = (hi << 2) | y3;
            inv_bf(v[r0], v[r0 | 2], w);
        }
    }
}
template <class TW>
BRC_HD void inv_level8(c64 (&v)[16], const TW &tw) {
    // size 512: pairs differ in y8 = rho bit2; j = y7<<7 | low7; y7 = 1 is the quarter turn
#pragma unroll
    for (int y3 = 0; y3 < 2; y3++) {
        const c64 w = tw(TWS_IC + 2 + y3);
#pragma unroll
        for (int y9 = 0; y9 < 2; y9++) {
            inv_bf(v[(y9 << 3) | y3], v[(y9 << 3) | y3 | 4], w);
            inv_bf_q(v[(y9 << 3) | 2 | y3], v[(y9 << 3) | 2 | y3 | 4], w);
        }
    }
}
template <class TW>
BRC_HD void inv_level9(c64 (&v)[16], const TW &tw) {
    // size 1024: pairs differ in y9 = rho bit3; j = (y8 y7)<<7 | low7; y8 = 1 is the quarter turn
#pragma unroll
    for (int lo = 0; lo < 4; lo++) {  // y7 y3
        const c64 w = tw(TWS_IC + 4 + lo);
        inv_bf(v[lo], v[lo | 8], w);
        inv_bf_q(v[lo | 4], v[lo | 12], w);
    }
}
template <class TW>
BRC_HD void inv_phaseC(c64 (&v)[16], const TW &tw) {
    inv_level7(v, tw);
    inv_level8(v, tw);
    inv_level9(v, tw);
}

// =================================================================================================================
// INVERSE TRANSFORM SPLIT OVER TWO WAVES (latency kernel): wave h = 0, 1 produces the coefficients y with y0 = h.
//   Level 0 of the transform above pairs W[2y'] = c~[m0] and W[2y'+1] = c~[m0 + 512] = conj(C[511 - m0]), m0 = br9(y'):
//   wave 0 forms the sums, wave 1 the differences, both straight from the spectrum - no exchange between the waves, no
//   butterfly computed twice.  Levels 1..9 then stay inside {y : y0 = h}: on y' = y >> 1 (9 bits) they are levels
//   k' = 0..8 with twiddle table(2^(k'+2), 2 j' + h), j' = y' mod 2^k'.  Layouts of y' = the forward transform's
//   (fwd_xA / fwd_xB / fwd_xC, 8 values per lane), so the two transposes are the forward ones.
//   Output of lane l, reg r: coefficient y = 2 (r<<6 | l) + h = r<<7 | l<<1 | h.
//   Per-lane twiddle image: 12 slots per h (half_slot_index); quarter-turn partners folded as above.
// =================================================================================================================
constexpr int kTwHalfSlots = 12;
enum { TWH_B = 0, TWH_C = 4, TWH_FIN = 8 };
BRC_HD int half_slot_index(int h, int slot, int lane) {
    const int l3 = lane & 7;
    if (slot == 0) return tw_idx(32, 2 * l3 + h);                                  // k' = 3
    if (slot == 1) return tw_idx(64, 2 * l3 + h);                                  // k' = 4, y'3 = 0
    if (slot < 4) return tw_idx(128, 2 * (((slot - 2) << 3) | l3) + h);            // k' = 5, y'4 = 0, y'3 = slot-2
    if (slot == 4) return tw_idx(256, 2 * lane + h);                               // k' = 6
    if (slot == 5) return tw_idx(512, 2 * lane + h);                               // k' = 7, y'6 = 0
    if (slot < 8) return tw_idx(1024, 2 * (((slot - 6) << 6) | lane) + h);         // k' = 8, y'7 = 0, y'6 = slot-6
    return tw_idx(2048, ((slot - 8) << 7) | (lane << 1) | h);                      // last merge, r = slot-8 < 4
}
template <class BASE>
inline void build_half_twiddles(c64 *out /*[2][kTwHalfSlots][64]*/, const BASE &base) {
    for (int h = 0; h < 2; h++)
        for (int s = 0; s < kTwHalfSlots; s++)
            for (int l = 0; l < 64; l++) out[(h * kTwHalfSlots + s) * 64 + l] = base(half_slot_index(h, s, l));
}
BRC_HD int half_y(int lane, int r, int h) { return (r << 7) | (lane << 1) | h; }

// E = C[m0], O = conj(C[511 - m0]) on entry (m0 = br3(r)<<6 | lane); level 0 and the three in-register levels
template <int H>
BRC_HD void half_phaseA(c64 (&v)[8], const c64 (&E)[8], const c64 (&O)[8], const PhaseAConsts &k) {
#pragma unroll
    for (int r = 0; r < 8; r++)  // polynomials.cpp:113-122: sum (y0 = 0) / difference (y0 = 1)
        v[r] = H == 0 ? c64{E[r].re + O[r].re, E[r].im + O[r].im} : c64{E[r].re - O[r].re, E[r].im - O[r].im};
    // k' = 0 (size 4): table(4, h) = 1 or -i
#pragma unroll
    for (int r = 0; r < 8; r += 2) {
        if (H == 0) inv_bf_one(v[r], v[r + 1]);
        else inv_bf_mi(v[r], v[r + 1]);
    }
    // k' = 1 (size 8): table(8, 2 (r&1) + h)
#pragma unroll
    for (int g = 0; g < 8; g += 4) {
        if (H == 0) {
            inv_bf_one(v[g], v[g + 2]);
            inv_bf_mi(v[g + 1], v[g + 3]);
        } else {
            inv_bf(v[g], v[g + 2], c64{k.c4, k.c4});
            inv_bf_neg(v[g + 1], v[g + 3], k.c4, k.c4);
        }
    }
    // k' = 2 (size 16): table(16, 2 (r&3) + h)
    if (H == 0) {
        inv_bf_one(v[0], v[4]);
        inv_bf(v[1], v[5], c64{k.c4, k.c4});
        inv_bf_mi(v[2], v[6]);
        inv_bf_neg(v[3], v[7], k.c4, k.c4);
    } else {
        inv_bf(v[0], v[4], c64{k.c8, k.s8});
        inv_bf(v[1], v[5], c64{k.s8, k.c8});
        inv_bf_neg(v[2], v[6], k.s8, k.c8);
        inv_bf_neg(v[3], v[7], k.c8, k.s8);
    }
}
// three levels on the register bits; TW: (slot) -> this lane's entry of the image of this wave's h; BASE = TWH_B / TWH_C
template <int BASE, class TW>
BRC_HD void half_phaseBC(c64 (&v)[8], const TW &tw) {
    {
        const c64 w = tw(BASE + 0);
#pragma unroll
        for (int r = 0; r < 8; r += 2) inv_bf(v[r], v[r + 1], w);
    }
    {
        const c64 w = tw(BASE + 1);
#pragma unroll
        for (int g = 0; g < 8; g += 4) {
            inv_bf(v[g], v[g + 2], w);
            inv_bf_q(v[g + 1], v[g + 3], w);
        }
    }
    {
        const c64 a = tw(BASE + 2), b = tw(BASE + 3);
        inv_bf(v[0], v[4], a);
        inv_bf(v[1], v[5], b);
        inv_bf_q(v[2], v[6], a);
        inv_bf_q(v[3], v[7], b);
    }
}

// last merge (size 2048) + scaling + truncation: polynomials.cpp:160-166 with E = 0, then :392.
// BOUNDED = true may be used when |value| < 2^51 is guaranteed (inside the external product the value is a sum of
// 6*1024 products of a digit in [-64,63] and a torus real in [-1/2,1/2), scaled by 2^32: below 2^50).  Then
// trunc(x) + 1.5*2^52 is exact and the low mantissa word is trunc(x) mod 2^32 - the same bits as the reference's
// Torus32(int64_t(x)) - at 2 instructions instead of a full f64 -> i64 conversion.
BRC_HD int32_t trunc_to_torus32_bounded(double x) {
    const double t = __builtin_trunc(x) + 6755399441055744.0;
    uint64_t u;
    __builtin_memcpy(&u, &t, 8);
    return (int32_t) (uint32_t) u;
}
template <bool BOUNDED>
BRC_HD int32_t scale_to_torus32(double tr) {
    const double scaled = tr * 4194304.0;  // (tr * 2^-10) * 2^32, both exact scalings
    return BOUNDED ? trunc_to_torus32_bounded(scaled) : (int32_t) (uint32_t) (uint64_t) (int64_t) scaled;
}
// output rho of lane l is coefficient y = inv_yC(l, rho); twiddle table(2048, y); rho >= 8 (y9 = 1) is the quarter
// turn of rho - 8:  Wr*(-s) + Wi*c = Wi*c - Wr*s exactly
template <bool BOUNDED, class TW>
BRC_HD void inv_final(int32_t (&out)[16], const c64 (&v)[16], const TW &tw) {
#pragma unroll
    for (int rho = 0; rho < 8; rho++) {
        const c64 w = tw(TWS_FIN + rho);
        out[rho] = scale_to_torus32<BOUNDED>(v[rho].re * w.re + v[rho].im * w.im);
        out[rho + 8] = scale_to_torus32<BOUNDED>(v[rho + 8].im * w.re - v[rho + 8].re * w.im);
    }
}
// the same for the split transform: output r of lane l is coefficient half_y(l, r, h); r >= 4 (y9 = 1) is the quarter turn
template <bool BOUNDED, class TW>
BRC_HD void half_final(int32_t (&out)[8], const c64 (&v)[8], const TW &tw) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const c64 w = tw(TWH_FIN + r);
        out[r] = scale_to_torus32<BOUNDED>(v[r].re * w.re + v[r].im * w.im);
        out[r + 4] = scale_to_torus32<BOUNDED>(v[r + 4].im * w.re - v[r + 4].re * w.im);
    }
}

// ---- integer pieces -------------------------------------------------------------------------------------------
// ((X^a - 1) * acc)[j] for a in [0, 2N): reference toruspolynomial-functions.cpp:106-124, all cases folded into
// one index computation: source index (j - a) mod 2N, negated when it falls in the upper half
BRC_HD int32_t rot_minus_one(const int32_t *acc, int a, int j) {
    const int s = (j - a) & (2 * kN - 1);
    const uint32_t v = (uint32_t) acc[s & (kN - 1)];
    const uint32_t rot = (s & kN) ? (0u - v) : v;
    return (int32_t) (rot - (uint32_t) acc[j]);
}
// gadget digit p (0..l-1) of a Torus32: reference tgsw-functions.cpp:297-315:
//   u = t + offset;  digit_p = ((u >> (32 - Bgbit(p+1))) & (Bg-1)) - Bg/2
// field - Bg/2 equals the field read as a signed Bgbit-bit number after flipping its top bit, so the digits of one
// coefficient cost one add, one xor (decomp_prepare, once) and one signed bit-field extract each (decomp_digit).
BRC_HD uint32_t decomp_prepare(int32_t t) { return ((uint32_t) t + kDecompOffset) ^ kDecompOffset; }
BRC_HD int32_t decomp_digit(uint32_t prepared, int p) {
    return ((int32_t) (prepared << (kBgbit * p))) >> (32 - kBgbit);
}
// the same as one v_bfe_i32 when p is not a compile-time constant (the shift pair costs two instructions then).  Used
// by the throughput kernel (-96 VALU instructions per CMux, +0.3 % measured); the latency kernel, whose register
// allocation sits at the limit, measured 3.7 % slower with it and keeps the shifts.
BRC_HD int32_t decomp_digit_rt(uint32_t prepared, int p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sbfe((int32_t) prepared, 32 - kBgbit * (p + 1), kBgbit);
#else
    return decomp_digit(prepared, p);
#endif
}
// modSwitchFromTorus32(x, 2N): reference numeric-functions.cpp:59-65 (2^64/2N = 2^53 wide intervals)
BRC_HD int32_t modswitch_2N(int32_t x) { return (int32_t) ((((uint32_t) x) + (1u << 20)) >> 21) & (2 * kN - 1); }

}  // namespace brc

// Host emulation of one wavefront running the per-lane routines of tfhe_amd/csrc/fast_core.h (the FAST arithmetic
// mode): 64 lanes in a loop, phase by phase, LDS scratch as a plain array.  Checked against the oracle within the
// tolerances of the reference's own stage tests (src/test/test-MulToTLwe.cpp:365-405: forward spectrum 1e-10 per unit
// of input magnitude 64; src/test/test-multiplication.cpp:66-77: products within 1 LSB).  Built and run by
// tests/test_emu_wave.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../../tfhe_amd/csrc/fast_core.h"
#include "../../oracle/tfhe_oracle.h"
using namespace brc;

static std::vector<c64> g_img(brf::kFastLaneEntries);
static PhaseAConsts g_k;
struct TWL { int lane; c64 operator()(int slot) const { return g_img[slot * 64 + lane]; } };

static void emu_forward(const double *p, c64 *Z /*[512], Z_k*/) {
    static c64 v[64][8], scratch[512];
    for (int l = 0; l < 64; l++) for (int r = 0; r < 8; r++) { int j = (br3(r) << 6) | l; v[l][r] = {p[j], p[j + 512]}; }
    for (int l = 0; l < 64; l++) brf::fwd_phaseA(v[l], g_k);
    for (int l = 0; l < 64; l++) for (int r = 0; r < 8; r++) scratch[swz(fwd_xA(l, r))] = v[l][r];
    for (int l = 0; l < 64; l++) for (int r = 0; r < 8; r++) v[l][r] = scratch[swz(fwd_xB(l, r))];
    for (int l = 0; l < 64; l++) brf::fwd_phaseBC<brf::FS_FB>(v[l], TWL{l});
    for (int l = 0; l < 64; l++) for (int r = 0; r < 8; r++) scratch[swz(fwd_xB(l, r))] = v[l][r];
    for (int l = 0; l < 64; l++) for (int r = 0; r < 8; r++) v[l][r] = scratch[swz(fwd_xC(l, r))];
    for (int l = 0; l < 64; l++) brf::fwd_phaseBC<brf::FS_FC>(v[l], TWL{l});
    for (int l = 0; l < 64; l++) for (int r = 0; r < 8; r++) Z[(r << 6) | l] = v[l][r];
}
static void emu_inverse(const c64 *Y /*[512]*/, int32_t *out, bool bounded) {
    static c64 v[64][8], scratch[512];
    for (int l = 0; l < 64; l++) for (int r = 0; r < 8; r++) v[l][r] = Y[(br3(r) << 6) | l];  // register renaming only
    for (int l = 0; l < 64; l++) brf::inv_phaseA(v[l], g_k);
    for (int l = 0; l < 64; l++) for (int r = 0; r < 8; r++) scratch[swz(fwd_xA(l, r))] = v[l][r];
    for (int l = 0; l < 64; l++) for (int r = 0; r < 8; r++) v[l][r] = scratch[swz(fwd_xB(l, r))];
    for (int l = 0; l < 64; l++) brf::inv_phaseBC<brf::FS_IB>(v[l], TWL{l});
    for (int l = 0; l < 64; l++) for (int r = 0; r < 8; r++) scratch[swz(fwd_xB(l, r))] = v[l][r];
    for (int l = 0; l < 64; l++) for (int r = 0; r < 8; r++) v[l][r] = scratch[swz(fwd_xC(l, r))];
    for (int l = 0; l < 64; l++) brf::inv_phaseBC<brf::FS_IC>(v[l], TWL{l});
    for (int l = 0; l < 64; l++) {
        int32_t lo[8], hi[8];
        if (bounded) brf::inv_final<true>(lo, hi, v[l], TWL{l}); else brf::inv_final<false>(lo, hi, v[l], TWL{l});
        for (int r = 0; r < 8; r++) { out[(r << 6) | l] = lo[r]; out[((r << 6) | l) + 512] = hi[r]; }
    }
}
// reference (LagrangeHalfC) order <-> Z_k
static void to_ref_order(const c64 *Z, double *C) {
    for (int i = 0; i < 512; i++) {
        const c64 z = Z[brf::ref_slot_source(i)];
        C[2 * i] = z.re;
        C[2 * i + 1] = (i & 1) ? -z.im : z.im;
    }
}
static void from_ref_order(const double *C, c64 *Y) {
    for (int k = 0; k < 512; k++) {
        if (k < 256) Y[k] = {C[4 * k], C[4 * k + 1]};
        else { const int i = 1023 - 2 * k; Y[k] = {C[2 * i], -C[2 * i + 1]}; }
    }
}

int main() {
    brf::build_fast_twiddles(g_img.data());
    g_k.c4 = std::sqrt(0.5);
    g_k.c8 = std::cos(M_PI / 8); g_k.s8 = std::sin(M_PI / 8);
    g_k.c16 = std::cos(M_PI / 16); g_k.s16 = std::sin(M_PI / 16);
    g_k.c316 = std::cos(3 * M_PI / 16); g_k.s316 = std::sin(3 * M_PI / 16);
    std::mt19937_64 rng(321);
    int bad = 0;
    double worst_fwd = 0, worst_tor = 0;
    long worst_inv = 0, worst_mul = 0, ndiff_inv = 0, ntot_inv = 0;
    for (int trial = 0; trial < 200; trial++) {
        int32_t p[1024]; double pd[1024], ref[1024], C[1024]; c64 Z[512];
        for (int i = 0; i < 1024; i++) { p[i] = (int) (rng() % 128) - 64; pd[i] = p[i]; }
        oracle_IntPolynomial_ifft(ref, p);
        emu_forward(pd, Z);
        to_ref_order(Z, C);
        for (int i = 0; i < 1024; i++) worst_fwd = std::fmax(worst_fwd, std::fabs(C[i] - ref[i]));
        int32_t t[1024];
        for (int i = 0; i < 1024; i++) { t[i] = (int32_t) rng(); pd[i] = t[i] * (1. / 4294967296.); }
        oracle_TorusPolynomial_ifft(ref, t);
        emu_forward(pd, Z);
        to_ref_order(Z, C);
        for (int i = 0; i < 1024; i++) worst_tor = std::fmax(worst_tor, std::fabs(C[i] - ref[i]));
        // inverse of a product-like spectrum (the oracle's own spectrum, scaled): coefficients within 1 of the oracle's
        const double scale = (double) (1 + rng() % 100000);
        double Cs[1024]; c64 Y[512];
        for (int i = 0; i < 1024; i++) Cs[i] = ref[i] * scale;
        int32_t r1[1024], r2[1024];
        oracle_TorusPolynomial_fft(r1, Cs);
        from_ref_order(Cs, Y);
        for (int b = 0; b < 2; b++) {
            if (b == 1) {  // bounded truncation: output below 2^50
                for (int i = 0; i < 1024; i++) Cs[i] = ref[i] * 2000.0;
                oracle_TorusPolynomial_fft(r1, Cs);
                from_ref_order(Cs, Y);
            }
            emu_inverse(Y, r2, b == 1);
            for (int i = 0; i < 1024; i++) {
                const long d = std::labs((long) (int32_t) ((uint32_t) r1[i] - (uint32_t) r2[i]));
                worst_inv = std::max(worst_inv, d);
                ndiff_inv += d != 0;
                ntot_inv++;
            }
        }
        // product against the exact (naive) product: test-multiplication.cpp:41-44,66-77
        int32_t a[1024], bq[1024], exact[1024], got[1024];
        double ad[1024], bd[1024]; c64 A[512], B[512], Pz[512];
        for (int i = 0; i < 1024; i++) { a[i] = (int) (rng() % 4096) - 2048; bq[i] = (int32_t) rng(); ad[i] = a[i]; bd[i] = bq[i] * (1. / 4294967296.); }
        oracle_mulNaive(exact, a, bq, 1024);
        emu_forward(ad, A);
        emu_forward(bd, B);
        for (int k = 0; k < 512; k++) Pz[k] = brf::cmul(A[k], B[k]);
        emu_inverse(Pz, got, false);
        for (int i = 0; i < 1024; i++)
            worst_mul = std::max(worst_mul, std::labs((long) (int32_t) ((uint32_t) exact[i] - (uint32_t) got[i])));
    }
    printf("forward |delta| max %.3g (int digits), %.3g (torus); inverse max |delta| %ld LSB (%ld of %ld differ); product vs exact max %ld LSB\n",
           worst_fwd, worst_tor, worst_inv, ndiff_inv, ntot_inv, worst_mul);
    if (worst_fwd > 1e-10 || worst_tor > 1e-12 || worst_inv > 1 || worst_mul > 1) bad++;
    printf(bad ? "EMU FAST FAIL (%d)\n" : "EMU FAST OK\n", bad);
    return bad != 0;
}

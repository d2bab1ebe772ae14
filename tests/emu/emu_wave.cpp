// Host emulation of one wavefront running the per-lane routines of tfhe_amd/csrc/br_core.h (64 lanes in a loop,
// phase by phase, LDS scratch as a plain array), checked bit-for-bit against the oracle.  Development aid for
// validating the kernel arithmetic in the GPU-less authoring container; built and run by tests/test_emu_wave.py.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#include <climits>
#include "../../tfhe_amd/csrc/br_core.h"
#include "../../oracle/tfhe_oracle.h"
using namespace brc;

static std::vector<c64> g_tw(kTwEntries);
static std::vector<c64> g_lane_tw(kTwLaneEntries);
static std::vector<c64> g_half_tw(2 * kTwHalfSlots * 64);
struct TWH { int h, lane; c64 operator()(int slot) const { return g_half_tw[(h * kTwHalfSlots + slot) * 64 + lane]; } };
struct TW { c64 operator()(int i) const { return g_tw[i]; } };        // base tables
struct TWL { int lane; c64 operator()(int slot) const { return g_lane_tw[slot * 64 + lane]; } };  // per-lane image
static void init_tw() {
    int L = oracle_trig_table_len();
    std::vector<double> rev(L), dir(L);
    oracle_get_trig_tables(rev.data(), dir.data());
    long off = 0;
    for (int size = 8; size <= 2048; size *= 2) {
        for (int j = 0; j < size / 2; j++) {
            int blk = j / 4, q = j % 4;
            g_tw[tw_idx(size, j)] = {dir[off + 8 * blk + q], dir[off + 8 * blk + 4 + q]};
        }
        off += size;
    }
    build_lane_twiddles(g_lane_tw.data(), TW{});
    build_half_twiddles(g_half_tw.data(), TW{});
}

static void emu_forward(const double *p, c64 *logical_out) {
    static c64 v[64][8], scratch[512];
    TW tw;
    for (int l = 0; l < 64; l++) for (int r = 0; r < 8; r++) { int j = (br3(r) << 6) | l; v[l][r] = {p[j], p[j + 512]}; }
    const PhaseAConsts kc = make_phaseA_consts(tw);
    for (int l = 0; l < 64; l++) fwd_phaseA(v[l], kc);
    for (int l = 0; l < 64; l++) for (int r = 0; r < 8; r++) scratch[swz(fwd_xA(l, r))] = v[l][r];
    for (int l = 0; l < 64; l++) for (int r = 0; r < 8; r++) v[l][r] = scratch[swz(fwd_xB(l, r))];
    for (int l = 0; l < 64; l++) fwd_phaseB(v[l], TWL{l});
    for (int l = 0; l < 64; l++) for (int r = 0; r < 8; r++) scratch[swz(fwd_xB(l, r))] = v[l][r];
    for (int l = 0; l < 64; l++) for (int r = 0; r < 8; r++) v[l][r] = scratch[swz(fwd_xC(l, r))];
    for (int l = 0; l < 64; l++) fwd_phaseC(v[l], TWL{l});
    for (int l = 0; l < 64; l++) for (int r = 0; r < 8; r++) logical_out[igray(fwd_xC(l, r))] = v[l][r];
}

static void emu_inverse(const c64 *C /*logical*/, int32_t *out, bool bounded) {
    static c64 v[64][16], scratch[1024];
    TW tw;
    for (int l = 0; l < 64; l++) for (int rho = 0; rho < 16; rho++) {
        int m = (br4(rho) << 6) | l;
        v[l][rho] = m < 512 ? C[m] : c64{C[1023 - m].re, -C[1023 - m].im};
    }
    const PhaseAConsts kc = make_phaseA_consts(tw);
    for (int l = 0; l < 64; l++) inv_phaseA(v[l], kc);
    for (int l = 0; l < 64; l++) for (int r = 0; r < 16; r++) scratch[inv_yA(l, r)] = v[l][r];
    for (int l = 0; l < 64; l++) for (int r = 0; r < 16; r++) v[l][r] = scratch[inv_yB(l, r)];
    for (int l = 0; l < 64; l++) inv_phaseB(v[l], TWL{l});
    for (int l = 0; l < 64; l++) for (int r = 0; r < 16; r++) scratch[inv_yB(l, r)] = v[l][r];
    for (int l = 0; l < 64; l++) for (int r = 0; r < 16; r++) v[l][r] = scratch[inv_yC(l, r)];
    for (int l = 0; l < 64; l++) inv_phaseC(v[l], TWL{l});
    for (int l = 0; l < 64; l++) { int32_t o[16]; if (bounded) inv_final<true>(o, v[l], TWL{l}); else inv_final<false>(o, v[l], TWL{l}); for (int r = 0; r < 16; r++) out[inv_yC(l, r)] = o[r]; }
}

// the inverse transform split over two waves (latency kernel): wave h produces the coefficients with y0 = h
static void emu_inverse_split(const c64 *C /*logical*/, int32_t *out, bool bounded) {
    TW tw;
    const PhaseAConsts kc = make_phaseA_consts(tw);
    for (int h = 0; h < 2; h++) {
        static c64 v[64][8], scratch[512];
        for (int l = 0; l < 64; l++) {
            c64 E[8], O[8];
            for (int r = 0; r < 8; r++) {
                const int m0 = (br3(r) << 6) | l;
                E[r] = C[m0];
                O[r] = {C[511 - m0].re, -C[511 - m0].im};
            }
            if (h == 0) half_phaseA<0>(v[l], E, O, kc); else half_phaseA<1>(v[l], E, O, kc);
        }
        for (int l = 0; l < 64; l++) for (int r = 0; r < 8; r++) scratch[swz(fwd_xA(l, r))] = v[l][r];
        for (int l = 0; l < 64; l++) for (int r = 0; r < 8; r++) v[l][r] = scratch[swz(fwd_xB(l, r))];
        for (int l = 0; l < 64; l++) half_phaseBC<TWH_B>(v[l], TWH{h, l});
        for (int l = 0; l < 64; l++) for (int r = 0; r < 8; r++) scratch[swz(fwd_xB(l, r))] = v[l][r];
        for (int l = 0; l < 64; l++) for (int r = 0; r < 8; r++) v[l][r] = scratch[swz(fwd_xC(l, r))];
        for (int l = 0; l < 64; l++) half_phaseBC<TWH_C>(v[l], TWH{h, l});
        for (int l = 0; l < 64; l++) {
            int32_t o[8];
            if (bounded) half_final<true>(o, v[l], TWH{h, l}); else half_final<false>(o, v[l], TWH{h, l});
            for (int r = 0; r < 8; r++) out[half_y(l, r, h)] = o[r];
        }
    }
}

int main() {
    init_tw();
    std::mt19937_64 rng(123);
    int bad = 0;
    for (int trial = 0; trial < 200; trial++) {
        int32_t p[1024]; double pd[1024];
        for (int i = 0; i < 1024; i++) { p[i] = (int) (rng() % 128) - 64; pd[i] = p[i]; }
        double ref[1024]; c64 got[512];
        oracle_IntPolynomial_ifft(ref, p);
        emu_forward(pd, got);
        if (memcmp(ref, got, sizeof(ref)) != 0) {
            int nd = 0; for (int i = 0; i < 512; i++) if (ref[2*i] != got[i].re || ref[2*i+1] != got[i].im) nd++;
            if (nd) { bad++; if (bad < 4) printf("forward trial %d: %d values differ\n", trial, nd); }
        }
        // torus version: scaled input
        int32_t t[1024];
        for (int i = 0; i < 1024; i++) { t[i] = (int32_t) rng(); pd[i] = t[i] * (1. / 4294967296.); }
        oracle_TorusPolynomial_ifft(ref, t);
        emu_forward(pd, got);
        { int nd = 0; for (int i = 0; i < 512; i++) if (ref[2*i] != got[i].re || ref[2*i+1] != got[i].im) nd++;
          if (nd) { bad++; if (bad < 4) printf("forward(torus) trial %d: %d values differ\n", trial, nd); } }
        // inverse on a product-like spectrum
        c64 C[512]; double Cd[1024];
        double scale = (double) (1 + rng() % 100000);
        for (int i = 0; i < 512; i++) { C[i] = {got[i].re * scale, got[i].im * scale}; Cd[2*i] = C[i].re; Cd[2*i+1] = C[i].im; }
        int32_t r1[1024], r2[1024];
        oracle_TorusPolynomial_fft(r1, Cd);
        emu_inverse(C, r2, false);
        { int nd = 0; for (int i = 0; i < 1024; i++) if (r1[i] != r2[i]) nd++;
          if (nd) { bad++; if (bad < 8) printf("inverse trial %d: %d coefficients differ\n", trial, nd); } }
        emu_inverse_split(C, r2, false);
        { int nd = 0; for (int i = 0; i < 1024; i++) if (r1[i] != r2[i]) nd++;
          if (nd) { bad++; if (bad < 8) printf("split inverse trial %d: %d coefficients differ\n", trial, nd); } }
        // bounded-range truncation (valid below 2^51): spectrum scaled so that the output stays below 2^50
        for (int i = 0; i < 512; i++) { C[i] = {got[i].re * 2000.0, got[i].im * 2000.0}; Cd[2*i] = C[i].re; Cd[2*i+1] = C[i].im; }
        oracle_TorusPolynomial_fft(r1, Cd);
        emu_inverse(C, r2, true);
        { int nd = 0; for (int i = 0; i < 1024; i++) if (r1[i] != r2[i]) nd++;
          if (nd) { bad++; if (bad < 8) printf("inverse(bounded) trial %d: %d coefficients differ\n", trial, nd); } }
        emu_inverse_split(C, r2, true);
        { int nd = 0; for (int i = 0; i < 1024; i++) if (r1[i] != r2[i]) nd++;
          if (nd) { bad++; if (bad < 8) printf("split inverse(bounded) trial %d: %d coefficients differ\n", trial, nd); } }
    }
    {   // gadget decomposition: prepared-word form == reference form (tgsw-functions.cpp:297-315)
        int32_t t[1024], want[3 * 1024];
        for (int i = 0; i < 1024; i++) t[i] = (int32_t) rng();
        t[0] = 0; t[1] = -1; t[2] = INT32_MIN; t[3] = INT32_MAX; t[4] = (int32_t) (0u - kDecompOffset);
        oracle_decompH(want, t);
        for (int p = 0; p < 3; p++)
            for (int i = 0; i < 1024; i++)
                if (decomp_digit(decomp_prepare(t[i]), p) != want[p * 1024 + i]) { bad++; break; }
        // (X^a - 1) * acc, modulus switch
        for (int a : {0, 1, 511, 1023, 1024, 1025, 2047}) {
            int32_t w[1024];
            oracle_mulByXaiMinusOne(w, a, t);
            for (int j = 0; j < 1024; j++) if (rot_minus_one(t, a, j) != w[j]) { bad++; break; }
        }
        for (int i = 0; i < 100000; i++) {
            int32_t x = (int32_t) rng();
            if (modswitch_2N(x) != oracle_modSwitchFromTorus32(x, 2048)) { bad++; break; }
        }
    }
    printf(bad ? "EMU FAIL (%d)\n" : "EMU OK\n", bad);
    return bad != 0;
}

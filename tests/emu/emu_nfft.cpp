// Host emulation of one wavefront running the per-lane routines of tfhe_amd/csrc/nfft_core.h (transforms with the ring
// dimension as a parameter: N = 512, 1024, 2048): 64 lanes in a loop, phase by phase, the LDS buffer as a plain array.
// Checks: forward spectrum against the direct evaluation P(omega^(4k+1)) in long double (|delta| <= 1e-10 per unit of
// digit magnitude 64, src/test/test-MulToTLwe.cpp:365-371), and the product of an IntPolynomial with a TorusPolynomial
// against the reference's exact schoolbook product (torusPolynomialMultNaive_aux, multiplication.cpp:31-43, restated in
// oracle_mulNaive) within the tolerance of src/test/test-multiplication.cpp:66-77 (operands as :41-44).
// Built and run by tests/test_emu_wave.py.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../../tfhe_amd/csrc/nfft_core.h"
#include "../../oracle/tfhe_oracle.h"
using namespace brc;

template <int LH, bool INV>
static void emu_transform(std::vector<c64> &in /*phase-0 order: [lane][r]*/, const c64 *tables) {
    using D = nfft::Dim<LH>;
    std::vector<c64> scratch(D::NH);
    const c64 *tw = tables + (INV ? D::NH / 2 : 0);
    for (int p = 0; p < D::PHASES; p++) {
        for (int l = 0; l < 64; l++) {
            c64(&v)[D::VPL] = *reinterpret_cast<c64(*)[D::VPL]>(&in[l * D::VPL]);
            nfft::phase<LH, INV>(v, p, l, tw);
        }
        if (p + 1 < D::PHASES) {
            for (int l = 0; l < 64; l++)
                for (int r = 0; r < D::VPL; r++)
                    scratch[nfft::swz<LH>(nfft::position<LH>(p, l, nfft::bitrev(l, 6), r))] = in[l * D::VPL + r];
            for (int l = 0; l < 64; l++)
                for (int r = 0; r < D::VPL; r++)
                    in[l * D::VPL + r] = scratch[nfft::swz<LH>(nfft::position<LH>(p + 1, l, nfft::bitrev(l, 6), r))];
        }
    }
}
template <int LH>
static void emu_forward(const double *p /*[N]*/, std::vector<c64> &Z /*[NH]*/, const c64 *tables) {
    using D = nfft::Dim<LH>;
    std::vector<c64> v(64 * D::VPL);
    for (int l = 0; l < 64; l++)
        for (int r = 0; r < D::VPL; r++) {
            const int j = (nfft::bitrev(r, D::R) << 6) | l;
            v[l * D::VPL + r] = {p[j], p[j + D::NH]};
        }
    emu_transform<LH, false>(v, tables);
    Z.resize(D::NH);
    for (int l = 0; l < 64; l++)
        for (int r = 0; r < D::VPL; r++) Z[(r << 6) | l] = v[l * D::VPL + r];
}
template <int LH>
static void emu_inverse(const std::vector<c64> &Y, int32_t *out /*[N]*/, const c64 *tables) {
    using D = nfft::Dim<LH>;
    std::vector<c64> v(64 * D::VPL);
    for (int l = 0; l < 64; l++)
        for (int r = 0; r < D::VPL; r++) v[l * D::VPL + r] = Y[(nfft::bitrev(r, D::R) << 6) | l];  // register renaming only
    emu_transform<LH, true>(v, tables);
    for (int l = 0; l < 64; l++)
        for (int r = 0; r < D::VPL; r++) {
            const int j = (r << 6) | l;
            const c64 u = tables[D::NH + j], w = v[l * D::VPL + r];
            out[j] = brf::to_torus32<false>(brf::fmad(w.re, u.re, -(w.im * u.im)));
            out[j + D::NH] = brf::to_torus32<false>(brf::fmad(w.re, u.im, w.im * u.re));
        }
}

template <int LH>
static int run() {
    using D = nfft::Dim<LH>;
    std::vector<c64> tables(D::kTableEntries);
    nfft::build_tables<LH>(tables.data());
    std::mt19937_64 rng(1234 + LH);
    int bad = 0;
    // forward against the direct evaluation
    {
        std::vector<double> p(D::N);
        for (auto &x : p) x = (double) ((int) (rng() % 128) - 64);
        std::vector<c64> Z;
        emu_forward<LH>(p.data(), Z, tables.data());
        const long double pi = 3.14159265358979323846264338327950288L;
        double worst = 0;
        for (int k = 0; k < D::NH; k += 37) {
            long double re = 0, im = 0;
            for (int j = 0; j < D::N; j++) {
                const long double a = pi * (long double) (((long long) j * (4 * k + 1)) % (2 * D::N)) / D::N;
                re += p[j] * cosl(a);
                im += p[j] * sinl(a);
            }
            worst = std::fmax(worst, std::fmax(std::fabs((double) (re - Z[k].re)), std::fabs((double) (im - Z[k].im))));
        }
        printf("N=%d forward: worst |delta| %.3g\n", D::N, worst);
        if (!(worst <= 1e-10)) bad++;
    }
    // products against the exact schoolbook product
    int worst_lsb = 0;
    for (int trial = 0; trial < 6; trial++) {
        std::vector<int32_t> a(D::N), b(D::N), want(D::N), got(D::N);
        for (int i = 0; i < D::N; i++) {
            a[i] = (int32_t) (rng() % (4 * D::N)) - 2 * D::N;  // src/test/test-multiplication.cpp:41-44
            b[i] = (int32_t) rng();
        }
        if (trial == 1) for (int i = 1; i < D::N; i++) a[i] = 0;
        oracle_mulNaive(want.data(), a.data(), b.data(), D::N);
        std::vector<double> pa(D::N), pb(D::N);
        for (int i = 0; i < D::N; i++) pa[i] = a[i], pb[i] = b[i] * (1.0 / 4294967296.0);
        std::vector<c64> A, B;
        emu_forward<LH>(pa.data(), A, tables.data());
        emu_forward<LH>(pb.data(), B, tables.data());
        for (int k = 0; k < D::NH; k++) A[k] = brf::cmul(A[k], B[k]);
        emu_inverse<LH>(A, got.data(), tables.data());
        for (int i = 0; i < D::N; i++) {
            const int d = std::abs((int32_t) ((uint32_t) got[i] - (uint32_t) want[i]));
            if (d > worst_lsb) worst_lsb = d;
        }
    }
    printf("N=%d products: worst difference to the exact product %d LSB\n", D::N, worst_lsb);
    if (worst_lsb > (LH == 10 ? 2 : 1)) bad++;
    return bad;
}

int main() {
    const int bad = run<8>() + run<9>() + run<10>();
    printf(bad ? "EMU NFFT FAILED\n" : "EMU NFFT OK\n");
    return bad != 0;
}

"""TGSW / TLWE operations with N, k, l, Bgbit as run-time parameters (include/tfhe_hip.h "run-time
parameters"; tfhe_amd/csrc/nk_ops.h, nfft_core.h).

Oracle: oracle_*_n of oracle/tfhe_oracle.c - the reference's generic decomposition and its EXACT schoolbook products
composed as its generic tGswExternMulToTLwe / tLweAddMulRTo (tgsw-functions.cpp:157-170, tlwe-functions.cpp:191-197).
Pinned by tests/golden/nk.npz (outputs of the reference itself: Torus32PolynomialDecompH_old at N = 512 / 1024,
torusPolynomialMultNaive_aux at N = 512 / 1024 / 2048; oracle/gen_golden_nk.py).  The reference cannot run its COMPOSED
external product at these parameters (hard-wired decomposition and N = 1024 transform in this fork), so for it parity is
the identity its own test checks (src/test/tgsw_fft_test.cpp:245-285): FFT result == sum of exact row products, here
within the transform's rounding, with the bound written at each assertion.  Parameter sets: src/test/tgsw_fft_test.cpp:28-35."""
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "nk.npz"))
PARAM_SETS = [(512, 1, 4, 8), (512, 2, 3, 10), (1024, 1, 3, 10), (1024, 2, 4, 8), (2048, 1, 4, 8), (2048, 2, 3, 10)]


def wrap(d):
    return (np.asarray(d, np.int64) + 2**31) % 2**32 - 2**31


def decomp_keys():
    return sorted(k[:-3] for k in GOLD.files if k.startswith("decomp_") and k.endswith("_in"))


def test_oracle_replays_the_reference_at_other_parameters(oracle):
    for key in decomp_keys():
        _, n, l, bgbit = key.split("_")
        got = oracle.decomp_n(GOLD[key + "_in"], int(n), int(l), int(bgbit))
        assert np.array_equal(got, GOLD[key + "_out"]), key
    for n in (512, 1024, 2048):
        for i in range(3):
            assert np.array_equal(oracle.mul_naive(GOLD["mul_%d_a" % n][i], GOLD["mul_%d_b" % n][i]), GOLD["mul_%d_out" % n][i])


def test_oracle_external_product_reconstructs(oracle):
    """the generic external product of the oracle against plain integer arithmetic in numpy (N = 512, k = 2): rows of a
    TGSW sample that is the gadget matrix times mu reproduce mu * acc up to the decomposition error (test-decomp-tgsw.cpp:72-79)"""
    n, k, l, bgbit = 512, 2, 3, 10
    rng = np.random.default_rng(1)
    acc = rng.integers(-2**31, 2**31, (k + 1, n)).astype(np.int32)
    tgsw = np.zeros(((k + 1) * l, k + 1, n), np.int32)
    for i in range(k + 1):
        for p in range(l):
            tgsw[i * l + p, i, 0] = 3 << (32 - (p + 1) * bgbit)          # mu = 3 on the gadget diagonal, no noise
    out = oracle.tgsw_extern_mul_n(acc, tgsw, n, k, l, bgbit)
    assert np.abs(wrap(out.astype(np.int64) - 3 * acc.astype(np.int64))).max() <= 3 * (1 << (32 - l * bgbit)) // 2 + 3


@pytest.mark.gpu
def test_decomposition_replays_the_reference(engine):
    for key in decomp_keys():
        _, n, l, bgbit = key.split("_")
        got = engine.decomp_n(int(n), int(l), int(bgbit), GOLD[key + "_in"][None])[0]
        assert np.array_equal(got, GOLD[key + "_out"]), key
    from tfhe_amd import TfheHipError
    with pytest.raises(TfheHipError):
        engine.decomp_n(512, 4, 9, np.zeros((1, 512), np.int32))            # l * Bgbit > 32


@pytest.mark.gpu
@pytest.mark.parametrize("n", [512, 1024, 2048])
def test_products_within_the_references_tolerance(n, engine, fast, oracle):
    """src/test/test-multiplication.cpp:66-77: FFT product == exact product within 1 LSB for |a_i| < 2N, b uniform
    (2 LSB at N = 2048, where the sums reach 2^53).  The golden rows are the reference's own exact products."""
    a, b, want = GOLD["mul_%d_a" % n], GOLD["mul_%d_b" % n], GOLD["mul_%d_out" % n]
    tol = 2 if n == 2048 else 1
    for eng in (engine, fast):                                               # the same transform in both arithmetic modes
        got = eng.negacyclic_mul_n(n, a, b)
        assert np.abs(wrap(got.astype(np.int64) - want)).max() <= tol
    rng = np.random.default_rng(n)
    cnt = 4 * 256 * 4 + 37                                                   # more than one pass of the persistent workgroups
    a2 = rng.integers(-2 * n, 2 * n, (cnt, n)).astype(np.int32)
    b2 = rng.integers(-2**31, 2**31, (cnt, n)).astype(np.int32)
    got2 = engine.negacyclic_mul_n(n, a2, b2)
    for i in (0, 1, 4095, 4096, cnt - 1):
        assert np.abs(wrap(got2[i].astype(np.int64) - oracle.mul_naive(a2[i], b2[i]))).max() <= tol, i
    assert np.array_equal(fast.negacyclic_mul_n(n, a2[:5], b2[:5]), got2[:5])


@pytest.mark.gpu
@pytest.mark.parametrize("n", [512, 1024, 2048])
def test_transforms_round_trip_and_reference_layout(n, engine, oracle):
    """spectra come in the reference's coefsC order (P(w^(2i+1)), i < N/2): checked against the direct evaluation; at
    N = 1024 also against the oracle's TorusPolynomial_ifft (1e-10, test-MulToTLwe.cpp:365-371); inverse(forward(p)) == p"""
    rng = np.random.default_rng(7 * n)
    p = rng.integers(-2**31, 2**31, (3, n)).astype(np.int32)
    p[0] = 0
    spec = engine.ifft_n(n, p, torus=True)
    ints = rng.integers(-64, 64, (2, n)).astype(np.int32)
    ispec = engine.ifft_n(n, ints, torus=False)
    j = np.arange(n)
    for i in (0, 1, n // 4, n // 2 - 1):
        w = np.exp(1j * np.pi * (2 * i + 1) * j / n)
        assert abs((ints[1] * w).sum() - complex(*ispec[1, i])) <= 1e-9
        assert abs(((p[1] / 2.0**32) * w).sum() - complex(*spec[1, i])) <= 1e-9
    if n == 1024:
        assert np.abs(spec[2] - oracle.torus_ifft(p[2])).max() <= 1e-10
    back = engine.fft_n(n, spec)
    assert np.abs(wrap(back.astype(np.int64) - p)).max() <= 1 and np.array_equal(back[0], p[0])


@pytest.mark.gpu
@pytest.mark.parametrize("n,k,l,bgbit", PARAM_SETS)
def test_external_products_equal_the_exact_sum_of_row_products(n, k, l, bgbit, engine, fast, oracle):
    """src/test/tgsw_fft_test.cpp:245-285 with real data: tGswFFTExternMulToTLwe(acc, gsw) == sum_i dec_i * row_i, the
    right-hand side exact (oracle: generic decomposition + schoolbook products).  Bound: the evaluation-domain sum of
    (k+1) l N products of magnitude Bg/2 * 2^31 carries up to 2^(log2(sqrt((k+1) l N)) + Bgbit + 30 - 52) LSB of FP64
    rounding: 1-2 LSB at these sets; asserted <= 4.  The coefficient-domain form rounds each of the (k+1) l row products
    on its own: <= (k+1) l LSB."""
    rng = np.random.default_rng(n + k)
    ngsw, count = 2, 5
    tgsw = rng.integers(-2**31, 2**31, (ngsw, (k + 1) * l, k + 1, n)).astype(np.int32)
    acc = rng.integers(-2**31, 2**31, (count, k + 1, n)).astype(np.int32)
    acc[0] = 0
    idx = np.array([0, 1, 1, 0, 1], np.int32)
    want = np.stack([oracle.tgsw_extern_mul_n(acc[g], tgsw[idx[g]], n, k, l, bgbit) for g in range(count)])
    gsw_fft = engine.ifft_n(n, tgsw.reshape(-1, n), torus=True).reshape(ngsw, (k + 1) * l, k + 1, n // 2, 2)   # tGswToFFTConvert
    for eng in (engine, fast):
        got_f = eng.tgsw_extern_mul_n(n, k, l, bgbit, acc, gsw_fft, idx, fft=True)
        got_c = eng.tgsw_extern_mul_n(n, k, l, bgbit, acc, tgsw, idx, fft=False)
        d_f = int(np.abs(wrap(got_f.astype(np.int64) - want)).max())
        d_c = int(np.abs(wrap(got_c.astype(np.int64) - want)).max())
        print("N=%d k=%d l=%d Bgbit=%d: FFT-domain sum %d LSB, row by row %d LSB from the exact result" % (n, k, l, bgbit, d_f, d_c))
        assert d_f <= 4 and d_c <= (k + 1) * l
    # a zero accumulator: digits are ((offset >> ..) & mask) - Bg/2 = 0 for every level (tgsw.cpp:21-27), so the product is 0
    assert not got_f[0].any() and not got_c[0].any()


@pytest.mark.gpu
@pytest.mark.parametrize("n,k", [(512, 2), (1024, 1), (2048, 2)])
def test_tlwe_addmul_r(n, k, engine, oracle):
    rng = np.random.default_rng(3 * n + k)
    count = 4
    res = rng.integers(-2**31, 2**31, (count, k + 1, n)).astype(np.int32)
    poly = rng.integers(-512, 512, (count, n)).astype(np.int32)
    poly[0] = 0
    sample = rng.integers(-2**31, 2**31, (count, k + 1, n)).astype(np.int32)
    got = engine.tlwe_addmul_r_n(n, k, res, poly, sample)
    want = np.stack([oracle.tlwe_addmul_r_n(res[g], poly[g], sample[g], n, k) for g in range(count)])
    assert np.abs(wrap(got.astype(np.int64) - want)).max() <= 1
    assert np.array_equal(got[0], res[0])


@pytest.mark.gpu
def test_argument_errors(engine):
    from tfhe_amd import TfheHipError
    z = np.zeros((1, 768), np.int32)
    with pytest.raises(TfheHipError, match="512, 1024 or 2048"):
        engine.negacyclic_mul_n(768, z, z)
    with pytest.raises(TfheHipError, match="k must be"):
        engine.tgsw_extern_mul_n(512, 3, 2, 8, np.zeros((1, 4, 512), np.int32), np.zeros((1, 8, 4, 512), np.int32), [0])
    with pytest.raises(TfheHipError, match="out of range"):
        engine.tgsw_extern_mul_n(512, 1, 2, 8, np.zeros((1, 2, 512), np.int32), np.zeros((1, 4, 2, 512), np.int32), [1])
    assert engine.negacyclic_mul_n(512, np.zeros((0, 512), np.int32), np.zeros((0, 512), np.int32)).shape == (0, 512)

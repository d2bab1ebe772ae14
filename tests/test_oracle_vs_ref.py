"""Authoring-container only: the oracle against the reference ITSELF (oracle/_ref/libtfhe_ref.so, compiled from
/root/reference by `make -C oracle ref`), bit for bit, on fresh random inputs and on reference-generated keys."""
import ctypes

import numpy as np
import pytest

from tests.oracle_lib import REF_SO, P, have_ref
from tfhe_amd import PARAMS as PR

pytestmark = [pytest.mark.ref, pytest.mark.skipif(not have_ref(), reason="reference build oracle/_ref absent")]


@pytest.fixture(scope="module")
def ref():
    lib = ctypes.CDLL(REF_SO)
    lib.ref_phase.restype = ctypes.c_int32
    lib.ref_modswitch_from.restype = ctypes.c_int32
    return lib


def test_tables_equal_reference_header(oracle, ref):
    L = oracle.lib.oracle_trig_table_len()
    assert L == ref.ref_trig_table_len()
    a, b, c, d = (np.zeros(L) for _ in range(4))
    oracle.lib.oracle_get_trig_tables(P(a), P(b))
    ref.ref_get_trig_tables(P(c), P(d))
    assert np.array_equal(a, c) and np.array_equal(b, d)       # src/include/tfhe_core.h:26,93
    assert np.array_equal(np.signbit(a), np.signbit(c)) and np.array_equal(np.signbit(b), np.signbit(d))
    x, y = np.zeros(2048, np.int32), np.zeros(2048, np.int32)
    oracle.lib.oracle_get_bit_reversed(P(x))
    ref.ref_get_bit_reversed(P(y))
    assert np.array_equal(x, y)                                  # :160


def test_primitives_random(oracle, ref):
    rng = np.random.default_rng(5)
    N = PR.N
    for _ in range(25):
        p = rng.integers(-64, 64, N).astype(np.int32)
        t = rng.integers(-2**31, 2**31, N).astype(np.int32)
        a, b = np.zeros((N // 2, 2)), np.zeros(N, np.int32)
        ref.ref_int_ifft(P(a), P(p))
        assert np.array_equal(oracle.int_ifft(p), a)
        ref.ref_torus_ifft(P(a), P(t))
        assert np.array_equal(oracle.torus_ifft(t), a)
        spec = np.ascontiguousarray(a * rng.integers(1, 10**5))
        ref.ref_torus_fft(P(b), P(spec))
        assert np.array_equal(oracle.torus_fft(spec), b)
        m = rng.integers(-2 * N, 2 * N, N).astype(np.int32)
        ref.ref_mul_fft(P(b), P(m), P(t))
        assert np.array_equal(oracle.mul_fft(m, t), b)
        d = np.zeros((3, N), np.int32)
        ref.ref_decomp(P(d), P(t))
        assert np.array_equal(oracle.decomp(t), d)
        ai = int(rng.integers(1, 2 * N))
        ref.ref_mul_by_xai(P(b), ai, P(t))
        assert np.array_equal(oracle.mul_by_xai(ai, t), b)
        ref.ref_mul_by_xai_minus_one(P(b), ai, P(t))
        assert np.array_equal(oracle.mul_by_xai_minus_one(ai, t), b)
    for ph in rng.integers(-2**31, 2**31, 2000):
        assert oracle.modswitch_from(int(ph), 2 * N) == ref.ref_modswitch_from(int(ph), 2 * N)


def test_gates_with_reference_generated_keys(oracle, ref):
    """keys, encryptions and gates all from the reference's own generator; the oracle gets the flat key arrays"""
    n, N = PR.n, PR.N
    ref.ref_keygen(ctypes.c_uint32(2024))
    lk, tk = np.zeros(n, np.int32), np.zeros(N, np.int32)
    bk, ks = np.zeros((n, 6, 2, N), np.int32), np.zeros((N, 8, 4, n + 1), np.int32)
    ref.ref_get_keys(P(lk), P(tk), P(bk), P(ks))
    ok = oracle.keys_from_coef(bk, ks)
    f = np.zeros((n, 6, 2, N // 2, 2))
    ref.ref_get_bkfft(P(f))
    assert np.array_equal(oracle.bkfft(ok), f)
    cts = np.zeros((3, n + 1), np.int32)
    for i, bit in enumerate((1, 0, 1)):
        ref.ref_encrypt(P(cts[i]), bit)
    for gate in (0, 3, 10):
        want = np.zeros(n + 1, np.int32)
        ref.ref_gate(gate, P(want), P(cts[0]), P(cts[1]), P(cts[2]))
        assert np.array_equal(oracle.gate(ok, gate, cts[0], cts[1], cts[2]), want), gate
    oracle.lib.oracle_keys_free(ok)


def test_client_noise_follows_the_references_gaussian32(ref, client):
    """the reference's own gaussian32 (numeric-functions.cpp:22-27), 300,000 draws at each of the two standard deviations,
    against the client's integer sampler: two-sample Kolmogorov-Smirnov, equal standard deviations"""
    from scipy.stats import ks_2samp
    ref.gaussian32.restype = ctypes.c_int32
    ref.gaussian32.argtypes = [ctypes.c_int32, ctypes.c_double]
    seed = (ctypes.c_uint32 * 1)(99)
    ref.tfhe_random_generator_setSeed(seed, 1)
    for m, sigma in ((1 << 25, 2.0 ** -15), (1 << 15, 2.0 ** -25)):
        r = np.array([ref.gaussian32(0, sigma) for _ in range(300_000)], np.float64)
        c = client.noise(12, m, 300_000).astype(np.float64)
        assert abs(r.std() / c.std() - 1) < 0.01
        assert ks_2samp(r, c).pvalue > 1e-3
        assert np.abs(r).max() < 6.5 * sigma * 2 ** 32 and np.abs(c).max() < 6.5 * sigma * 2 ** 32

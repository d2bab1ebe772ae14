"""Leveled TGSW / TLWE operations on caller-supplied operands: tGswExternMulToTLwe
(tgsw-functions.cpp:157-170), tGswFFTExternMulToTLwe with arbitrary TGswSampleFFT operands (tgsw-fft-operations.cpp:92-120)
and tLweAddMulRTo (tlwe-functions.cpp:191-197).  tests/golden/leveled.npz holds the reference's own outputs
(oracle/gen_golden_leveled.py); the oracle (CPU suite) and the GPU library (-m gpu, EXACT arithmetic) replay them bit for
bit; the FAST arithmetic stays within 1 LSB per product."""
import os

import numpy as np
import pytest

from tfhe_amd import PARAMS as P


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "leveled.npz"))


def wrap(d):
    return (np.asarray(d, np.int64) + 2**31) % 2**32 - 2**31


def test_oracle_replays_the_reference(gold, oracle):
    g = gold
    for i, ix in enumerate(g["index"]):
        assert np.array_equal(oracle.tgsw_to_fft(g["tgsw"][ix]), g["gsw_fft"][ix])
        assert np.array_equal(oracle.tgsw_extern_mul(g["acc"][i], g["tgsw"][ix]), g["extern_mul_out"][i]), i
        assert np.array_equal(oracle.extern_mul_fft(g["acc"][i], g["gsw_fft"][ix]), g["fft_extern_mul_out"][i]), i
        assert np.array_equal(oracle.tlwe_addmul_r(g["addmul_in"][i], g["addmul_poly"][i], g["addmul_sample"][i]),
                              g["addmul_out"][i]), i


@pytest.mark.ref
def test_oracle_equals_the_reference_on_random_operands(oracle):
    import ctypes
    from tests.oracle_lib import REF_SO, P as ptr, have_ref
    if not have_ref():
        pytest.skip("reference build absent")
    R = ctypes.CDLL(REF_SO)
    rng = np.random.default_rng(7)
    for _ in range(3):
        tgsw = rng.integers(-2**31, 2**31, (P.kpl, 2, P.N)).astype(np.int32)
        acc = rng.integers(-2**31, 2**31, (2, P.N)).astype(np.int32)
        want = acc.copy()
        R.ref_tgsw_extern_mul(ptr(want), ptr(tgsw))
        assert np.array_equal(oracle.tgsw_extern_mul(acc, tgsw), want)


@pytest.mark.gpu
def test_gpu_replays_the_reference(gold, engine):
    g = gold
    assert np.array_equal(engine.tgsw_extern_mul(g["acc"], g["tgsw"], g["index"]), g["extern_mul_out"])
    assert np.array_equal(engine.tgsw_fft_extern_mul(g["acc"], g["gsw_fft"], g["index"]), g["fft_extern_mul_out"])
    assert np.array_equal(engine.tlwe_addmul_r(g["addmul_in"], g["addmul_poly"], g["addmul_sample"]), g["addmul_out"])
    assert np.array_equal(engine.torus_ifft(g["tgsw"].reshape(-1, P.N)).reshape(g["gsw_fft"].shape), g["gsw_fft"])   # tGswToFFTConvert


@pytest.mark.gpu
def test_gpu_batch_equals_the_oracle_and_fast_mode_is_close(engine, oracle, keys):
    from tfhe_amd import Engine, TfheHipError
    rng = np.random.default_rng(11)
    ngsw, count = 5, 300                       # 300: above the one-wave launch shape of the external product
    tgsw = rng.integers(-2**31, 2**31, (ngsw, P.kpl, 2, P.N)).astype(np.int32)
    acc = rng.integers(-2**31, 2**31, (count, 2, P.N)).astype(np.int32)
    idx = rng.integers(0, ngsw, count).astype(np.int32)
    gfft = np.stack([oracle.tgsw_to_fft(t) for t in tgsw])
    out_c = engine.tgsw_extern_mul(acc, tgsw, idx)
    out_f = engine.tgsw_fft_extern_mul(acc, gfft, idx)
    for i in (0, 1, 150, 299):
        assert np.array_equal(out_c[i], oracle.tgsw_extern_mul(acc[i], tgsw[idx[i]])), i
        assert np.array_equal(out_f[i], oracle.extern_mul_fft(acc[i], gfft[idx[i]])), i
    # the two forms of the external product agree up to the roundings of the row-by-row form (one truncation per row)
    assert np.abs(wrap(out_c.astype(np.int64) - out_f)).max() <= P.kpl
    with pytest.raises(TfheHipError):
        engine.tgsw_extern_mul(acc[:1], tgsw, np.array([ngsw], np.int32))
    # operand arrays are addressed with 32-bit byte offsets: more than 2 GiB of samples in one call is refused before
    # anything is read (the arrays passed here are far smaller than the count claims)
    import ctypes
    one = acc[:1].copy()
    rc = engine.lib.tfhe_hip_tgsw_fft_extern_mul_batch(engine.ctx, one.ctypes.data_as(ctypes.c_void_p),
                                                       gfft.ctypes.data_as(ctypes.c_void_p), idx[:1].ctypes.data_as(ctypes.c_void_p),
                                                       ctypes.c_int64(30000), ctypes.c_int64(1))
    assert rc != 0 and b"2 GiB" in engine.lib.tfhe_hip_last_error()
    e = Engine(0)
    try:
        e.set_arithmetic("fast")               # no keys needed for these stages
        assert np.abs(wrap(e.tgsw_fft_extern_mul(acc, gfft, idx).astype(np.int64) - out_f)).max() <= 4
        assert np.abs(wrap(e.tgsw_extern_mul(acc, tgsw, idx).astype(np.int64) - out_c)).max() <= P.kpl
        poly = rng.integers(-64, 64, (count, P.N)).astype(np.int32)
        res = rng.integers(-2**31, 2**31, (count, 2, P.N)).astype(np.int32)
        exact = engine.tlwe_addmul_r(res, poly, acc)
        assert np.array_equal(exact[7], oracle.tlwe_addmul_r(res[7], poly[7], acc[7]))
        assert np.abs(wrap(e.tlwe_addmul_r(res, poly, acc).astype(np.int64) - exact)).max() <= 1
    finally:
        e.close()

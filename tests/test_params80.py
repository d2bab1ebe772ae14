"""The 80-bit parameter set (n = 500, l = 2, Bgbit = 10; tfhe_gate_bootstrapping.cpp:23-48), SURVEY 8(f) N4.

PINNED PER STAGE, composed bootstrap unpinned.  This fork hard-codes the default set's n, l, Bgbit as compile-time
constants in its composed hot path (src/include/tfhe_core.h:13-24, used in tGswTorus32PolynomialDecompH and
lwe-bootstrapping-functions-fft.cpp), so the reference cannot run a whole bootstrap of this set.  Its stages, however,
take their parameters at run time: lweKeySwitch, the generic gadget decomposition Torus32PolynomialDecompH_old,
tGswToFFTConvert and the transform-domain pieces of tGswFFTExternMulToTLwe.  tests/golden/stages_80.npz holds inputs
and the reference's own outputs of those (oracle/gen_golden_80.py, ref80_* of oracle/ref_harness.cpp); the oracle built
for this set (oracle/liboracle_80.so) and the GPU library (libtfhe_hip_80.so) must replay them bit for bit.  On top:
  * the oracle against an independent numpy statement of the gadget decomposition, and through decryption of its gates;
  * the GPU library against that oracle, bit for bit, in every launch shape (the composed bootstrap: same CMux code as
    the pinned stages, iterated)."""
import os

import numpy as np
import pytest

from tfhe_amd import Gate, PARAMS_80 as P
from tests.truth import TRUTH


@pytest.fixture(scope="module")
def client80():
    from tfhe_amd import Client
    return Client(P)


@pytest.fixture(scope="module")
def keys80(client80):
    return client80.keygen(42)


@pytest.fixture(scope="module")
def oracle80():
    from tests.oracle_lib import Oracle
    return Oracle(P)


@pytest.fixture(scope="module")
def okeys80(oracle80, keys80):
    return oracle80.keys_from_coef(keys80.bk_coef, keys80.ks)


@pytest.fixture(scope="module")
def engine80(keys80):
    from tfhe_amd import Engine
    e = Engine(0, P)
    e.load_keys_coef(keys80.bk_coef, keys80.ks)
    yield e
    e.close()


def test_shapes_and_noise_of_the_client_keys(client80, keys80):
    assert keys80.lwe_key.shape == (500,) and keys80.bk_coef.shape == (500, 4, 2, 1024)
    assert keys80.ks.shape == (1024, 8, 4, 501)
    bits = np.random.default_rng(0).integers(0, 2, 4000).astype(np.uint8)
    cts = client80.encrypt(keys80, bits, seed=3)
    got, ph = client80.decrypt(keys80, cts)
    assert np.array_equal(got, bits)
    noise = ph.astype(np.int64) - np.where(bits == 1, 1 << 29, -(1 << 29))
    want = 2.44e-5 * 2.0 ** 32                           # ks_stdev of the set (tfhe_gate_bootstrapping.cpp:35)
    assert 0.93 * want < noise.std() < 1.07 * want


def test_gadget_decomposition_of_the_oracle(oracle80):
    """tGswTorus32PolynomialDecompH (tgsw-functions.cpp:297-331) for Bgbit = 10, l = 2, stated independently:
    digits in [-512, 512) with sum_p d_p 2^(32-10(p+1)) = the coefficient truncated to 20 bits (the offset holds no
    rounding term)"""
    rng = np.random.default_rng(1)
    poly = rng.integers(-2 ** 31, 2 ** 31, 1024, dtype=np.int64).astype(np.int32)
    poly[:4] = [0, -1, 2 ** 31 - 1, -2 ** 31]
    dec = oracle80.decomp(poly)
    assert dec.shape == (2, 1024) and dec.min() >= -512 and dec.max() < 512
    recomposed = (dec[0].astype(np.int64) << 22) + (dec[1].astype(np.int64) << 12)
    err = (recomposed - poly.astype(np.int64) + 2 ** 31) % 2 ** 32 - 2 ** 31
    assert err.max() <= 0 and err.min() > -2 ** 12       # the 12 dropped bits
    offset = (512 << 22) + (512 << 12)
    u = (poly.astype(np.int64) + offset) % 2 ** 32
    assert np.array_equal(dec[0], ((u >> 22) & 1023) - 512) and np.array_equal(dec[1], ((u >> 12) & 1023) - 512)


def test_oracle_gates_decrypt(client80, keys80, oracle80, okeys80):
    rng = np.random.default_rng(2)
    bits = rng.integers(0, 2, (3, 6)).astype(np.uint8)
    a, b, c = (client80.encrypt(keys80, bits[i], seed=10 + i) for i in range(3))
    for gate in (Gate.NAND, Gate.XOR, Gate.MUX):
        out = oracle80.gate_batch(okeys80, gate, a, b, c if gate == Gate.MUX else None, nthreads=6)
        got, ph = client80.decrypt(keys80, out)
        assert np.array_equal(got, TRUTH[int(gate)](bits[0], bits[1], bits[2])), gate
        assert np.abs(np.abs(ph.astype(np.int64)) - (1 << 29)).max() < (1 << 27)


def test_library_reports_its_parameter_set():
    from tfhe_amd.engine import load_library
    import ctypes
    got = (ctypes.c_int32 * 8)()
    load_library(P).tfhe_hip_param_set(got)
    assert list(got) == [80, 500, 1024, 1, 2, 10, 8, 2]


@pytest.mark.gpu
def test_gates_equal_the_oracle_in_every_launch_shape(engine80, client80, keys80, oracle80, okeys80, monkeypatch):
    from tfhe_amd import Engine
    rng = np.random.default_rng(4)
    m = 24
    bits = rng.integers(0, 2, (3, m)).astype(np.uint8)
    a, b, c = (client80.encrypt(keys80, bits[i], seed=20 + i) for i in range(3))
    want = {g: oracle80.gate_batch(okeys80, g, a, b, c if g == Gate.MUX else None, nthreads=8)
            for g in (Gate.NAND, Gate.XNOR, Gate.ORYN, Gate.MUX)}
    for g, w in want.items():
        assert np.array_equal(engine80.gate_batch(g, a, b, c if g == Gate.MUX else None), w), g
        assert np.array_equal(client80.decrypt(keys80, w)[0], TRUTH[int(g)](bits[0], bits[1], bits[2]))
    for shape in ("1", "4", "8"):                        # the default above was the one-workgroup-per-gate shape
        monkeypatch.setenv("TFHE_HIP_BR_WAVES", shape)
        e = Engine(0, P)
        try:
            e.load_keys_coef(keys80.bk_coef, keys80.ks)
            assert np.array_equal(e.gate_batch(Gate.NAND, a, b), want[Gate.NAND]), shape
            assert np.array_equal(e.gate_batch(Gate.MUX, a, b, c), want[Gate.MUX]), shape
        finally:
            e.close()
    monkeypatch.delenv("TFHE_HIP_BR_WAVES")


@pytest.mark.gpu
def test_stages_equal_the_oracle(engine80, client80, keys80, oracle80, okeys80):
    rng = np.random.default_rng(6)
    x = client80.encrypt(keys80, rng.integers(0, 2, 5).astype(np.uint8), seed=40)
    ext = engine80.bootstrap_batch(x, with_ks=False)
    for i in range(5):
        assert np.array_equal(ext[i], oracle80.bootstrap_woks(okeys80, x[i]))
    ks = engine80.keyswitch_batch(ext)
    for i in range(5):
        assert np.array_equal(ks[i], oracle80.keyswitch(okeys80, ext[i]))
    acc = rng.integers(-2 ** 31, 2 ** 31, (3, 2, 1024), dtype=np.int64).astype(np.int32)
    idx = np.array([0, 17, 499], np.int32)
    got = engine80.extern_mul_batch(acc.copy(), idx)
    for i in range(3):
        assert np.array_equal(got[i], oracle80.extern_mul(okeys80, acc[i], int(idx[i])))


@pytest.mark.gpu
def test_throughput_shape_on_a_full_batch(engine80, client80, keys80, oracle80, okeys80):
    """2,304 gates (more than one wave per SIMD on every CU) through the 8-wave kernel and the MFMA keyswitch with its
    ragged last tile; a sample of the outputs against the oracle, all of them through decryption"""
    rng = np.random.default_rng(8)
    m = 2304 + 5
    bits = rng.integers(0, 2, (2, m)).astype(np.uint8)
    a, b = client80.encrypt(keys80, bits[0], seed=50), client80.encrypt(keys80, bits[1], seed=51)
    out = engine80.gate_batch(Gate.NAND, a, b)
    assert np.array_equal(client80.decrypt(keys80, out)[0], 1 - (bits[0] & bits[1]))
    pick = np.array([0, 1, 63, 64, 1151, 2303, 2304, m - 1])
    want = oracle80.gate_batch(okeys80, Gate.NAND, a[pick], b[pick], None, nthreads=8)
    assert np.array_equal(out[pick], want)


@pytest.mark.gpu
def test_fast_arithmetic_of_the_80_bit_library(engine80, client80, keys80, oracle80, okeys80):
    """tfhe_hip_set_arithmetic(FAST) in the second build (l = 2: four forward waves in the latency kernel): one CMux
    within 4 LSB of the oracle's in both kernels, gates decrypt to the truth table with the exact mode's noise level,
    keyswitch still exact"""
    from tfhe_amd import Engine
    e = Engine(0, P)
    try:
        e.set_arithmetic("fast")
        e.load_keys_coef(keys80.bk_coef, keys80.ks)
        rng = np.random.default_rng(12)
        wrap = lambda d: (np.asarray(d, np.int64) + 2**31) % 2**32 - 2**31  # noqa: E731
        for cnt in (5, 520):                              # latency kernel | throughput kernel
            acc = rng.integers(-2 ** 31, 2 ** 31, (cnt, 2, 1024), dtype=np.int64).astype(np.int32)
            bara = np.zeros((cnt, P.n), np.int32)
            bara[np.arange(cnt), rng.integers(0, P.n, cnt)] = rng.integers(1, 2048, cnt)
            out = e.blind_rotate_batch(acc, bara)
            for g in (0, 1, cnt - 1):
                assert np.abs(wrap(out[g].astype(np.int64) - oracle80.blind_rotate(okeys80, acc[g], bara[g]))).max() <= 4
        for m in (24, 1100):
            bits = rng.integers(0, 2, (3, m)).astype(np.uint8)
            a, b, c = (client80.encrypt(keys80, bits[i], seed=60 + i) for i in range(3))
            for g in (Gate.NAND, Gate.MUX):
                got = e.gate_batch(g, a, b, c if g == Gate.MUX else None)
                dec, ph = client80.decrypt(keys80, got)
                assert np.array_equal(dec, TRUTH[int(g)](bits[0], bits[1], bits[2])), (g, m)
                assert np.abs(np.abs(ph.astype(np.int64)) - (1 << 29)).max() < (1 << 28)
            if m == 1100:
                err = np.abs(ph.astype(np.int64)) - (1 << 29)
                ref = np.abs(client80.decrypt(keys80, engine80.gate_batch(Gate.MUX, a, b, c))[1].astype(np.int64)) - (1 << 29)
                assert 0.85 < err.std() / ref.std() < 1.15
        u = rng.integers(-2 ** 31, 2 ** 31, (70, 1025), dtype=np.int64).astype(np.int32)
        assert np.array_equal(e.keyswitch_batch(u), engine80.keyswitch_batch(u))
    finally:
        e.close()


@pytest.mark.gpu
def test_client_side_kernels(engine80, client80, keys80):
    from tfhe_amd import Engine
    bits = np.random.default_rng(9).integers(0, 2, 777).astype(np.uint8)
    want = client80.encrypt(keys80, bits, seed=5)
    assert np.array_equal(engine80.encrypt_batch(5, keys80.lwe_key, bits), want)
    assert np.array_equal(engine80.phase_batch(keys80.lwe_key, want), client80.decrypt(keys80, want)[1])
    e2 = Engine(0, P)
    try:
        coef = e2.keygen_bk(42, keys80.lwe_key, keys80.tlwe_key, keys80.ks, want_coef=True)
        assert np.array_equal(coef, keys80.bk_coef)
        assert np.array_equal(e2.get_bkfft().view(np.int64), engine80.get_bkfft().view(np.int64))
        bk, ks = e2.keygen_cloud(42, keys80.lwe_key, keys80.tlwe_key, want_keys=True)   # keyswitch key as well
        assert np.array_equal(ks, keys80.ks) and np.array_equal(bk, keys80.bk_coef)
        a, b = want[:50], want[50:100]
        assert np.array_equal(e2.gate_batch(Gate.NOR, a, b), engine80.gate_batch(Gate.NOR, a, b))
    finally:
        e2.close()


def test_wire_formats_of_the_80_bit_library(keys80, client80, tmp_path):
    """tfhe_io.cpp's formats with this set's numbers: the parameter text names n = 500, l = 2, Bgbit = 10 and the
    set's standard deviations; key file and ciphertext stream round-trip; the default library refuses the file"""
    from tfhe_amd import wire, TfheHipError, PARAMS
    text = wire.write_params(P).decode()
    fields = dict(ln.split(":") for ln in text.splitlines() if ":" in ln)
    fields = {k.strip(): v.strip() for k, v in fields.items()}
    assert (fields["n"], fields["l"], fields["Bgbit"], fields["N"], fields["ks_t"]) == ("500", "2", "10", "1024", "8")
    assert fields["alpha_min"] == "0.00000001"     # the TLWE section's 7.18e-9 as "%.8lf" (tfhe_generic_streams.cpp:46-50)
    assert "alpha_min: 0.00002440" in text         # the LWE section's 2.44e-5
    assert wire.read_params(text.encode(), P) == len(text)
    with pytest.raises(TfheHipError):
        wire.read_params(text.encode(), PARAMS)
    path = tmp_path / "cloud80.key"
    wire.write_cloud_keyset_file(path, keys80.bk_coef, keys80.ks, params=P)
    bk, ks, var = wire.read_cloud_keyset_file(path, P)
    assert np.array_equal(bk, keys80.bk_coef) and np.array_equal(ks, keys80.ks)
    with pytest.raises(TfheHipError):
        wire.read_cloud_keyset_file(path, PARAMS)
    cts = client80.encrypt(keys80, np.array([0, 1, 1], np.uint8), seed=8)
    blob = wire.write_samples(cts, params=P)
    assert len(blob) == 3 * wire.sample_bytes(P) == 3 * (4 + 4 * 501 + 8)
    assert np.array_equal(wire.read_samples(blob, params=P)[0], cts)


# ---- pinned against the reference, stage by stage (tests/golden/stages_80.npz) -----------------------------------------
@pytest.fixture(scope="module")
def stages80():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "stages_80.npz"))


def test_oracle_replays_the_reference_stages(stages80, oracle80, okeys80):
    g = stages80
    for i in range(g["decomp_in"].shape[0]):     # Torus32PolynomialDecompH_old, l = 2, Bgbit = 10
        dec = oracle80.decomp(g["decomp_in"][i])
        assert np.array_equal(dec, g["decomp_out"][i])
        # the reconstruction check of src/test/test-decomp-tgsw.cpp:72-79 (written there for Bgbit = 10)
        rec = (dec[0].astype(np.int64) << 22) + (dec[1].astype(np.int64) << 12)
        err = (rec - g["decomp_in"][i].astype(np.int64) + 2**31) % 2**32 - 2**31
        assert np.abs(err).max() < 2**12
    for i in range(0, g["ks_in"].shape[0], 7):   # lweKeySwitch at n = 500
        assert np.array_equal(oracle80.keyswitch(okeys80, g["ks_in"][i]), g["ks_out"][i]), i
    bk = oracle80.bkfft(okeys80)
    for r, i in enumerate(g["extmul_index"]):    # tGswToFFTConvert, tGswFFTExternMulToTLwe
        assert np.array_equal(bk[i], g["bkfft_rows"][r]), i
        assert np.array_equal(oracle80.extern_mul(okeys80, g["extmul_acc"][r], int(i)), g["extmul_out"][r]), i


@pytest.mark.gpu
def test_gpu_replays_the_reference_stages(stages80, engine80):
    g = stages80
    assert np.array_equal(engine80.keyswitch_batch(g["ks_in"]), g["ks_out"])
    assert np.array_equal(engine80.get_bkfft()[g["extmul_index"]], g["bkfft_rows"])
    assert np.array_equal(engine80.extern_mul_batch(g["extmul_acc"], g["extmul_index"]), g["extmul_out"])
    assert np.array_equal(engine80.decomp_batch(g["decomp_in"]), g["decomp_out"])     # Torus32PolynomialDecompH_old, l = 2, Bgbit = 10

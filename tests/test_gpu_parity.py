"""GPU parity: every C-ABI entry point of libtfhe_hip.so against the oracle, BIT FOR BIT, on seeded inputs.

The oracle (oracle/tfhe_oracle.c) is itself pinned bit-for-bit to the reference (tests/test_oracle_vs_ref.py in the
authoring container, tests/golden/*.npz elsewhere).  The bar for this whole file is exact equality - spectra as IEEE
doubles, ciphertexts as int32 - because the kernels perform the reference's floating-point operations one for one
(DESIGN.md "Arithmetic contract")."""
import numpy as np
import pytest

from tfhe_amd import Gate, PARAMS as P

pytestmark = pytest.mark.gpu


def test_key_conversion_bit_exact(engine, oracle, okeys):
    # tGswToFFTConvert on the GPU == reference/oracle conversion (reference lwe-bootstrapping-functions-fft.cpp:47-50)
    got = engine.get_bkfft()
    want = oracle.bkfft(okeys)
    assert np.array_equal(got, want)


def test_forward_transforms_bit_exact(engine, oracle):
    rng = np.random.default_rng(10)
    ints = rng.integers(-64, 64, (16, P.N)).astype(np.int32)
    ints[0] = 0
    ints[1] = 63
    ints[2] = -64
    got = engine.int_ifft(ints)
    for i in range(ints.shape[0]):
        assert np.array_equal(got[i], oracle.int_ifft(ints[i])), i
    tor = rng.integers(-2**31, 2**31, (16, P.N)).astype(np.int32)
    tor[0] = -2**31
    tor[1] = 2**31 - 1
    got = engine.torus_ifft(tor)
    for i in range(tor.shape[0]):
        assert np.array_equal(got[i], oracle.torus_ifft(tor[i])), i


def test_inverse_transform_bit_exact(engine, oracle):
    rng = np.random.default_rng(11)
    tor = rng.integers(-2**31, 2**31, (12, P.N)).astype(np.int32)
    spec = np.stack([oracle.torus_ifft(t) for t in tor]) * rng.integers(1, 90000, (12, 1, 1))
    spec[0] = 0.0
    got = engine.torus_fft(spec)
    for i in range(spec.shape[0]):
        assert np.array_equal(got[i], oracle.torus_fft(spec[i])), i


def test_negacyclic_mul_bit_exact_and_vs_naive(engine, oracle):
    # distribution of reference src/test/test-multiplication.cpp:41-44; FFT product within 1 LSB of the exact
    # product (:66-77) and identical to the reference's FFT product
    rng = np.random.default_rng(12)
    a = rng.integers(-2 * P.N, 2 * P.N, (24, P.N)).astype(np.int32)
    b = rng.integers(-2**31, 2**31, (24, P.N)).astype(np.int32)
    got = engine.negacyclic_mul(a, b)
    for i in range(a.shape[0]):
        assert np.array_equal(got[i], oracle.mul_fft(a[i], b[i])), i
        d = (got[i].astype(np.int64) - oracle.mul_naive(a[i], b[i]).astype(np.int64) + 2**31) % 2**32 - 2**31
        assert np.abs(d).max() <= 1


def test_negacyclic_mul_2048_vs_naive(engine, oracle):
    """BASELINE configs[4], N = 2048: inputs as src/test/test-multiplication.cpp:41-44 (a in [-2N, 2N), b uniform); the
    oracle is the reference's naive product (no reference transform exists for N = 2048).  The GPU composes four
    N = 1024 FFT products, so it must (i) equal the same composition of the oracle's N = 1024 FFT products exactly and
    (ii) stay within 2 LSB of the exact product"""
    rng = np.random.default_rng(12)
    N2, cnt = 2 * P.N, 5
    a = rng.integers(-2 * N2, 2 * N2, (cnt, N2)).astype(np.int32)
    b = rng.integers(-2**31, 2**31, (cnt, N2)).astype(np.int32)
    a[0], b[1] = 0, 0
    got = engine.negacyclic_mul2048(a, b)
    for g in range(cnt):
        ae, ao, be, bo = (np.ascontiguousarray(x) for x in (a[g, 0::2], a[g, 1::2], b[g, 0::2], b[g, 1::2]))
        p0, p1, p2, p3 = (oracle.mul_fft(x, y).astype(np.int64) for x, y in ((ae, be), (ao, bo), (ae, bo), (ao, be)))
        want = np.empty(N2, np.int64)
        want[0::2] = p0 + np.concatenate(([-p1[-1]], p1[:-1]))
        want[1::2] = p2 + p3
        want = ((want + 2**31) % 2**32 - 2**31).astype(np.int32)
        assert np.array_equal(got[g], want), g
        exact = oracle.mul_naive(a[g], b[g]).astype(np.int64)
        d = (got[g].astype(np.int64) - exact + 2**31) % 2**32 - 2**31
        assert np.abs(d).max() <= 2, (g, np.abs(d).max())
    assert not got[0].any() and not got[1].any()


def test_external_product_bit_exact(engine, oracle, okeys):
    rng = np.random.default_rng(13)
    acc = rng.integers(-2**31, 2**31, (8, 2, P.N)).astype(np.int32)
    idx = np.array([0, 1, 2, 100, 314, 500, 628, 629], np.int32)
    got = engine.extern_mul_batch(acc, idx)
    for i in range(8):
        assert np.array_equal(got[i], oracle.extern_mul(okeys, acc[i], int(idx[i]))), i


def test_blind_rotate_seam_bit_exact(engine, oracle, okeys):
    # the reference's own device seam: tfhe_blindRotate_FFT(accum, bkFFT, bara, n)
    rng = np.random.default_rng(14)
    acc = rng.integers(-2**31, 2**31, (3, 2, P.N)).astype(np.int32)
    bara = rng.integers(0, 2 * P.N, (3, P.n)).astype(np.int32)
    bara[1, :17] = 0          # the skipped iterations (lwe-bootstrapping-functions-fft.cpp:154)
    bara[2, ::2] = 2 * P.N - 1
    got = engine.blind_rotate_batch(acc, bara)
    for i in range(3):
        assert np.array_equal(got[i], oracle.blind_rotate(okeys, acc[i], bara[i])), i


def test_blind_rotate_rotates_the_message(engine, keys, oracle):
    """the algebra the reference's unit test checks with fake samples (src/test/bootstrapping_test_fft.cpp:99-144),
    here with the real noisy key and no oracle in the loop: blind rotation of the trivial sample (0, msg) by bara must
    decrypt, under the TLWE key, to X^(sum_i s_i bara_i) * msg up to the bootstrapping-key noise"""
    rng = np.random.default_rng(21)
    cnt = 3
    msg = rng.integers(-2**31, 2**31, (cnt, P.N)).astype(np.int32)
    acc = np.zeros((cnt, 2, P.N), np.int32)
    acc[:, 1] = msg
    bara = rng.integers(0, 2 * P.N, (cnt, P.n)).astype(np.int32)
    out = engine.blind_rotate_batch(acc, bara)
    s = keys.tlwe_key.astype(np.int32)
    for g in range(cnt):
        a_s = oracle.mul_naive(s, out[g, 0])                      # exact negacyclic product: key (binary) * mask
        phase = (out[g, 1].astype(np.int64) - a_s.astype(np.int64))
        shift = int((keys.lwe_key.astype(np.int64) * bara[g]).sum() % (2 * P.N))
        j = np.arange(P.N)
        src = (j - shift) % (2 * P.N)
        want = np.where(src >= P.N, -msg[g, src % P.N].astype(np.int64), msg[g, src % P.N].astype(np.int64))
        err = (phase - want + 2**31) % 2**32 - 2**31
        # predicted standard deviation: sqrt(n (k+1) l N (Bg/2)^2) * 2^-25 = 2^-8 of the torus = 2^24 LSB
        rms = float(np.sqrt((err.astype(np.float64) ** 2).mean()))
        assert 2**22.5 < rms < 2**25, (g, rms)
        assert np.abs(err).max() < 2**27, (g, int(np.abs(err).max()))


def test_keyswitch_bit_exact(engine, oracle, okeys):
    rng = np.random.default_rng(15)
    for count in (1, 5, 32, 33, 70):      # ragged around the 32-ciphertext tile
        u = rng.integers(-2**31, 2**31, (count, P.N + 1)).astype(np.int32)
        u[0, :10] = 0                     # all-zero digits
        got = engine.keyswitch_batch(u)
        for i in sorted({0, count // 2, count - 1}):
            assert np.array_equal(got[i], oracle.keyswitch(okeys, u[i])), (count, i)


def test_keyswitch_preserves_the_phase(engine, keys):
    """what src/test/lwekeyswitch_test.cpp:96-152 checks with a noiseless key, here with the real one and no oracle:
    the phase of the switched sample under the LWE key equals the phase of the input under the extracted (TLWE) key
    up to keyswitch noise (sigma 2^-15 per selected row, ~6144 rows) and the 2^-17 rounding of each coefficient"""
    rng = np.random.default_rng(16)
    u = rng.integers(-2**31, 2**31, (64, P.N + 1)).astype(np.int32)
    out = engine.keyswitch_batch(u)
    s_in, s_out = keys.tlwe_key.astype(np.int64), keys.lwe_key.astype(np.int64)
    ph_in = u[:, P.N].astype(np.int64) - (u[:, :P.N].astype(np.int64) * s_in).sum(axis=1)
    ph_out = out[:, P.n].astype(np.int64) - (out[:, :P.n].astype(np.int64) * s_out).sum(axis=1)
    err = (ph_out - ph_in + 2**31) % 2**32 - 2**31
    # predicted sigma: sqrt(6144) * 2^-15 = 2^-8.7 of the torus = 2^23.3 LSB
    assert np.abs(err).max() < 2**26.5, int(np.abs(err).max())
    assert 2**21.5 < np.sqrt((err.astype(np.float64) ** 2).mean()) < 2**24.5


def test_bootstrap_bit_exact(engine, oracle, okeys, client, keys):
    bits = np.array([0, 1, 1, 0, 1], np.uint8)
    x = client.encrypt(keys, bits, seed=21)
    got_woks = engine.bootstrap_batch(x, with_ks=False)
    got = engine.bootstrap_batch(x, with_ks=True)
    for i in range(2):
        assert np.array_equal(got_woks[i], oracle.bootstrap_woks(okeys, x[i])), i
        assert np.array_equal(got[i], oracle.bootstrap(okeys, x[i])), i
    dec, _ = client.decrypt(keys, got)
    assert np.array_equal(dec, bits)
    dec, _ = client.decrypt(keys, got_woks, extracted=True)
    assert np.array_equal(dec, bits)


from tests.truth import TRUTH  # noqa: E402


@pytest.mark.parametrize("gate", list(Gate))
def test_every_gate_bit_exact_and_truth_table(gate, engine, oracle, okeys, client, keys):
    # 8 input combinations = the full truth table of a 3-input gate (reference boots_gates_test.cpp:59-91)
    a = np.array([0, 1, 0, 1, 0, 1, 0, 1], np.uint8)
    b = np.array([0, 0, 1, 1, 0, 0, 1, 1], np.uint8)
    c = np.array([0, 0, 0, 0, 1, 1, 1, 1], np.uint8)
    ca, cb, cc = client.encrypt(keys, a, 31), client.encrypt(keys, b, 32), client.encrypt(keys, c, 33)
    got = engine.gate_batch(gate, ca, cb, cc)
    dec, phases = client.decrypt(keys, got)
    assert np.array_equal(dec, TRUTH[int(gate)](a.astype(np.int64), b.astype(np.int64), c.astype(np.int64)))
    ncheck = 8 if gate >= Gate.NOT else 2
    for i in range(ncheck):
        assert np.array_equal(got[i], oracle.gate(okeys, gate, ca[i], cb[i], cc[i])), i


def test_batch_of_independent_nands_decrypts(engine, client, keys, oracle, okeys):
    # config "batch of independent bootsNAND gates", reduced; every output must decrypt to NAND and sampled
    # ciphertexts must equal the oracle's (reference test-gate-bootstrapping.cpp:78-88)
    rng = np.random.default_rng(40)
    count = 1000                       # not a multiple of the 4-/8-wave workgroup nor of the keyswitch tile
    a = rng.integers(0, 2, count).astype(np.uint8)
    b = rng.integers(0, 2, count).astype(np.uint8)
    ca, cb = client.encrypt(keys, a, 41), client.encrypt(keys, b, 42)
    got = engine.gate_batch(Gate.NAND, ca, cb)
    dec, phases = client.decrypt(keys, got)
    assert np.array_equal(dec, 1 - (a & b))
    assert np.abs(np.abs(phases.astype(np.int64)) - (1 << 29)).max() < (1 << 28)   # |phase -+ MU| < 1/16
    for i in (0, 499, 999):
        assert np.array_equal(got[i], oracle.gate(okeys, Gate.NAND, ca[i], cb[i])), i


def test_empty_batch_and_argument_errors(engine):
    from tfhe_amd import TfheHipError
    out = engine.gate_batch(Gate.NAND, np.zeros((0, P.n + 1), np.int32), np.zeros((0, P.n + 1), np.int32))
    assert out.shape == (0, P.n + 1)
    with pytest.raises(TfheHipError):
        engine.gate_batch(99, np.zeros((1, P.n + 1), np.int32), np.zeros((1, P.n + 1), np.int32))
    with pytest.raises(TfheHipError):
        engine.gate_batch(Gate.NAND, np.zeros((1, P.n + 1), np.int32), None, count=1)
    with pytest.raises(TfheHipError):
        engine.extern_mul_batch(np.zeros((1, 2, P.N), np.int32), np.array([630], np.int32))


def test_full_size_batch_16384_and_mux(engine, client, keys, oracle, okeys):
    """BASELINE configs[1] at its full size: 16,384 independent NAND gates, every output decrypted, sampled
    ciphertexts identical to the oracle; plus a MUX batch (configs[4]) whose outputs all decrypt to a ? b : c."""
    rng = np.random.default_rng(77)
    count = 16384
    a, b, c = (rng.integers(0, 2, count).astype(np.uint8) for _ in range(3))
    ca, cb = client.encrypt(keys, a, 301), client.encrypt(keys, b, 302)
    got = engine.gate_batch(Gate.NAND, ca, cb)
    dec, phases = client.decrypt(keys, got)
    assert np.array_equal(dec, 1 - (a & b))
    assert np.abs(np.abs(phases.astype(np.int64)) - (1 << 29)).max() < (1 << 28)
    pick = np.unique(np.concatenate([np.arange(96), rng.integers(0, count, 64), np.arange(count - 48, count)]))
    assert pick.shape[0] >= 192
    assert np.array_equal(got[pick], oracle.gate_batch(okeys, Gate.NAND, ca[pick], cb[pick], None, nthreads=16))
    # size-independent property: a bootstrapped gate's output depends only on its own inputs (batch position and
    # batch size do not matter): re-running a slice reproduces the same ciphertexts bit for bit
    again = engine.gate_batch(Gate.NAND, ca[5000:5100], cb[5000:5100])
    assert np.array_equal(again, got[5000:5100])
    m = 4096 + 37   # host-buffer calls above 4,096 gates are pipelined in chunks: a ragged second chunk, three operands
    cc = client.encrypt(keys, c[:m], 303)
    mux = engine.gate_batch(Gate.MUX, ca[:m], cb[:m], cc)
    dec, _ = client.decrypt(keys, mux)
    assert np.array_equal(dec, np.where(a[:m] == 1, b[:m], c[:m]))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["1", "4", "6", "8"])
def test_every_launch_shape_gives_the_reference_ciphertexts(shape, keys, monkeypatch):
    """the blind rotation picks waves-per-workgroup by batch size (6 = one workgroup per ciphertext); every shape,
    forced here through TFHE_HIP_BR_WAVES, must reproduce the reference's outputs of tests/golden/gates.npz"""
    import os
    from tfhe_amd import Engine, Gate
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "gates.npz"))
    monkeypatch.setenv("TFHE_HIP_BR_WAVES", shape)
    eng = Engine(0)
    try:
        eng.load_keys_coef(keys.bk_coef, keys.ks)
        for gate in (Gate.NAND, Gate.XOR, Gate.ANDYN):
            out = eng.gate_batch(gate, g["ca"][:4], g["cb"][:4])
            assert np.array_equal(out, g["out"][int(gate), :4]), (shape, gate)
        out = eng.gate_batch(Gate.MUX, g["ca"], g["cb"], g["cc"])
        assert np.array_equal(out, g["out"][int(Gate.MUX)]), shape
    finally:
        eng.close()


@pytest.mark.gpu
def test_bulk_encrypt_and_phase_on_the_gpu(engine, client, keys, oracle):
    """SURVEY 8(f) N3, the part that parallelises: bootsSymEncrypt (tfhe_gate_bootstrapping.cpp:154-159) and lwePhase
    (lwe-functions.cpp:63-72) in bulk.  The ciphertexts equal the portable client's (tfhe_amd/csrc/client_keys.c) bit
    for bit - the reference's own generator is not portable, so against the reference the parity is semantic: the
    oracle's lwePhase (pinned to the reference's in tests/test_oracle_vs_ref.py) recovers +-1/8 with noise 2^-15."""
    rng = np.random.default_rng(5)
    for count in (1, 63, 64, 65, 1000, 4099):            # ragged tails of the 64-sample tiles
        bits = rng.integers(0, 2, count).astype(np.uint8)
        want = client.encrypt(keys, bits, seed=77)
        got = engine.encrypt_batch(77, keys.lwe_key, bits)
        assert np.array_equal(got, want), count
    # first_index continues a stream: the second half of a batch can be produced by a second call
    tail = engine.encrypt_batch(77, keys.lwe_key, bits[2000:], first_index=2000)
    assert np.array_equal(tail, want[2000:])
    ph = engine.phase_batch(keys.lwe_key, want)
    for i in (0, 1, 64, 4098):
        assert int(ph[i]) == oracle.phase(want[i], keys.lwe_key)
    _, ph_client = client.decrypt(keys, want)
    assert np.array_equal(ph, ph_client)
    assert np.array_equal((ph > 0).astype(np.uint8), bits)
    noise = ph.astype(np.int64) - np.where(bits == 1, 1 << 29, -(1 << 29))
    assert 0.9 * 2 ** 17 < noise.std() < 1.1 * 2 ** 17 and abs(noise.mean()) < 2 ** 13
    # device-resident output feeds a level directly: fresh ciphertexts born in the pool
    d_pool = engine.dev_alloc(3 * (P.n + 1) * 4)
    engine.encrypt_batch_dev(d_pool, 9, keys.lwe_key, np.array([1, 1], np.uint8))
    engine.gate_level_dev(d_pool, 3, [[int(Gate.NAND), 2, 0, 1, -1]])
    res = engine.from_dev(d_pool, (3, P.n + 1))
    engine.dev_free(d_pool)
    assert np.array_equal(res[:2], client.encrypt(keys, np.array([1, 1], np.uint8), seed=9))
    assert int(engine.phase_batch(keys.lwe_key, res[2:])[0]) < 0
    assert engine.encrypt_batch(1, keys.lwe_key, np.zeros(0, np.uint8)).shape == (0, P.n + 1)


@pytest.mark.gpu
def test_bootstrapping_key_generated_on_the_gpu(client, keys, engine):
    """tfhe_createLweBootstrappingKey (lwe-bootstrapping-functions.cpp:185-217) on the GPU.  The
    coefficient-form key equals the portable client's for the same seed bit for bit (3,780 TLWE rows), the transformed
    key installed on the device equals the one the session engine derived from the client's key, and a context keyed
    this way evaluates gates to the same ciphertexts."""
    import time
    from tfhe_amd import Engine
    e2 = Engine(0)
    try:
        t0 = time.perf_counter()
        coef = e2.keygen_bk(42, keys.lwe_key, keys.tlwe_key, keys.ks, want_coef=True)
        dt = time.perf_counter() - t0
        assert np.array_equal(coef, np.asarray(keys.bk_coef).reshape(coef.shape))
        assert np.array_equal(e2.get_bkfft().view(np.int64), engine.get_bkfft().view(np.int64))
        bits = np.random.default_rng(3).integers(0, 2, (2, 64)).astype(np.uint8)
        a, b = client.encrypt(keys, bits[0], seed=31), client.encrypt(keys, bits[1], seed=32)
        assert np.array_equal(e2.gate_batch(Gate.XOR, a, b), engine.gate_batch(Gate.XOR, a, b))
        # a different seed gives a different key that still bootstraps correctly (same secret keys, fresh randomness)
        e2.keygen_bk(43, keys.lwe_key, keys.tlwe_key, keys.ks)
        out = e2.gate_batch(Gate.XOR, a, b)
        assert not np.array_equal(out, engine.gate_batch(Gate.XOR, a, b))
        assert np.array_equal(client.decrypt(keys, out)[0], bits[0] ^ bits[1])
        print("keygen_bk %.3f s" % dt)
    finally:
        e2.close()


@pytest.mark.gpu
@pytest.mark.parametrize("count", [257, 700, 769, 1025])
def test_batch_sizes_around_the_launch_shape_thresholds(count, engine, client, keys, oracle, okeys):
    """launch_br picks the latency kernel up to 768 gates (one to three rounds of 256 workgroups), 4 waves per workgroup
    up to 1,024 and 8 above; sizes next to the thresholds, with ragged last rounds / workgroups: every output decrypts,
    a sample of them equals the reference's ciphertexts"""
    rng = np.random.default_rng(count)
    bits = rng.integers(0, 2, (2, count)).astype(np.uint8)
    a, b = client.encrypt(keys, bits[0], seed=60), client.encrypt(keys, bits[1], seed=61)
    out = engine.gate_batch(Gate.ANDNY, a, b)
    assert np.array_equal(client.decrypt(keys, out)[0], TRUTH[int(Gate.ANDNY)](bits[0], bits[1], None))
    pick = np.array([0, 255, 256, count // 2, count - 2, count - 1])
    want = oracle.gate_batch(okeys, Gate.ANDNY, a[pick], b[pick], None, nthreads=6)
    assert np.array_equal(out[pick], want)


@pytest.mark.gpu
def test_whole_cloud_keyset_generated_on_the_gpu(client, keys, engine):
    """SURVEY 8(f) N3 completed: lweCreateKeySwitchKey (lwe-keyswitch-functions.cpp:163-215) on the GPU as well.  The
    client draws the 24,576 keyswitch samples from one sequential xoshiro256** stream; the kernel's threads jump to
    their positions with powers of the generator's GF(2) transition matrix.  Both keys equal the client's bit for bit,
    and a context keyed this way produces the same ciphertexts."""
    import time
    from tfhe_amd import Engine
    e2 = Engine(0)
    try:
        t0 = time.perf_counter()
        bk, ks = e2.keygen_cloud(42, keys.lwe_key, keys.tlwe_key, want_keys=True)
        dt = time.perf_counter() - t0
        assert np.array_equal(ks, np.asarray(keys.ks).reshape(ks.shape))
        assert np.array_equal(bk, np.asarray(keys.bk_coef).reshape(bk.shape))
        bits = np.random.default_rng(4).integers(0, 2, (2, 40)).astype(np.uint8)
        a, b = client.encrypt(keys, bits[0], seed=33), client.encrypt(keys, bits[1], seed=34)
        assert np.array_equal(e2.gate_batch(Gate.OR, a, b), engine.gate_batch(Gate.OR, a, b))
        t0 = time.perf_counter()
        e2.keygen_cloud(7, keys.lwe_key, keys.tlwe_key)
        dt2 = time.perf_counter() - t0
        assert np.array_equal(client.decrypt(keys, e2.gate_batch(Gate.OR, a, b))[0], bits[0] | bits[1])
        print("keygen_cloud %.3f s with host copies (first call builds the jump matrices), %.3f s without" % (dt, dt2))
    finally:
        e2.close()


@pytest.mark.gpu
@pytest.mark.parametrize("count", [4097, 6145])
def test_host_buffer_pipeline_chunk_boundaries(count, engine, client, keys):
    """tfhe_hip_gate_batch on host buffers: a first chunk of 2,048 gates, then chunks of 4,096, copies on a second
    stream; the result must equal the device-resident call on the same inputs whatever the tail chunk's size (1 gate
    for 6,145; 2,049 for 4,097)"""
    rng = np.random.default_rng(count)
    bits = rng.integers(0, 2, (2, count)).astype(np.uint8)
    a, b = client.encrypt(keys, bits[0], seed=70), client.encrypt(keys, bits[1], seed=71)
    host = engine.gate_batch(Gate.XNOR, a, b)
    d_a, d_b, d_o = (engine.dev_alloc(a.nbytes) for _ in range(3))
    engine.to_dev(d_a, a)
    engine.to_dev(d_b, b)
    engine.gate_batch_dev(Gate.XNOR, d_o, d_a, d_b, None, count=count)
    dev = engine.from_dev(d_o, a.shape)
    for p in (d_a, d_b, d_o):
        engine.dev_free(p)
    assert np.array_equal(host, dev)
    assert np.array_equal(client.decrypt(keys, host)[0], 1 - (bits[0] ^ bits[1]))


@pytest.mark.gpu
def test_host_pipeline_error_leaves_the_context_usable(engine, client, keys, monkeypatch):
    """a failure in the middle of the chunked host-buffer pipeline (injected at chunk 1 through
    TFHE_HIP_TEST_FAIL_CHUNK) is reported only after both streams have drained - nothing writes to the caller's buffers
    afterwards - and the same context then evaluates the same call correctly"""
    from tfhe_amd import Engine, TfheHipError
    count = 6145 + 4096
    rng = np.random.default_rng(8)
    bits = rng.integers(0, 2, (2, count)).astype(np.uint8)
    a, b = client.encrypt(keys, bits[0], seed=80), client.encrypt(keys, bits[1], seed=81)
    monkeypatch.setenv("TFHE_HIP_TEST_FAIL_CHUNK", "1")
    e2 = Engine(0)
    monkeypatch.delenv("TFHE_HIP_TEST_FAIL_CHUNK")
    try:
        e2.load_keys_coef(keys.bk_coef, keys.ks)
        with pytest.raises(TfheHipError, match="injected failure"):
            e2.gate_batch(Gate.OR, a, b)
        out = e2.gate_batch(Gate.OR, a, b)           # the hook fires once
        assert np.array_equal(out, engine.gate_batch(Gate.OR, a, b))
        assert np.array_equal(client.decrypt(keys, out)[0], bits[0] | bits[1])
    finally:
        e2.close()


def test_ragged_last_round_is_launched_on_its_own(engine, oracle, okeys, client, keys):
    """2 x 2,048 + 100 gates: the remainder beyond whole rounds of the throughput shape runs as a launch of its own
    (latency kernel); every row still equals the oracle's - sampled on both sides of the split - and decrypts"""
    rng = np.random.default_rng(91)
    count = 4096 + 100
    bits = rng.integers(0, 2, (3, count)).astype(np.uint8)
    ct = [client.encrypt(keys, bits[i], 900 + i) for i in range(3)]
    out = engine.gate_batch(Gate.MUX, *ct)
    dec, _ = client.decrypt(keys, out)
    assert np.array_equal(dec, np.where(bits[0] == 1, bits[1], bits[2]))
    pick = [0, 4095, 4096, 4097, count - 1]
    want = oracle.gate_batch(okeys, Gate.MUX, ct[0][pick], ct[1][pick], ct[2][pick], nthreads=5)
    assert np.array_equal(out[pick], want)
    acc = rng.integers(-2**31, 2**31, (2048 + 3, 2, P.N)).astype(np.int32)      # the reference seam as well
    bara = rng.integers(0, 2 * P.N, (2048 + 3, P.n)).astype(np.int32)
    got = engine.blind_rotate_batch(acc, bara)
    for g in (0, 2047, 2048, 2050):
        assert np.array_equal(got[g], oracle.blind_rotate(okeys, acc[g], bara[g])), g


@pytest.mark.parametrize("cnt", [3, 600, 1100, 2100])
def test_single_cmux_bit_exact(cnt, engine, oracle, okeys):
    """tfhe_MuxRotate_FFT on its own (lwe-bootstrapping-functions-fft.cpp:68-77): a blind rotation with ONE non-zero
    rotation amount is one CMux from the given accumulator - in the latency kernel (3), the one-wave-per-ciphertext
    shapes (600, 1100) and the throughput shape with a ragged tail (2100); rotation amounts include 1, N and 2N - 1"""
    rng = np.random.default_rng(100 + cnt)
    acc = rng.integers(-2**31, 2**31, (cnt, 2, P.N)).astype(np.int32)
    bara = np.zeros((cnt, P.n), np.int32)
    pos = rng.integers(0, P.n, cnt)
    amt = rng.integers(1, 2 * P.N, cnt)
    pos[:3], amt[:3] = [0, P.n - 1, 314], [1, P.N, 2 * P.N - 1]
    bara[np.arange(cnt), pos] = amt
    got = engine.blind_rotate_batch(acc, bara)
    for g in sorted({0, 1, 2, cnt // 2, cnt - 1}):
        assert np.array_equal(got[g], oracle.blind_rotate(okeys, acc[g], bara[g])), g


def test_gadget_decomposition_equals_the_reference(engine, oracle):
    """tGswTorus32PolynomialDecompH as a stage (tfhe_hip_tgsw_decomp_batch): the reference's own outputs
    (tests/golden/primitives.npz), the oracle on random and extreme coefficients, and the reconstruction bound of
    src/test/test-decomp-tgsw.cpp:72-79"""
    import os
    prim = np.load(os.path.join(os.path.dirname(__file__), "golden", "primitives.npz"))
    assert np.array_equal(engine.decomp_batch(prim["torus_polys"][:prim["decomp"].shape[0]]), prim["decomp"])
    rng = np.random.default_rng(17)
    polys = rng.integers(-2**31, 2**31, (37, P.N)).astype(np.int32)
    polys[0, :6] = [0, -1, 1, 2**31 - 1, -2**31, 1 << 10]
    got = engine.decomp_batch(polys)
    assert got.min() >= -64 and got.max() < 64
    for i in (0, 1, 36):
        assert np.array_equal(got[i], oracle.decomp(polys[i]))
    rec = sum(got[:, p].astype(np.int64) << (32 - 7 * (p + 1)) for p in range(3))
    err = (rec - polys.astype(np.int64) + 2**31) % 2**32 - 2**31
    assert err.max() <= 0 and err.min() > -(1 << 11)     # the 11 dropped bits (the offset holds no rounding term)
    assert engine.decomp_batch(polys[:0]).shape == (0, 3, P.N)


def test_gpu_replays_the_reference_golden_vectors(engine, keys):
    """the reference's own outputs (tests/golden/primitives.npz, bootstrap.npz, written by oracle/gen_golden.py from
    /root/reference) against the GPU directly, with no oracle in between: the three transforms, the FFT product, key
    conversion, external product, one CMux (tfhe_MuxRotate_FFT), whole blind rotations, bootstrap without and with
    keyswitch, keyswitch on random samples"""
    import os
    gold = os.path.join(os.path.dirname(__file__), "golden")
    prim, boot = np.load(os.path.join(gold, "primitives.npz")), np.load(os.path.join(gold, "bootstrap.npz"))
    assert int(boot["key_seed"][0]) == keys.seed
    assert np.array_equal(engine.int_ifft(prim["int_polys"]), prim["int_ifft"])          # IEEE doubles, compared exactly
    assert np.array_equal(engine.torus_ifft(prim["torus_polys"]), prim["torus_ifft"])
    assert np.array_equal(engine.torus_fft(prim["spectra"]), prim["torus_fft"])
    assert np.array_equal(engine.negacyclic_mul(prim["mul_a"], prim["mul_b"]), prim["mul_fft"])
    assert np.array_equal(engine.get_bkfft()[boot["bkfft_index"]], boot["bkfft_rows"])
    assert np.array_equal(engine.extern_mul_batch(boot["extmul_acc"], boot["extmul_index"]), boot["extmul_out"])
    bara = np.zeros((4, P.n), np.int32)
    bara[np.arange(4), boot["extmul_index"]] = boot["mux_bara"]
    keep = boot["mux_bara"] != 0                               # a rotation by 0 is no CMux at all in a blind rotation
    assert np.array_equal(engine.blind_rotate_batch(boot["extmul_acc"], bara)[keep], boot["mux_out"][keep])
    assert np.array_equal(engine.blind_rotate_batch(boot["br_acc"], boot["br_bara"]), boot["br_out"])
    assert np.array_equal(engine.bootstrap_batch(boot["bs_in"], with_ks=False), boot["bs_woks"])
    assert np.array_equal(engine.bootstrap_batch(boot["bs_in"], with_ks=True), boot["bs_out"])
    assert np.array_equal(engine.keyswitch_batch(boot["bs_woks"]), boot["ks_out"])
    assert np.array_equal(engine.keyswitch_batch(boot["ks_rand_in"]), boot["ks_rand_out"])

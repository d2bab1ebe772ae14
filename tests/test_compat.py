"""Struct-level libtfhe shim (include/tfhe_hip_compat.h): layout of the struct views and, on the GPU, the libtfhe-style
bootsXXX_batch functions and the reference seam tfhe_hip_blindRotate_FFT driven through pointer-rich structs."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from tests.oracle_lib import P, have_ref
from tfhe_amd import PARAMS as PR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HARNESS = os.path.join(ROOT, "tests", "compat", "libcompat_harness.so")

REF_LAYOUT_PROG = r'''
#include <cstdio>
#include <cstddef>
#include "tfhe.h"
int main() {
    long long o[] = { sizeof(LweSample), offsetof(LweSample, b), offsetof(LweSample, current_variance),
      sizeof(LweParams), sizeof(TLweParams), offsetof(TLweParams, extracted_lweparams),
      sizeof(TGswParams), offsetof(TGswParams, tlwe_params), offsetof(TGswParams, offset),
      sizeof(LweKeySwitchKey), offsetof(LweKeySwitchKey, ks),
      sizeof(TLweSample), sizeof(TLweSampleFFT), sizeof(TGswSampleFFT),
      sizeof(LweBootstrappingKeyFFT), offsetof(LweBootstrappingKeyFFT, bkFFT), offsetof(LweBootstrappingKeyFFT, ks),
      sizeof(TFheGateBootstrappingParameterSet), sizeof(TFheGateBootstrappingCloudKeySet),
      offsetof(TFheGateBootstrappingCloudKeySet, bkFFT), sizeof(LagrangeHalfCPolynomial), sizeof(TorusPolynomial) };
    for (auto v : o) printf("%lld\n", v);
}
'''


def harness_layout():
    lib = ctypes.CDLL(HARNESS)
    o = np.zeros(22, np.int64)
    lib.harness_layout(P(o))
    return o.tolist()


def test_views_have_plausible_layout():
    o = harness_layout()
    assert o[0] == 24 and o[1] == 8 and o[2] == 16          # LweSample {ptr, int32, double}
    assert o[20] == 8192 and o[21] == 4096                    # LagrangeHalfCPolynomial, TorusPolynomial


@pytest.mark.ref
@pytest.mark.skipif(not have_ref(), reason="reference headers absent")
def test_views_match_reference_headers(tmp_path):
    src = tmp_path / "layout.cpp"
    src.write_text(REF_LAYOUT_PROG)
    exe = str(tmp_path / "layout")
    subprocess.run(["g++", "-std=c++11", "-w", "-Wno-invalid-offsetof", "-I/root/reference/src/include", str(src),
                    "-o", exe], check=True)
    ref = [int(x) for x in subprocess.run([exe], check=True, capture_output=True, text=True).stdout.split()]
    assert ref == harness_layout()


@pytest.fixture(scope="module")
def keyset(oracle, okeys, keys):
    lib = ctypes.CDLL(HARNESS)
    lib.harness_make_keyset.restype = ctypes.c_void_p
    bkfft = np.ascontiguousarray(oracle.bkfft(okeys))
    h = ctypes.c_void_p(lib.harness_make_keyset(P(bkfft), P(np.ascontiguousarray(keys.ks))))
    yield lib, h
    ctypes.CDLL(os.path.join(ROOT, "tfhe_amd", "libtfhe_hip.so")).tfhe_hip_default_context_reset()
    lib.harness_free_keyset(h)


@pytest.mark.gpu
def test_boots_batch_functions_on_reference_structs(keyset, client, keys, oracle, okeys):
    from tests.truth import TRUTH
    lib, h = keyset
    a = np.array([0, 1, 0, 1, 0, 1, 0, 1], np.uint8)
    b = np.array([0, 0, 1, 1, 0, 0, 1, 1], np.uint8)
    c = np.array([0, 0, 0, 0, 1, 1, 1, 1], np.uint8)
    ca, cb, cc = client.encrypt(keys, a, 31), client.encrypt(keys, b, 32), client.encrypt(keys, c, 33)
    for gate in range(15):
        out = np.zeros((8, PR.n + 1), np.int32)
        lib.harness_gate(h, gate, P(out), P(ca), P(cb), P(cc), 8)
        dec, _ = client.decrypt(keys, out)
        assert np.array_equal(dec, TRUTH[gate](a.astype(np.int64), b.astype(np.int64), c.astype(np.int64))), gate
        assert np.array_equal(out[5], oracle.gate(okeys, gate, ca[5], cb[5], cc[5])), gate


@pytest.mark.gpu
def test_reference_seam_signature(keyset, oracle, okeys):
    lib, h = keyset
    rng = np.random.default_rng(8)
    acc = rng.integers(-2**31, 2**31, (2, PR.N)).astype(np.int32)
    bara = rng.integers(0, 2 * PR.N, PR.n).astype(np.int32)
    want = oracle.blind_rotate(okeys, acc, bara)
    lib.harness_blind_rotate(h, P(acc), P(bara))
    assert np.array_equal(acc, want)


@pytest.mark.gpu
def test_chunked_struct_path_on_a_large_ragged_batch(keyset, client, keys, engine):
    """tfhe_hip_gate_batch_samples packs LweSample arrays chunk by chunk (4,096) with a helper thread; a batch that
    spans three chunks with a ragged tail must equal the flat entry point's output row for row"""
    lib, h = keyset
    lib.harness_time_nand.restype = ctypes.c_double
    count = 2 * 4096 + 123
    rng = np.random.default_rng(77)
    a, b = rng.integers(0, 2, count).astype(np.uint8), rng.integers(0, 2, count).astype(np.uint8)
    ca, cb = client.encrypt(keys, a, 41), client.encrypt(keys, b, 42)
    out = np.zeros((count, PR.n + 1), np.int32)
    ms = lib.harness_time_nand(h, P(ca), P(cb), P(out), count, 1)
    from tfhe_amd import Gate
    assert np.array_equal(out, engine.gate_batch(Gate.NAND, ca, cb))
    dec, _ = client.decrypt(keys, out)
    assert np.array_equal(dec, 1 - (a & b))
    print("struct-level bootsNAND_batch: %d gates in %.1f ms" % (count, ms))


@pytest.mark.gpu
def test_key_swapped_behind_the_same_pointer_is_noticed(keyset, client, oracle, okeys, keys):
    """the shim keeps the uploaded key across calls (the reference re-copies bkFFT every call,
    lwe-bootstrapping-functions-fft.cpp:112-133): a different key behind the SAME keyset pointer must be uploaded again
    (content fingerprint), not silently ignored"""
    from tfhe_amd import Gate
    lib, h = keyset
    a = np.array([0, 1, 0, 1], np.uint8)
    b = np.array([0, 0, 1, 1], np.uint8)
    ca, cb = client.encrypt(keys, a, 61), client.encrypt(keys, b, 62)
    out = np.zeros((4, PR.n + 1), np.int32)
    lib.harness_gate(h, int(Gate.NAND), P(out), P(ca), P(cb), None, 4)
    assert np.array_equal(out[2], oracle.gate(okeys, Gate.NAND, ca[2], cb[2]))
    keys2 = client.keygen(43)                                    # other secret keys, other cloud keys
    ok2 = oracle.keys_from_coef(keys2.bk_coef, keys2.ks)
    try:
        lib.harness_overwrite_keyset(h, P(np.ascontiguousarray(oracle.bkfft(ok2))), P(np.ascontiguousarray(keys2.ks)))
        ca2, cb2 = client.encrypt(keys2, a, 63), client.encrypt(keys2, b, 64)
        lib.harness_gate(h, int(Gate.NAND), P(out), P(ca2), P(cb2), None, 4)
        assert np.array_equal(client.decrypt(keys2, out)[0], 1 - (a & b))
        assert np.array_equal(out[1], oracle.gate(ok2, Gate.NAND, ca2[1], cb2[1]))
    finally:                                                     # the module's keyset fixture goes back to seed 42
        lib.harness_overwrite_keyset(h, P(np.ascontiguousarray(oracle.bkfft(okeys))), P(np.ascontiguousarray(keys.ks)))
    lib.harness_gate(h, int(Gate.NAND), P(out), P(ca), P(cb), None, 4)
    assert np.array_equal(out[2], oracle.gate(okeys, Gate.NAND, ca[2], cb[2]))


GATES_LIB = os.path.join(ROOT, "tfhe_amd", "libtfhe_hip_gates.so")
SINGLE = ["bootsNAND", "bootsOR", "bootsAND", "bootsXOR", "bootsXNOR", "bootsNOR", "bootsANDNY", "bootsANDYN", "bootsORNY",
          "bootsORYN", "bootsMUX", "bootsNOT", "bootsCOPY", "bootsCONSTANT"]


def test_gates_library_exports_the_reference_names():
    """libtfhe_hip_gates.so: exactly the reference's single-gate entry points and its device seam
    (src/include/tfhe_gate_bootstrapping_functions.h:44-87, src/include/tfhe.h:42), nothing else"""
    out = subprocess.run(["nm", "-D", "--defined-only", GATES_LIB], check=True, capture_output=True, text=True).stdout
    names = sorted(ln.split()[-1] for ln in out.splitlines() if " T " in ln)
    assert names == sorted(SINGLE + ["tfhe_blindRotate_FFT", "tfhe_bootstrap_FFT", "tfhe_bootstrap_woKS_FFT"])


@pytest.mark.gpu
def test_single_gate_drop_ins_equal_the_batched_calls(keyset, client, keys, oracle, okeys):
    """an unmodified gate-by-gate application (src/test/test-gate-bootstrapping.cpp:69-75) resolved against
    libtfhe_hip_gates.so: every bootsXXX call = the batched shim with a batch of one = the oracle, bit for bit"""
    from tests.truth import TRUTH
    lib, h = keyset
    gl = ctypes.CDLL(GATES_LIB)
    a = np.array([0, 1, 0, 1], np.uint8)
    b = np.array([0, 0, 1, 1], np.uint8)
    c = np.array([1, 0, 0, 1], np.uint8)
    ca, cb, cc = client.encrypt(keys, a, 41), client.encrypt(keys, b, 42), client.encrypt(keys, c, 43)
    for gate, name in enumerate(SINGLE):
        kind = 1 if name == "bootsMUX" else 2 if name in ("bootsNOT", "bootsCOPY") else 3 if name == "bootsCONSTANT" else 0
        fn = ctypes.cast(getattr(gl, name), ctypes.c_void_p)
        got, want = np.zeros((4, PR.n + 1), np.int32), np.zeros((4, PR.n + 1), np.int32)
        lib.harness_single_gates(h, fn, kind, P(got), P(ca), P(cb), P(cc), 4, 1)
        lib.harness_gate(h, gate, P(want), P(ca), P(cb), P(cc), 4)
        assert np.array_equal(got, want), name
        dec, _ = client.decrypt(keys, got)
        assert np.array_equal(dec, TRUTH[gate](a.astype(np.int64), b.astype(np.int64), c.astype(np.int64))), name
    assert np.array_equal(got[0][:PR.n], np.zeros(PR.n, np.int32))          # the last one was bootsCONSTANT(1): trivial sample
    one = np.zeros((1, PR.n + 1), np.int32)
    lib.harness_single_gates(h, ctypes.cast(gl.bootsXOR, ctypes.c_void_p), 0, P(one), P(ca[1:2]), P(cb[1:2]), None, 1, 0)
    assert np.array_equal(one[0], oracle.gate(okeys, 3, ca[1], cb[1]))
    rng = np.random.default_rng(8)
    acc = rng.integers(-2**31, 2**31, (2, PR.N)).astype(np.int32)
    bara = rng.integers(0, 2 * PR.N, PR.n).astype(np.int32)
    got = acc.copy()
    lib.harness_blind_rotate_fn(h, ctypes.cast(gl.tfhe_blindRotate_FFT, ctypes.c_void_p), P(got), P(bara))
    assert np.array_equal(got, oracle.blind_rotate(okeys, acc, bara))
    # tfhe_bootstrap_FFT / tfhe_bootstrap_woKS_FFT (src/include/tfhe.h:44-45), one sample per call
    x = client.encrypt(keys, np.array([1, 0, 1], np.uint8), 44)
    full, woks = np.zeros((3, PR.n + 1), np.int32), np.zeros((3, PR.N + 1), np.int32)
    lib.harness_bootstrap_fn(h, ctypes.cast(gl.tfhe_bootstrap_FFT, ctypes.c_void_p), 1, P(full), P(x), 1 << 29, 3)
    lib.harness_bootstrap_fn(h, ctypes.cast(gl.tfhe_bootstrap_woKS_FFT, ctypes.c_void_p), 0, P(woks), P(x), 1 << 29, 3)
    for i in range(3):
        assert np.array_equal(woks[i], oracle.bootstrap_woks(okeys, x[i]))
        assert np.array_equal(full[i], oracle.keyswitch(okeys, woks[i]))


REF_APP = os.path.join(ROOT, "oracle", "_ref", "ref_gate_app")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(REF_APP + "_gpu"), reason="oracle/_ref applications not built (make -C oracle ref)")
def test_reference_application_with_gates_from_the_gpu_library():
    """oracle/ref_app.cpp - reference key generation (Gaussian noise, the reference's own generator), reference
    encryption, a gate-by-gate circuit, reference decryption - built twice from the same source: linked against the
    reference build alone, and with libtfhe_hip_gates.so ahead of it on the link line.  Same keys and inputs in both
    runs (the reference's generator starts from its default seed): the fingerprint over all output ciphertexts must be
    identical, i.e. the GPU gates are a drop-in on keys this repository did not generate"""
    env = dict(os.environ, TFHE_HIP_ARITH="exact")
    cpu = subprocess.run([REF_APP, "3"], check=True, capture_output=True, text=True, timeout=300).stdout.split()
    gpu = subprocess.run([REF_APP + "_gpu", "3"], check=True, capture_output=True, text=True, timeout=300, env=env).stdout.split()
    assert cpu[:3] == gpu[:3] and cpu[0] == "gates=18" and cpu[1] == "correct=18", (cpu, gpu)
    ms = lambda w: float(w[3].split("=")[1])  # noqa: E731
    print("reference application: %.1f ms per gate on the CPU, %.1f ms with the GPU gates" % (ms(cpu), ms(gpu)))
    assert ms(gpu) < ms(cpu)
    fast = subprocess.run([REF_APP + "_gpu", "3"], check=True, capture_output=True, text=True, timeout=300,
                          env=dict(os.environ, TFHE_HIP_ARITH="fast")).stdout.split()
    assert fast[:2] == cpu[:2] and fast[2] != cpu[2]          # FAST: same decrypted bits, other ciphertexts


@pytest.mark.gpu
@pytest.mark.parametrize("prog", ["test-addition-boot", "test-gate-bootstrapping"])
@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "test-addition-boot_gpu")),
                    reason="oracle/_ref programs not built (make -C oracle ref)")
def test_the_references_own_test_programs_run_on_the_gpu_gates(prog):
    """src/test/test-addition-boot.cpp (4-bit full_adder_MUX, full_adder and comparison_MUX circuits, each verified by
    decryption; it prints ERROR!!! on a wrong bit) and src/test/test-gate-bootstrapping.cpp, compiled UNMODIFIED from the
    reference's sources: linked against the reference build alone and with libtfhe_hip_gates.so ahead of it.  The GPU
    variant must verify like the CPU one and take less time for its circuits"""
    import re
    exe = os.path.join(ROOT, "oracle", "_ref", prog)
    cpu = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=600).stdout
    gpu = subprocess.run([exe + "_gpu"], check=True, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, TFHE_HIP_ARITH="exact")).stdout
    assert "ERROR" not in cpu and "ERROR" not in gpu
    t = lambda out: [float(x) for x in re.findall(r"microsecs\)\.\.\. ([0-9.]+)", out)]  # noqa: E731
    assert len(t(cpu)) == len(t(gpu)) >= 1
    print("%s, microseconds per timed section: CPU %s, GPU gates %s" % (prog, t(cpu), t(gpu)))
    if len(t(gpu)) > 1:                                               # (the first timed section also creates the context and uploads the key)
        assert sum(t(gpu)[1:]) < sum(t(cpu)[1:])


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "test-bootstrapping-fft_gpu")),
                    reason="oracle/_ref programs not built (make -C oracle ref)")
def test_the_references_bootstrap_program_through_the_seam():
    """src/test/test-bootstrapping-fft.cpp, unmodified: its calls of tfhe_bootstrap_FFT (src/include/tfhe.h:45) resolve to
    libtfhe_hip_gates.so when that library comes first - the whole bootstrap on the GPU.  (Were only the seam replaced,
    the reference's tfhe_bootstrap_FFT would reach this library through its inner call of tfhe_blindRotate_FFT,
    lwe-bootstrapping-functions-fft.cpp:203: the relocation is a PLT slot, and that is how the first version of this test
    ran.)  The phases it prints must be the same text in both runs"""
    exe = os.path.join(ROOT, "oracle", "_ref", "test-bootstrapping-fft")
    cpu = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=600).stdout
    gpu = subprocess.run([exe + "_gpu"], check=True, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, TFHE_HIP_ARITH="exact")).stdout
    phases = lambda out: [ln for ln in out.splitlines() if ln.startswith("phase")]  # noqa: E731
    assert len(phases(cpu)) == 2 and phases(cpu) == phases(gpu), (cpu, gpu)
    # ... and the seam really is this library's: with a device that does not exist the GPU variant dies in the shim
    # (like die_dramatically), the CPU variant does not care
    bad = dict(os.environ, TFHE_HIP_DEVICE="99", TFHE_HIP_DEVICES="")
    assert subprocess.run([exe + "_gpu"], capture_output=True, text=True, timeout=600, env=bad).returncode != 0
    assert subprocess.run([exe], capture_output=True, text=True, timeout=600, env=bad).returncode == 0

"""The C-ABI library loads and exports every function include/tfhe_hip.h declares (no GPU compute here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tfhe_hip_\w+|boots\w+_batch)\s*\(", text)))


def test_exports_every_declared_symbol():
    from tfhe_amd.engine import EXPORTS, load_library
    lib = load_library()
    names = declared("tfhe_hip.h")
    assert len(names) >= 20
    for nm in names:
        assert hasattr(lib, nm), nm
    assert sorted(EXPORTS) == names          # the Python binding covers the whole header


def test_compat_header_symbols():
    hdr = os.path.join(ROOT, "include", "tfhe_hip_compat.h")
    if not os.path.exists(hdr):
        pytest.skip("compat layer not built")
    from tfhe_amd.engine import load_library
    lib = load_library()
    text = re.sub(r"/\*.*?\*/", "", open(hdr).read(), flags=re.S)
    for nm in set(re.findall(r"\b(tfhe_hip_\w+|boots[A-Z]\w*)\s*\(", text)):
        assert hasattr(lib, nm), nm


def test_no_device_is_an_error_not_a_crash():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from tfhe_amd import Engine, TfheHipError
    with pytest.raises(TfheHipError):
        Engine(0)


def test_null_arguments_are_rejected():
    from tfhe_amd.engine import load_library
    lib = load_library()
    lib.tfhe_hip_last_error.restype = ctypes.c_char_p
    assert lib.tfhe_hip_create(None, 0) < 0
    assert lib.tfhe_hip_sync(None) < 0
    assert lib.tfhe_hip_gate_batch(None, 0, None, None, None, None, ctypes.c_int64(1)) < 0
    assert b"null" in lib.tfhe_hip_last_error()
    lib.tfhe_hip_destroy(None)


def test_product_does_not_touch_the_oracle():
    """the shipped path must not import, link or load anything under oracle/"""
    for rel in ("tfhe_amd/engine.py", "tfhe_amd/client.py", "tfhe_amd/__init__.py", "tfhe_amd/csrc/tfhe_hip.hip",
                "tfhe_amd/csrc/br_core.h", "tfhe_amd/csrc/client_keys.c", "tfhe_amd/csrc/wire_io.cpp",
                "tfhe_amd/csrc/compat_libtfhe.cpp", "tfhe_amd/wire.py", "tfhe_amd/netlist.py", "tfhe_amd/parallel.py"):
        text = open(os.path.join(ROOT, rel)).read()
        assert "liboracle" not in text and "tfhe_oracle" not in text and "oracle_lib" not in text, rel


def _build_example(tmp_path, name="nand_batch"):
    import subprocess
    exe = str(tmp_path / name)
    subprocess.run(["gcc", "-O2", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "examples", name + ".c"), "-L" + os.path.join(ROOT, "tfhe_amd"), "-ltfhe_hip",
                    "-ltfhe_client", "-Wl,-rpath," + os.path.join(ROOT, "tfhe_amd"), "-o", exe], check=True)
    return exe


def test_plain_c_example_compiles_and_links(tmp_path):
    """examples/nand_batch.c: a C99 host against include/tfhe_hip.h alone (no C++ in the header, every symbol resolves)"""
    assert os.path.exists(_build_example(tmp_path))
    assert os.path.exists(_build_example(tmp_path, "group_nand"))


@pytest.mark.gpu
def test_plain_c_example_runs(tmp_path):
    import subprocess
    out = subprocess.run([_build_example(tmp_path), "300"], check=True, capture_output=True, text=True, timeout=120).stdout
    assert "300 NAND gates, 0 wrong" in out


@pytest.mark.gpu
def test_group_c_example_runs(tmp_path):
    """examples/group_nand.c: a device group of two contexts on GPU 0 (the test box has one GPU); the sharded batch
    equals the single-context batch bit for bit"""
    import subprocess
    out = subprocess.run([_build_example(tmp_path, "group_nand"), "5001", "0", "0"], check=True, capture_output=True,
                         text=True, timeout=180).stdout
    assert "5001 NAND gates on 2 member(s), 0 wrong, group == single context: yes" in out
    assert "600 two-level circuits over the group, 0 wrong, replicas equal: yes" in out

"""Builds tests/emu/emu_wave.cpp (the per-lane kernel routines of tfhe_amd/csrc/br_core.h compiled for the host, 64
lanes emulated in a loop) and checks that it reproduces the oracle's transforms bit for bit."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_kernel_arithmetic_on_host(tmp_path):
    exe = str(tmp_path / "emu_wave")
    subprocess.run(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-o", exe,
                    os.path.join(ROOT, "tests", "emu", "emu_wave.cpp"), "-L" + os.path.join(ROOT, "oracle"), "-loracle",
                    "-Wl,-rpath," + os.path.join(ROOT, "oracle")], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert "EMU OK" in out


def test_fast_mode_arithmetic_on_host(tmp_path):
    """tests/emu/emu_fast.cpp: the per-lane routines of tfhe_amd/csrc/fast_core.h (FAST arithmetic mode) against the
    oracle within the tolerances of the reference's own stage tests"""
    exe = str(tmp_path / "emu_fast")
    subprocess.run(["g++", "-std=c++17", "-O2", "-mfma", "-ffp-contract=off", "-o", exe,
                    os.path.join(ROOT, "tests", "emu", "emu_fast.cpp"), "-L" + os.path.join(ROOT, "oracle"), "-loracle",
                    "-Wl,-rpath," + os.path.join(ROOT, "oracle")], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert "EMU FAST OK" in out


def test_generic_dimension_transforms_on_host(tmp_path):
    """tests/emu/emu_nfft.cpp: the per-lane routines of tfhe_amd/csrc/nfft_core.h (N = 512 / 1024 / 2048) against the
    direct evaluation of the transform and the reference's exact products"""
    exe = str(tmp_path / "emu_nfft")
    subprocess.run(["g++", "-std=c++17", "-O2", "-mfma", "-ffp-contract=off", "-o", exe,
                    os.path.join(ROOT, "tests", "emu", "emu_nfft.cpp"), "-L" + os.path.join(ROOT, "oracle"), "-loracle",
                    "-Wl,-rpath," + os.path.join(ROOT, "oracle")], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout
    assert "EMU NFFT OK" in out

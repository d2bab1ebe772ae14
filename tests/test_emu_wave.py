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

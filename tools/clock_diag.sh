#!/bin/bash
# Diagnostic build (never shipped): stamps s_memtime / s_memrealtime around the 630-iteration CMux loop of every wave
# and prints the in-kernel clock (MI355X_MICROARCH.md, DVFS give-back item 6).  Build here, run on the GPU box:
#   bash tools/clock_diag.sh build    &&    gpurun -- 'bash tools/clock_diag.sh run'
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DTFHE_HIP_CLOCK_DIAG \
    -Iinclude -o tfhe_amd/diag_clock.so tfhe_amd/csrc/tfhe_hip.hip tfhe_amd/csrc/compat_libtfhe.cpp tfhe_amd/csrc/wire_io.cpp
else
  TFHE_HIP_LIB=$PWD/tfhe_amd/diag_clock.so timeout -k 10 300 python -u bench.py --steps 12 --warmup 2 --no-cpu-baseline 2>&1 | grep -v "^{" | tail -20
fi

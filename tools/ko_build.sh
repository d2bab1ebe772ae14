#!/bin/bash
# knock-out builds of the blind-rotation kernel (diagnostic; WRONG results, timing only): tfhe_amd/ko_<mask>.so
# mask bits: 1 transposes, 2 twiddle reads, 4 key loads, 8 rotated accumulator reads;  usage: tools/ko_build.sh 1 3 7 15
cd "$(dirname "$0")/.."
for m in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DTFHE_HIP_KO=$m \
    -Iinclude -o tfhe_amd/ko_$m.so tfhe_amd/csrc/tfhe_hip.hip tfhe_amd/csrc/compat_libtfhe.cpp tfhe_amd/csrc/wire_io.cpp 2>/dev/null &
done
wait
ls -la tfhe_amd/ko_*.so

#!/bin/bash
# Host-side sanitizer runs (ASan + UBSan, gcc) of the CPU code that ships or checks: the wire-format reader/writer with
# every truncation of the text part and random header corruption, and client keygen/encrypt + oracle gates + decrypt.
# Both parameter sets.  No GPU code is involved (GPU sanitizers are not available on this pool).
set -e
cd "$(dirname "$0")/../.."
out=$(mktemp -d)
for lam in 128 80; do
  g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -DTFHE_HIP_LAMBDA=$lam -Iinclude \
      -o $out/wire$lam tools/sanitize/wire_drv.cpp tfhe_amd/csrc/wire_io.cpp
  $out/wire$lam
  gcc -std=gnu11 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -ffp-contract=off \
      -DTFHE_HIP_LAMBDA=$lam -DORC_LAMBDA=$lam -Ioracle -o $out/co$lam tools/sanitize/client_oracle_drv.c \
      tfhe_amd/csrc/client_keys.c oracle/tfhe_oracle.c -lm -lpthread
  ASAN_OPTIONS=detect_leaks=0 $out/co$lam
done
rm -rf $out
echo "sanitizer runs clean"

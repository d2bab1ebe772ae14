#!/bin/bash
# A/B: runs bench.py against each library given on the command line (TFHE_HIP_LIB override), 2 rounds interleaved
# usage: bash tools/ab_bench.sh [--arith fast|exact|both] lib1.so lib2.so ...
arith=fast
if [ "$1" = "--arith" ]; then arith=$2; shift 2; fi
for round in 1 2; do
  for lib in "$@"; do
    TFHE_HIP_LIB=$PWD/$lib timeout -k 10 200 python -u bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-inclusive --arith $arith 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']), d['kernel_ms'], round(d['per_gate_latency_ms'],3), (d.get('exact_arithmetic_same_run') or {}).get('kernel_ms'))" || exit 1
  done
done

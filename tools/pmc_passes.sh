#!/bin/bash
# rocprofv3 counter passes of one bench.py run each (one pass per counter group, never combined with a trace domain), plus
# a --kernel-trace --stats pass.  Run on the GPU box:  bash tools/pmc_passes.sh <tag> [bench.py arguments]
# Results: gpurun_out/<tag>_{stats,sq1,sq2,tcc,fetch,write}/ ; fold them with tools/summarise_pmc.py <tag> <round> [mode].
set -u
tag=$1; shift
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
run() {  # name, rocprofv3 options...
    local name=$1; shift
    timeout -s KILL 240 rocprofv3 "$@" --output-format csv -d gpurun_out/${tag}_${name} -- python3 bench.py --no-cpu-baseline --no-host-inclusive --steps 1 --warmup 0 $BENCH_ARGS \
        > gpurun_out/${tag}_${name}.out 2> gpurun_out/${tag}_${name}.err || { echo "pass $name failed"; exit 1; }
}
BENCH_ARGS="$*"
run sq1 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU
run sq2 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL GRBM_GUI_ACTIVE
run tcc --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run fetch --pmc FETCH_SIZE
run write --pmc WRITE_SIZE
BENCH_ARGS="$* --steps 5 --warmup 1"
timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_stats -- python3 bench.py --no-cpu-baseline --no-host-inclusive $BENCH_ARGS \
    > gpurun_out/${tag}_stats.out 2> gpurun_out/${tag}_stats.err || { echo "stats pass failed"; exit 1; }
echo "passes done"

#!/bin/bash
# one measured run of every bench workload beside the headline, each under rocprofv3 --kernel-trace --stats
# (GPU box):  bash tools/side_profiles.sh <tag>   ->  gpurun_out/<tag>_<workload>/ , gpurun_out/<tag>_<workload>.json
set -u
tag=$1
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-.}"
for w in mux fft fft512 fft2048 "fft2048 --fft-kernel fused" comparator adder; do
    timeout -s KILL 280 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_${w// /} -- python3 bench.py --workload $w --steps 2 \
        > gpurun_out/${tag}_${w// /}.json 2> gpurun_out/${tag}_${w// /}.err || { echo "$w failed"; exit 1; }
done
echo "side profiles done"

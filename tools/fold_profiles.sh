#!/bin/bash
# copies the results of tools/pmc_passes.sh r3fast|r3exact, bench.py and tools/side_profiles.sh r3side from gpurun_out/ into
# profiles/<round>_*  (authoring container, after the GPU call):  bash tools/fold_profiles.sh r03
set -e
rnd=${1:-r03}
cd "$(dirname "$0")/.."
python tools/summarise_pmc.py r3fast $rnd fast > /dev/null
python tools/summarise_pmc.py r3exact $rnd exact > /dev/null
for m in fast exact; do
  cp "$(ls -t gpurun_out/r3${m}_stats/*/*_kernel_stats.csv | head -1)" profiles/${rnd}_kernel_stats_$m.csv
  tail -1 gpurun_out/r3${m}_stats.out > profiles/${rnd}_bench_line_${m}_profiled.json
done
tail -1 gpurun_out/r3_bench_final.json > profiles/${rnd}_bench_line_final.json
[ -f gpurun_out/r3_bench_80.json ] && tail -1 gpurun_out/r3_bench_80.json > profiles/${rnd}_bench_line_80bit.json
echo folded

#!/bin/bash
# samples board power and clocks with rocm-smi while bench.py runs (is the blind rotation power-limited?)
cd "$(dirname "$0")/.."
timeout -k 10 280 python -u bench.py --steps 60 --warmup 2 --no-cpu-baseline > gpurun_out/power_bench.log 2>&1 &
BP=$!
n=0
while kill -0 $BP 2>/dev/null && [ $n -lt 400 ]; do
  timeout 10 rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E "Package Power|sclk|Sensor junction" | sed 's/.*: //' | tr '\n' ';' >> gpurun_out/power_samples.log
  echo >> gpurun_out/power_samples.log
  n=$((n+1))
  sleep 0.3
done
wait $BP
tail -1 gpurun_out/power_bench.log | cut -c1-160
grep -v "(9[0-9]Mhz)\|(1[0-9][0-9]Mhz)" gpurun_out/power_samples.log | tail -12

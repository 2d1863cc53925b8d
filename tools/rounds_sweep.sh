#!/bin/bash
# balanced-plan granularity sweep for the fp32 scan (tuning aid)
set -e
mkdir -p gpurun_out; out=gpurun_out/rounds_sweep.txt; : > $out
for r in 1 2 3 4 8; do
  for shp in "50000 40000 512" "5000 40000 512" "100000 100000 512"; do
    LEMON_ROUNDS=$r timeout -k 10 200 python tools/scan_time.py $shp 51 f32 | sed "s/^/rounds=$r /" >> $out
  done
done
cat $out

#!/bin/bash
# re-selection trigger sweep for the fp32 scan (tuning aid)
set -e
mkdir -p gpurun_out; out=gpurun_out/stale_sweep.txt; : > $out
for r in 24 32 48 64 96; do
  for shp in "50000 40000 512" "65536 65536 768"; do
    LEMON_STALE=$r timeout -k 10 200 python tools/scan_time.py $shp 51 f32 | sed "s/^/stale=$r /" >> $out
  done
done
cat $out

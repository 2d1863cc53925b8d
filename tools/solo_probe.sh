#!/bin/bash
# what limits a workgroup that is alone on its CU?  (diagnostic instantiation, results invalid when ablated)
set -e
mkdir -p gpurun_out; out=gpurun_out/solo_probe.txt; : > $out
export LEMON_PHASE_PROF=1 LEMON_SPLITS=1
for a in 0 1 2 3; do
  for shp in "32768 40000 512" "65536 40000 512"; do
    LEMON_ABLATE=$a timeout -k 10 200 python tools/scan_time.py $shp 51 f32 2>&1 | tail -2 | sed "s/^/ablate=$a /" >> $out
  done
done
cat $out

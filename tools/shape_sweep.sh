#!/bin/bash
# fp32 scan across shapes (tuning aid)
set -e
mkdir -p gpurun_out
out=gpurun_out/shape_sweep.txt; : > $out
for shp in "50000 40000 512" "50000 40000 768" "65536 40000 512" "131072 40000 512" "262144 40000 512" "50000 262144 512" "65536 65536 512" "65536 65536 768" "262144 262144 512"; do
  timeout -k 10 200 python tools/scan_time.py $shp 51 f32 >> $out
done
cat $out

#!/bin/bash
# sweep the split count of the fp32 scan on the headline shape (tuning aid)
set -e
mkdir -p gpurun_out
for s in 1 2 3 4 5 6 7 9 12; do
  LEMON_SPLITS=$s timeout -k 10 120 python tools/scan_time.py 50000 40000 512 51 f32 | sed "s/^/splits=$s /" >> gpurun_out/splits_sweep.txt
done
cat gpurun_out/splits_sweep.txt

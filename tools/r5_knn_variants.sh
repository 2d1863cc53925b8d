#!/bin/bash
# same-box A/B of library variants (.variants/liblemon_<v>.so for v in $VARS) on the 1 M x 768 self-join, interleaved
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
K="--workload knn --knn_n 1000000 --knn_d 768 --steps 1 --warmup 0 --no_cpu_baseline"
cp $R/lemon_amd/liblemon_hip.so /tmp/liblemon_tree.so
for r in 1 2; do
  for v in $VARS; do
    cp $R/.variants/liblemon_$v.so $R/lemon_amd/liblemon_hip.so || exit 1
    timeout -k 10 200 python3 $R/bench.py $K > $OUT/r5_knnv_${v}_$r.json 2> $OUT/r5_knnv_${v}_$r.err || exit 1
    python3 - <<PY
import json
d=json.load(open("$OUT/r5_knnv_${v}_$r.json"))
print("$v round $r: %.1f ms  frac %.4f  launches %d avg %.2f ms" % (d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launches"], d["roofline"]["avg_launch_ms"]))
PY
  done
done
cp /tmp/liblemon_tree.so $R/lemon_amd/liblemon_hip.so

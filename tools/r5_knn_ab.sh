#!/bin/bash
# same-box A/B of the 1 M x 768 self-join: QS4 (16x16x32) vs QS2 (32x32x16), alternating
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
K="--workload knn --knn_n 1000000 --knn_d 768 --steps 1 --warmup 0 --no_cpu_baseline"
for r in 1 2; do
  for v in 1 0; do
    LEMON_QS4=$v timeout -k 10 200 python3 $R/bench.py $K > $OUT/r5_knn_qs4_${v}_$r.json 2> $OUT/r5_knn_qs4_${v}_$r.err || exit 1
    python3 - <<PY
import json
d=json.load(open("$OUT/r5_knn_qs4_${v}_$r.json"))
print("QS4=$v round $r: %.1f ms  frac %.4f  launches %d avg %.2f ms" % (d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launches"], d["roofline"]["avg_launch_ms"]))
PY
  done
done

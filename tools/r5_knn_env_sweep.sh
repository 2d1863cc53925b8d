#!/bin/bash
# same-box sweep of the scan's tuning knobs on the 1 M x 768 self-join: each line of $CASES is "label ENV=VAL ..."
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
K="--workload knn --knn_n 1000000 --knn_d 768 --steps 1 --warmup 0 --no_cpu_baseline"
for r in 1 2; do
  while read -r label envs; do
    [ -z "$label" ] && continue
    env $envs timeout -k 10 200 python3 $R/bench.py $K > $OUT/r5_sweep_${label}_$r.json 2> $OUT/r5_sweep_${label}_$r.err || { echo "$label failed"; tail -3 $OUT/r5_sweep_${label}_$r.err; continue; }
    python3 - <<PY
import json
d=json.load(open("$OUT/r5_sweep_${label}_$r.json"))
print("$label round $r: %.1f ms  frac %.4f  launches %d avg %.2f ms" % (d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["launches"], d["roofline"]["avg_launch_ms"]))
PY
  done <<< "$CASES"
done

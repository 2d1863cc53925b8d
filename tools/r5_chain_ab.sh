#!/bin/bash
# same-box A/B of the chain's residual form (LEMON_CHAIN_RES=1: operand-form residual between output projection and fc2, no fp32 tensor
# from the output projection; 0: fp32 residual stream), interleaved headline steps
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
P="--steps 4 --warmup 2 --no_cpu_baseline --no_knn_1m --no_mscoco --no_f32_gemm_check"
for r in 1 2 3; do
  for v in ${MODES:-1 0}; do
    LEMON_CHAIN_RES=$v timeout -k 10 300 python3 $R/bench.py $P > $OUT/r5_chain_res${v}_$r.json 2> $OUT/r5_chain_res${v}_$r.err || exit 1
    python3 - <<PY
import json
d=json.loads(open("$OUT/r5_chain_res${v}_$r.json").read().strip().splitlines()[-1])
print("CHAIN_RES=$v round $r: %.1f scores/s  %.1f ms/step  gemm frac %.4f share %.3f  auroc %s" % (d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("share_of_timed_region", 0), d.get("auroc_check", {}).get("neighbour_terms_only", "")))
PY
  done
done

#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_encoder.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r4_t2.log 2>&1; tail -15 gpurun_out/r4_t2.log
timeout -k 10 600 python bench.py > gpurun_out/r4_bench1.json 2> gpurun_out/r4_bench1.err; tail -3 gpurun_out/r4_bench1.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4_bench1.json"))
print(d["value"], d["ms_per_step"], d["dtype"])
print(json.dumps(d["roofline"])[:900])
print(json.dumps(d["cpu_baseline"])[:2500])
print(d["encoder"]["frac"], d.get("value_f32_gemm_mode"))
print(json.dumps(d["knn_1m"]["roofline"])[:600])
PY

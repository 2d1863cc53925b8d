#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_encoder.py tests/test_gpu_configs.py -m gpu -q -x -k "not tier_b" > $OUT/gputest4.log 2>&1
tail -4 $OUT/gputest4.log
bash tools/perf_bf16.sh > $OUT/perf_bf16.log 2>&1
cat $OUT/perf_bf16.log

#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout -k 10 600 python3 -m pytest tests/test_disc_golden.py tests/test_gpu_loop_golden.py tests/test_gpu_dedup.py "tests/test_gpu_parity.py::test_clip_logits_confidence_matches_reference_golden" -m gpu -q -s > $OUT/gputest3.log 2>&1
tail -5 $OUT/gputest3.log
bash tools/perf_f32.sh > $OUT/perf_f32.log 2>&1
cat $OUT/perf_f32.log
timeout -k 10 900 python3 bench.py > $OUT/bench_call3.json 2> $OUT/bench_call3.err || { tail -20 $OUT/bench_call3.err; exit 3; }
cat $OUT/bench_call3.json

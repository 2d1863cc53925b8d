#!/bin/bash
# GPU call: regenerate the stamped GEMM results file, run the GPU tests, then the default bench.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python3 tools/tune_gemms.py vit-b-32:1000 vit-b-16:256 vit-l-14:128 > $OUT/tune.log 2>&1 || { tail -20 $OUT/tune.log; exit 1; }
cp $OUT/linear_gfx950.csv lemon_amd/data/linear_gfx950.csv
head -3 lemon_amd/data/linear_gfx950.csv
timeout -k 10 1500 python3 -m pytest tests -m gpu -x -q -s > $OUT/gputest.log 2>&1
rc=$?
tail -25 $OUT/gputest.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python3 bench.py > $OUT/bench_call1.json 2> $OUT/bench_call1.err || { tail -20 $OUT/bench_call1.err; exit 3; }
cat $OUT/bench_call1.json

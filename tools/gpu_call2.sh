#!/bin/bash
# GPU call: the whole GPU test suite (all failures listed), then the default bench.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout -k 10 1000 python3 -m pytest tests -m gpu -q --maxfail=10 -s --durations=15 > $OUT/gputest.log 2>&1
rc=$?
grep -n "tier-B\|passed\|failed\|FAILED\|Error" $OUT/gputest.log | tail -30
[ $rc -ne 0 ] && { tail -60 $OUT/gputest.log; exit $rc; }
timeout -k 10 600 python3 bench.py > $OUT/bench_call2.json 2> $OUT/bench_call2.err || { tail -20 $OUT/bench_call2.err; exit 3; }
cat $OUT/bench_call2.json

#!/bin/bash
# Round-end validation on the GPU box (via gpurun): the whole GPU test suite, then tools/gpu_profile.sh (bench + rocprofv3 summaries).
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
timeout -k 10 900 python3 -m pytest tests -m gpu -q --maxfail=10 --durations=8 > $OUT/gputest_full.log 2>&1
rc=$?
tail -14 $OUT/gputest_full.log
[ $rc -ne 0 ] && exit $rc
bash tools/gpu_profile.sh r2 > $OUT/profile_r2.log 2>&1 || { echo "profile step failed: $?"; tail -5 $OUT/profile_r2.log; }
cat $OUT/bench_r2.json

#!/bin/bash
# k_scan_f32: fair-share priority turns of different lengths (LEMON_FAIR = log2 of the turn in 10 ns ticks, 0 = off,
# 100 / 101 = static priority for the even / odd wave slots), regular and diagnostic instantiation
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONPATH=$R
for shape in "50000 40000 512" "131072 131072 512" ${MORE_SHAPES:+"262144 262144 512" "50000 40000 768" "20000 1000000 512" "50000 40000 128"}; do
  for f in ${FAIRS:-0 13 18}; do
    echo "== $shape fair=$f"
    LEMON_FAIR=$f timeout -k 10 300 python3 tools/scan_time.py $shape 51 f32 2>&1 | tail -1
    [ -n "$LIVES" ] && LEMON_FAIR=$f LEMON_PHASE_PROF=1 timeout -k 10 300 python3 tools/scan_time.py $shape 51 f32 2>&1 | grep -E "lives" | tail -1
  done
done

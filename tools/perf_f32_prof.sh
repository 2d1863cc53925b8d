#!/bin/bash
# k_scan_f32 knob sweeps on one box: KNOB=<env name> VALUES="..." (LEMON_FAIR: log2 of the priority turn in 10 ns ticks,
# 0 = off, 100 / 101 = static priority for the even / odd wave slots; LEMON_SEG_COST: tile times a segment is charged
# in the plan; LEMON_XCDS: 1 = plan not XCD-aware)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONPATH=$R
KNOB=${KNOB:-LEMON_FAIR}
for rep in 1 2; do
for shape in ${SHAPES:-"50000 40000 512" "131072 131072 512"}; do
  for f in ${VALUES:-0 13 18}; do
    echo "== $shape $KNOB=$f"
    env $KNOB=$f timeout -k 10 300 python3 tools/scan_time.py $shape 51 f32 2>&1 | tail -1
    [ -n "$LIVES" ] && env $KNOB=$f LEMON_PHASE_PROF=1 timeout -k 10 300 python3 tools/scan_time.py $shape 51 f32 2>&1 | grep -E "lives|phase" | tail -2
  done
done
done

#!/bin/bash
# fabric traffic and L2 behaviour of ONE scan shape under different plans (rocprofv3 --pmc, one group per run):
# usage: tools/pmc_scan.sh "NQ N D" ; env knobs (LEMON_XCDS, LEMON_SPLITS ...) are inherited
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
export PYTHONPATH=$R
SHAPE=${1:-"50000 40000 512"}
cd /tmp && export TMPDIR=/tmp
for cfg in "8" "1"; do
  export LEMON_XCDS=$cfg
  for G in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum"; do
    T=pmcscan_${cfg}_$(echo $G | cut -d' ' -f1)
    rm -rf $OUT/$T
    rocprofv3 --pmc $G --kernel-include-regex "k_scan_f32" --output-format csv -d $OUT/$T -- python3 $R/tools/scan_time.py $SHAPE 51 f32 > /dev/null 2> $OUT/$T.err || echo "$T failed"
  done
done
python3 - <<PY
import csv,glob,collections
for cfg in ("8","1"):
    agg=collections.defaultdict(list)
    for f in glob.glob("$OUT/pmcscan_%s_*/*/*counter_collection.csv" % cfg):
        for r in csv.DictReader(open(f)):
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("LEMON_XCDS=%s" % cfg, {k: "%.4g (x%d launches, mean)" % (sum(v)/len(v), len(v)) for k,v in agg.items()})
PY

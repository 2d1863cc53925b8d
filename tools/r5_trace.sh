#!/bin/bash
# kernel trace of the 1 M x 768 self-join: per-launch durations of the scan / final kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
TAG=${1:-a}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r5_prof_knn_$TAG -- python3 $R/bench.py --workload knn --knn_n 1000000 --knn_d 768 --steps 1 --warmup 0 --no_cpu_baseline $EXTRA > $OUT/r5_prof_knn_$TAG.json 2> $OUT/r5_prof_knn_$TAG.err || exit 1
cd $R
f=$(find $OUT/r5_prof_knn_$TAG -name "*kernel_stats.csv" | head -1); head -6 $f | cut -c1-220
t=$(find $OUT/r5_prof_knn_$TAG -name "*kernel_trace.csv" | head -1)
python3 - $t <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
out=[]
for r in rows:
    n=r["Kernel_Name"]
    if "qs4" in n or "k_bf16_final" in n or "k_scan_bf16_qs" in n or "k_merge" in n:
        short = "qs4" if "qs4" in n else "final" if "final" in n else "merge" if "merge" in n else "qs"
        out.append("%s:%s:%.2f" % (short, r.get("Grid_Size_X", r.get("Grid_Size","?")), (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6))
print(" ".join(out))
PY
find $OUT/r5_prof_knn_$TAG -name "*kernel_trace.csv" -delete

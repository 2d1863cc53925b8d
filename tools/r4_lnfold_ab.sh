#!/bin/bash
# round 4: per-kernel time with and without the LayerNorm fold (one profiled step each)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
P="--steps 1 --warmup 1 --no_cpu_baseline --no_knn_1m --no_f32_gemm_check"
for f in 0 1; do
  export LEMON_LNFOLD=$f
  rm -rf $OUT/prof_lnfold$f
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_lnfold$f -- python3 $R/bench.py $P > $OUT/prof_lnfold$f.json 2> $OUT/prof_lnfold$f.err || exit 1
  f2=$(find $OUT/prof_lnfold$f -name "*kernel_stats.csv" | head -1)
  echo "== LEMON_LNFOLD=$f"; python3 - "$f2" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:9]:
    print(f"  {r['Name'][:100]:100s} {int(r['Calls']):5d} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:8.1f} us {100*float(r['TotalDurationNs'])/tot:5.1f}%")
print("  total", tot/1e6)
PY
done
find $OUT -name "*_kernel_trace.csv" -size +20M -delete

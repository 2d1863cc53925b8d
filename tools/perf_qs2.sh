#!/bin/bash
# bf16 scan A/B on one box (checksums must agree) + per-kernel totals of the chunked 1M x 768 scan (rocprofv3 kernel trace)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONPATH=$R
run() { echo "== $*"; env "$@" timeout -k 10 400 python3 tools/scan_time.py ${SHAPE:-262144 262144 768 51} bf16 2>&1 | grep -v amdgpu.ids | tail -${TAILN:-1}; }
for shape in "1000000 1000000 768 51" "262144 262144 768 51"; do
  SHAPE="$shape" run LEMON_QS2=0
  SHAPE="$shape" run LEMON_QS2=1
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3_trace_1m -- python3 $R/tools/scan_time.py 1000000 1000000 768 51 bf16 > $R/gpurun_out/r3_trace_1m.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
f = sorted(glob.glob('gpurun_out/r3_trace_1m/**/*kernel_trace.csv', recursive=True), key=lambda p: __import__('os').path.getmtime(p))[-1]
rows = [r for r in csv.DictReader(open(f)) if 'bf16' in r['Kernel_Name'] or 'k_merge' in r['Kernel_Name']]
agg = collections.defaultdict(list)
for r in rows:
    agg[r['Kernel_Name'][:64]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
for k_, v in agg.items():
    print(f"{k_:66s} n={len(v):4d} total {sum(v):9.1f} ms  mean {sum(v)/len(v):7.2f}  min {min(v):7.2f} max {max(v):7.2f}")
PY

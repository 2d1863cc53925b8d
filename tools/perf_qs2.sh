#!/bin/bash
# k_scan_bf16_qs2 A/B on one box (checksums must agree): accumulators in VGPRs vs AccVGPRs, refresh cadence, chunk size, the
# one-block kernel, and the per-launch durations of the chunked 1M x 768 scan (rocprofv3 kernel trace)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONPATH=$R
run() { echo "== $*"; env "$@" timeout -k 10 400 python3 tools/scan_time.py ${SHAPE:-262144 262144 768 51} bf16 2>&1 | grep -v amdgpu.ids | tail -${TAILN:-1}; }
for shape in "262144 262144 768 51" "1000000 1000000 768 51"; do
  SHAPE="$shape" run LEMON_QS2=0
  SHAPE="$shape" run LEMON_QS2_ACCV=0
  SHAPE="$shape" run LEMON_QS2_ACCV=1
  SHAPE="$shape" run LEMON_REFRESH=64
  SHAPE="$shape" run LEMON_REFRESH=48
  SHAPE="$shape" run LEMON_REFRESH=32
done
SHAPE="1000000 1000000 768 51" run LEMON_CHUNK_MB=32
SHAPE="1000000 1000000 768 51" run LEMON_CHUNK_MB=128
SHAPE="50000 40000 512 51" run A=0
SHAPE="262144 262144 512 51" run LEMON_QS2_ACCV=0
SHAPE="262144 262144 512 51" run LEMON_QS2_ACCV=1
SHAPE="262144 262144 768 51" run LEMON_QS2_ACCV=0 METRIC=l2
SHAPE="262144 262144 768 51" run LEMON_QS2_ACCV=1 METRIC=l2
SHAPE="262144 262144 768 51" run LEMON_ABLATE=1
SHAPE="1000000 1000000 768 51" TAILN=2 run LEMON_PHASE_PROF=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3_trace_1m -- python3 $R/tools/scan_time.py 1000000 1000000 768 51 bf16 > $R/gpurun_out/r3_trace_1m.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/r3_trace_1m/**/*kernel_trace.csv', recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if 'k_scan_bf16' in r['Kernel_Name']]
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows]
per = 23
print('launches', len(d), 'first search per-chunk ms:', ' '.join(f'{x:.1f}' for x in d[:per]))
print('last search per-chunk ms:', ' '.join(f'{x:.1f}' for x in d[-per:]))
PY

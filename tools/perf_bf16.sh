#!/bin/bash
# k_scan_bf16_qs tuning aid (GPU box): phase stamps + ablations at 262144 x 768 and the 1M x 768 time.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONPATH=$R
run() { echo "== $*"; env "$@" python3 tools/scan_time.py ${SHAPE:-262144 262144 768 51} bf16 2>&1 | grep -v amdgpu.ids | tail -${TAILN:-3}; }
run A=0
TAILN=4 run LEMON_PHASE_PROF=1
run LEMON_ABLATE=1
run LEMON_ABLATE=2
run LEMON_ABLATE=3
SHAPE="1000000 1000000 768 51" run A=0

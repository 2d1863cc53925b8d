#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONPATH=$R
run() { echo "== $*"; env "$@" python3 tools/scan_time.py ${SHAPE:-262144 262144 768 51} bf16 2>&1 | grep -v amdgpu.ids | tail -${TAILN:-1}; }
run A=0
TAILN=3 run LEMON_PHASE_PROF=1
for r in 96 128 160 224; do run LEMON_REFRESH=$r; done
SHAPE="1000000 1000000 768 51" run A=0
SHAPE="1000000 1000000 768 51" run LEMON_REFRESH=128

#!/bin/bash
# instruction-class cost of the fp32 scan's epilogue under the co-resident MFMA stream (diagnostic instantiation)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONPATH=$R
run() { echo "== $*"; env "$@" python3 tools/scan_time.py 50000 40000 512 51 f32 2>&1 | grep -v amdgpu | tail -2; }
run LEMON_PHASE_PROF=1 LEMON_ABLATE=4
run LEMON_PHASE_PROF=1 LEMON_ABLATE=32
run LEMON_PHASE_PROF=1 LEMON_ABLATE=128

#!/bin/bash
# k_scan_f32 tuning aid (GPU box): headline shape + a big shape, with the phase stamps and the diagnostic ablations.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONPATH=$R
run() { echo "== $*"; env "$@" python3 tools/scan_time.py 50000 40000 512 51 f32 2>&1 | tail -2; }
run A=0
run LEMON_PHASE_PROF=1
run LEMON_PHASE_PROF=1 LEMON_ABLATE=4
run LEMON_PHASE_PROF=1 LEMON_ABLATE=1
echo "== big"; python3 tools/scan_time.py 262144 262144 512 51 f32 2>&1 | tail -1
for s in ${STALES:-}; do run LEMON_STALE=$s; done

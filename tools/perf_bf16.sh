#!/bin/bash
# k_scan_bf16_qs tuning aid (GPU box): whole grid per launch against one generation per launch (LEMON_GEN), phase
# stamps, the filter-ablated loop, and the 1M x 768 time
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONPATH=$R
run() { echo "== $*"; env "$@" timeout -k 10 600 python3 tools/scan_time.py ${SHAPE:-262144 262144 768 51} bf16 2>&1 | grep -v amdgpu.ids | tail -${TAILN:-1}; }
for g in ${GENS:-0 256 512}; do
  run LEMON_GEN=$g
  run LEMON_GEN=$g LEMON_ABLATE=1
done
TAILN=2 run LEMON_PHASE_PROF=1
for g in ${GENS:-0 256 512}; do SHAPE="1000000 1000000 768 51" run LEMON_GEN=$g; done

#!/bin/bash
# k_scan_bf16_qs A/B aid (GPU box): the in-tree library, then the variant libraries named in $VARIANTS
# (.variants/liblemon_<X>.so, tools/build_variant.sh): 262 144^2 x 768 whole and filter-ablated, phase stamps, 1M x 768
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONPATH=$R
run() { echo "== $LABEL $*"; env "$@" timeout -k 10 600 python3 tools/scan_time.py ${SHAPE:-262144 262144 768 51} bf16 2>&1 | grep -v amdgpu.ids | tail -${TAILN:-1}; }
one() {
  LABEL=$1
  run A=0
  run LEMON_ABLATE=1
  TAILN=2 run LEMON_PHASE_PROF=1
  SHAPE="1000000 1000000 768 51" run A=0
}
one in-tree
for v in ${VARIANTS:-}; do
  cp .variants/liblemon_$v.so lemon_amd/liblemon_hip.so || exit 1
  one "variant-$v"
done

#!/bin/bash
# k_scan_f32 A/B aid.  The in-tree library first (times + result checksums + phase/clock stamps + the kernel parity
# tests), then any variant libraries .variants/liblemon_<X>.so named in $VARIANTS (tools/build_variant.sh; each is copied
# over the in-tree library on the GPU box, whose copy of the repository is scratch)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONPATH=$R
one() {
  for shape in "50000 40000 512" "262144 262144 512" ${MORE_SHAPES:+"50000 40000 768" "20000 1000000 512" "5000 40000 512" "40000 40000 512" "92783 50000 512"}; do
    echo "== $1: $shape"; timeout -k 10 300 python3 tools/scan_time.py $shape 51 f32 2>&1 | tail -1 || exit 1
  done
  LEMON_PHASE_PROF=1 timeout -k 10 300 python3 tools/scan_time.py 50000 40000 512 51 f32 2>&1 | grep -E "phase|lives" | tail -2
}
if [ -z "$NO_TESTS" ]; then
  timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_dedup.py tests/test_gpu_loop_golden.py tests/test_gpu_configs.py -x -q 2>&1 | tail -3
fi
one in-tree
for x in ${XCDS:-}; do export LEMON_XCDS=$x; one "in-tree LEMON_XCDS=$x"; unset LEMON_XCDS; done
for v in ${VARIANTS:-}; do
  cp .variants/liblemon_$v.so lemon_amd/liblemon_hip.so || exit 1
  one "variant $v"
done

#!/bin/bash
# 16-bit filter scan A/B: the in-tree library first (parity tests + times + result checksums), then the variants named in
# $VARIANTS (.variants/liblemon_<X>.so, tools/build_variant.sh), copied over the in-tree library on the (scratch) GPU box copy
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export PYTHONPATH=$R
one() {
  for shape in "50000 40000 512" "262144 262144 768" "1000000 1000000 768" "262144 262144 512"; do
    echo "== $1: $shape"; timeout -k 10 300 python3 tools/scan_time.py $shape 51 bf16 2>&1 | tail -1 || exit 1
  done
  echo "== $1: l2 262144 262144 768"; METRIC=l2 timeout -k 10 300 python3 tools/scan_time.py 262144 262144 768 51 bf16 2>&1 | tail -1
}
if [ -z "$NO_TESTS" ]; then
  timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_loop_golden.py tests/test_gpu_configs.py -x -q -m gpu 2>&1 | tail -3
fi
one in-tree
for v in ${VARIANTS:-}; do
  cp .variants/liblemon_$v.so lemon_amd/liblemon_hip.so || exit 1
  one "variant $v"
done

#!/bin/bash
# Build .variants/liblemon_<NAME>.so from the kernel sources of git revision REF (default HEAD): the same-box A/B
# partner of the working tree (boxes differ by ~4 % for the fp32 scan and more for the power-limited bf16 scan, so
# timings from two gpurun calls do not compare).  usage: tools/build_variant.sh [REF] [NAME]
set -e
REF=${1:-HEAD}; NAME=${2:-prev}
R=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d /tmp/variant.XXXXXX)
mkdir -p $T/lemon_amd/csrc $T/include $R/.variants
(cd $R && git archive $REF lemon_amd/csrc include | tar -x -C $T)
objs=""
for f in $T/lemon_amd/csrc/*.hip; do
  o=$T/$(basename $f .hip).o
  (cd $T/lemon_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -c $f -o $o) &
  objs="$objs $o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/.variants/liblemon_$NAME.so $objs -lhipblaslt
rm -rf $T
ls -la $R/.variants/liblemon_$NAME.so

#!/bin/bash
# .variants/liblemon_s<SLOTS>.so: the working tree's library with k_scan_f16_qs4 built for another set of issue slots
# (-DLEMON_QS4_SLOTS=abcd: reads behind MFMAs a, a + b, ...; DMA pieces behind MFMAs 6 + c and 6 + d).  usage: tools/r5_build_slots.sh 0126 1257 ...
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/.variants
for v in "$@"; do
  (cd $R/lemon_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-pass-failed -DLEMON_QS4_SLOTS=1$v-10000 -c knn_bf16.hip -o /tmp/knn_bf16_s$v.o) &
done
wait
for v in "$@"; do
  objs=$(ls $R/lemon_amd/csrc/_obj/*.o | grep -v knn_bf16.o)
  hipcc --offload-arch=gfx950 -shared -fPIC -o $R/.variants/liblemon_s$v.so $objs /tmp/knn_bf16_s$v.o -lhipblaslt
done
ls -la $R/.variants/liblemon_s*.so

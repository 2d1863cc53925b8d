#!/bin/bash
# .variants/liblemon_phases.so: the working tree's library with the phase-stamped k_scan_f16_qs4 (-DLEMON_QS4_PHASES)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
(cd $R/lemon_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-pass-failed -DLEMON_QS4_PHASES $1 -c knn_bf16.hip -o /tmp/knn_bf16_ph.o)
objs=$(ls $R/lemon_amd/csrc/_obj/*.o | grep -v knn_bf16.o)
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/.variants/liblemon_phases.so $objs /tmp/knn_bf16_ph.o -lhipblaslt
ls -la $R/.variants/liblemon_phases.so

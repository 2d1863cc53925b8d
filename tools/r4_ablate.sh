#!/bin/bash
# round 4: what pulls the clock down inside the GEMM?  Diagnostic builds that drop one cost at a time (results are wrong by design).
cd "$(dirname "$0")/.."
for a in ${ABL:-0 1 2 3 4 7}; do
  echo "== ablate $a (1: no DMA in the loop, 2: fragments read once, 4: hi x hi product only, 8: no activation DMA = a third less delivery)"
  timeout -k 10 120 .variants/gemm_abl$a 2 5 2>&1 | grep -E "fc2 \(\+res|fc1 \(SiLU" | grep -E "16x16x32|phases" | sed -e 's/max|err.*//' | cut -c1-230
done
echo "== ablate 0, all-zero operands"
GEMM_ZERO=1 timeout -k 10 120 .variants/gemm_abl0 2 5 2>&1 | grep -E "fc2 \(\+res|fc1 \(SiLU" | grep -E "16x16x32|phases" | sed -e 's/max|err.*//' | cut -c1-230

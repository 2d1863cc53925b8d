#!/bin/bash
# round 4: what do the LayerNorm-fold epilogues cost per GEMM (same box, interleaved with the plain kernel)?
cd "$(dirname "$0")/.."
for mode in fold emit; do for rep in 1 2; do
  echo "== $mode (rep $rep)"
  GEMM_LN=$mode timeout -k 10 120 .variants/gemm_ab 3 10 2>&1 | grep -E "mfma 16" | sed -e 's/.*\(fc1 (\|fc2 (\|qkv\|out-proj\|text fc1\|text fc2\)/\1/' | cut -c1-150
done; done

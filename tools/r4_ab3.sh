#!/bin/bash
# round 4: A/B/C of builds of the GEMM micro harness (.variants/gemm_ab_<name> for the names in $VARS), interleaved
cd "$(dirname "$0")/.."
for rep in 1 2 3; do for v in $VARS; do
  echo "== $v (rep $rep)"
  timeout -k 10 120 .variants/gemm_ab_$v 3 10 2>&1 | grep -E "16x16x32" | sed -e 's/.*\(fc1 (\|fc2 (\|qkv\|out-proj\|text fc1\|text fc2\)/\1/' | cut -c1-160
done; done

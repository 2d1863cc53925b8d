#!/bin/bash
# runs the prebuilt GEMM micro-benchmarks (.variants/<name>, built in the container with hipcc) at the ViT-B/32 tower shapes
R=${GRAFT_REPO_ROOT:-$(pwd)}
for b in ${BINS:-gemm_f16x3}; do
  for s in ${SHAPES:-"50000 2304 768" "50000 768 768" "50000 3072 768" "50000 768 3072"}; do
    timeout -k 5 60 $R/.variants/$b $s || echo "$b $s: rc=$?"
  done
done

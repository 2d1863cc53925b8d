#!/bin/bash
# wall-clock split of a full `run_lemon` run on CIFAR-100-shaped synthetic data (diagnostic)
mkdir -p gpurun_out
t0=$(date +%s.%N)
python -m lemon_amd.run_lemon --output_dir /tmp/lemon_e2e --dataset cifar100 --noise_type asymmetric \
  --data_root synthetic:${1:-50000} --clip_path random --knn_k 50 --encoder_batch 1000 > gpurun_out/e2e.log 2> gpurun_out/e2e.err
rc=$?
t1=$(date +%s.%N)
echo "rc=$rc total wall $(python3 -c "print($t1 - $t0)") s"
grep -i "finished" gpurun_out/e2e.log gpurun_out/e2e.err
tail -3 gpurun_out/e2e.err

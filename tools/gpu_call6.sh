#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_trainer.py tests/test_gpu_models.py -q -x -p no:cacheprovider --timeout 300 > gpurun_out/c6_pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/c6_pytest.log
for S in 1 2 4 8; do
timeout 300 python bench.py --streams $S --no-cpu-baseline --no-roofline > gpurun_out/c6_bench_s$S.log 2>&1; echo "bench streams $S rc=$?"; tail -n 1 gpurun_out/c6_bench_s$S.log | cut -c 1-230
done
timeout 300 python bench.py --mode train --steps 20 --warmup 3 > gpurun_out/c6_train.log 2>&1; echo "train rc=$?"; tail -n 1 gpurun_out/c6_train.log | cut -c 1-1200
timeout 300 python bench.py --gpus 1 --mode train --steps 10 --warmup 3 --global-batch 32 > gpurun_out/c6_train32.log 2>&1; echo "train32 rc=$?"; tail -n 1 gpurun_out/c6_train32.log | cut -c 1-400

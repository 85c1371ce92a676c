#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_trainer.py -q -x -p no:cacheprovider --timeout 300 > gpurun_out/c7_pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/c7_pytest.log
for G in 0 1; do for S in 1 2 3 4 6; do
timeout 300 python bench.py --streams $S --graph $G --no-cpu-baseline --no-roofline > gpurun_out/c7_bench_g${G}_s$S.log 2>&1; echo "bench graph $G streams $S rc=$? $(tail -n 1 gpurun_out/c7_bench_g${G}_s$S.log | cut -c 64-90,190-220)"
done; done

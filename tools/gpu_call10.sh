#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py -q -x -p no:cacheprovider --timeout 600 -s -k "campplus_training" > gpurun_out/c10_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "\[|passed|failed|Error|assert" gpurun_out/c10_pytest.log | head -10
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train -o tr -- python bench.py --mode train --steps 5 --warmup 2 > gpurun_out/c10_prof.log 2>&1
echo "rocprof rc=$?"
find gpurun_out/prof_train -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} gpurun_out/kernel_stats_train_r2.csv
find gpurun_out/prof_train -name "*kernel_trace.csv" -delete
head -n 40 gpurun_out/kernel_stats_train_r2.csv | cut -c 1-170

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py -q -x -p no:cacheprovider --timeout 600 -s -k "ecapa_training_step_mixed" > gpurun_out/c12_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "\[amp|\[ecapa|passed|failed|Error|assert" gpurun_out/c12_pytest.log | head -40
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_train2 -o tr -- python bench.py --mode train --steps 5 --warmup 2 > gpurun_out/c12_prof.log 2>&1
find gpurun_out/prof_train2 -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} gpurun_out/kernel_stats_train_amp_r2.csv
find gpurun_out/prof_train2 -name "*kernel_trace.csv" -delete
head -n 26 gpurun_out/kernel_stats_train_amp_r2.csv | cut -c 1-150

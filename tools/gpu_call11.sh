#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train.py -q -x -p no:cacheprovider --timeout 600 -s -k "mixed_precision" > gpurun_out/c11_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "\[amp|\[ecapa|passed|failed|Error|assert" gpurun_out/c11_pytest.log | head -40
for A in 0 1; do
timeout 300 python bench.py --mode train --steps 10 --warmup 3 --amp $A > gpurun_out/c11_train_amp$A.log 2>&1; echo "train amp $A rc=$?"; tail -n 1 gpurun_out/c11_train_amp$A.log | cut -c 1-260
done

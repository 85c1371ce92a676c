#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py -q -x -p no:cacheprovider --timeout 600 -s -k "ecapa_training_step_mixed" > gpurun_out/c13_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "\[amp|\[ecapa|passed|failed|Error|assert" gpurun_out/c13_pytest.log | head -40

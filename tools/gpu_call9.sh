#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_models.py tests/test_gpu_train.py -q -x -p no:cacheprovider --timeout 600 -s -k "large or named or melspectrogram_specaugment or campplus_training" > gpurun_out/c9_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "\[|passed|failed|Error|assert" gpurun_out/c9_pytest.log | head -40

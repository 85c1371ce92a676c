#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -q -x --tb=short -p no:cacheprovider --timeout 300 2>&1 | tail -n 4
VP_BF16_ONLY=1 timeout 120 python tools/model_probe.py 64 ResNetSE ERes2Net 2>&1 | grep -E "ResNetSE|ERes2Net"
VP_BF16_ONLY=1 timeout 120 python tools/model_probe.py 256 EcapaTdnn TDNN 2>&1 | grep -E "Ecapa|TDNN"

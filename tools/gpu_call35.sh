#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -q -x --tb=short -p no:cacheprovider --timeout 300 -k "c32 or campp or cam_ or resnet" 2>&1 | tail -n 4
VP_BF16_ONLY=1 python tools/model_probe.py 256 CAMPPlus 2>/dev/null | grep CAMP
VP_BF16_ONLY=1 python tools/model_probe.py 64 CAMPPlus ResNetSE 2>/dev/null | grep "B="

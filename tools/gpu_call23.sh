#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -x --tb=short -p no:cacheprovider --timeout 300 -k "cam or CAM" 2>&1 | tail -n 12
for B in 256 64; do
VP_BF16_ONLY=1 timeout 120 python tools/model_probe.py $B CAMPPlus 2>&1 | grep CAMP
VPMI_CAM_UNFUSED=1 VP_BF16_ONLY=1 timeout 120 python tools/model_probe.py $B CAMPPlus 2>&1 | grep CAMP
done

VPMI_FCM_GENERAL=1 VP_BF16_ONLY=1 timeout 120 python tools/model_probe.py 256 CAMPPlus 2>&1 | grep CAMP

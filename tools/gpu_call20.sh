#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -q -x --tb=short -p no:cacheprovider --timeout 200 -k "res2 or ecapa or Ecapa or score" 2>&1 | tail -n 4
VPMI_LIB=voiceprintrecognition-paddlepaddle_amd/lib/libvpmi_timing.so python tools/res2_timing.py 2>&1 | tail -n 1
VPMI_RES2_DOUBLE=1 VPMI_LIB=voiceprintrecognition-paddlepaddle_amd/lib/libvpmi_timing.so python tools/res2_timing.py 2>&1 | tail -n 1
timeout 200 python bench.py --steps 200 --no-cpu-baseline --no-train-line 2>/dev/null | tail -n 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'])"

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
PKG=voiceprintrecognition-paddlepaddle_amd
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "wide_tiles" -p no:cacheprovider --timeout 200 > gpurun_out/c3_pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/c3_pytest.log
timeout 400 python tools/gemm_probe.py 5 3,4,5,6,7,8,9 > gpurun_out/c3_gemm.log 2>&1; echo "gemm rc=$?"; cat gpurun_out/c3_gemm.log
VPMI_CONV256=4 VPMI_LIB=$PWD/$PKG/lib/libvpmi_timing.so timeout 200 python tools/tile_timing.py > gpurun_out/c3_tiles_s4.log 2>&1; cat gpurun_out/c3_tiles_s4.log
VPMI_CONV256=5 VPMI_LIB=$PWD/$PKG/lib/libvpmi_timing.so timeout 200 python tools/tile_timing.py > gpurun_out/c3_tiles_s5.log 2>&1; cat gpurun_out/c3_tiles_s5.log

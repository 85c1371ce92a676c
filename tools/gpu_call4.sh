#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
PKG=voiceprintrecognition-paddlepaddle_amd
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv1d" -p no:cacheprovider --timeout 200 > gpurun_out/c4_pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/c4_pytest.log
timeout 400 python tools/gemm_probe.py 5 3,8,9 > gpurun_out/c4_gemm.log 2>&1; echo "gemm rc=$?"; cat gpurun_out/c4_gemm.log
PSUM=1 timeout 400 python tools/gemm_probe.py 5 3,8,9 > gpurun_out/c4_gemm_psum.log 2>&1; echo "gemm psum rc=$?"; cat gpurun_out/c4_gemm_psum.log
VPMI_CONV256=9 VPMI_LIB=$PWD/$PKG/lib/libvpmi_timing.so timeout 200 python tools/tile_timing.py > gpurun_out/c4_tiles_s9.log 2>&1; cat gpurun_out/c4_tiles_s9.log
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/c4_bench.log 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/c4_bench.log | cut -c 1-300
VPMI_CONV256=9 timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-roofline > gpurun_out/c4_bench9.log 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/c4_bench9.log | cut -c 1-300

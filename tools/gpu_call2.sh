#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
PKG=voiceprintrecognition-paddlepaddle_amd
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "wide_tiles" -p no:cacheprovider --timeout 200 > gpurun_out/c2_pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/c2_pytest.log
timeout 300 python tools/gemm_probe.py 5 3,4,5 > gpurun_out/c2_gemm.log 2>&1; echo "gemm rc=$?"; cat gpurun_out/c2_gemm.log
VPMI_LIB=$PWD/$PKG/lib/libvpmi_timing.so timeout 200 python tools/tile_timing.py > gpurun_out/c2_tiles_s5.log 2>&1; cat gpurun_out/c2_tiles_s5.log
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/c2_bench.log 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/c2_bench.log | cut -c 1-400

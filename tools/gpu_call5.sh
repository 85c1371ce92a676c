#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
PKG=voiceprintrecognition-paddlepaddle_amd
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -k "conv1d" -p no:cacheprovider --timeout 200 > gpurun_out/c5_pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 gpurun_out/c5_pytest.log
PSUM=1 timeout 400 python tools/gemm_probe.py 5 3,4,5 > gpurun_out/c5_gemm_psum.log 2>&1; echo "gemm psum rc=$?"; cat gpurun_out/c5_gemm_psum.log
timeout 400 python tools/gemm_probe.py 5 3,4,5 > gpurun_out/c5_gemm.log 2>&1; echo "gemm rc=$?"; cat gpurun_out/c5_gemm.log
timeout 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/c5_bench.log 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/c5_bench.log | cut -c 1-300
timeout 600 python -m pytest tests/test_gpu_models.py -q -x -p no:cacheprovider --timeout 200 > gpurun_out/c5_pytest_models.log 2>&1; echo "pytest models rc=$?"; tail -n 3 gpurun_out/c5_pytest_models.log

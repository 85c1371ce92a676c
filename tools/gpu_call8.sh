#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -q -x -p no:cacheprovider --timeout 400 -s -k "res2_chain or asp_fused or se_gate or bench_config" > gpurun_out/c8_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "^\[|passed|failed|Error|assert" gpurun_out/c8_pytest.log | head -60
timeout 300 python bench.py --steps 100 --no-cpu-baseline --no-roofline > gpurun_out/c8_bench.log 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/c8_bench.log | cut -c 1-2500

#!/bin/bash
# One GPU-box session: parity tests (per-test timeout so a hang cannot eat the budget), smoke, bench.
# Usage (from the repo root on the GPU box): bash tools/gpu_round.sh <tag>
TAG=${1:-r}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 700 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s --timeout 150 --timeout-method thread \
    > gpurun_out/pytest_$TAG.log 2>&1
echo "pytest rc=$?"
grep -E "passed|failed|error|Timeout" gpurun_out/pytest_$TAG.log | tail -n 5
grep -E "^(FAILED|ERROR)|rel-L2|score err|1-cos" gpurun_out/pytest_$TAG.log | head -n 60
timeout 240 python __graft_entry__.py smoke > gpurun_out/smoke_$TAG.log 2>&1
echo "smoke rc=$?"
tail -n 4 gpurun_out/smoke_$TAG.log
timeout 400 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_$TAG.log 2>&1
echo "bench rc=$?"
tail -n 3 gpurun_out/bench_$TAG.log | cut -c 1-3000
if [ "${PROF:-1}" = "1" ]; then
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/prof_$TAG.log 2>&1
  echo "rocprof rc=$?"
  find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} gpurun_out/kernel_stats_$TAG.csv
  find gpurun_out/prof_$TAG -name "*kernel_trace.csv" -delete
  head -n 30 gpurun_out/kernel_stats_$TAG.csv | cut -c 1-200
fi

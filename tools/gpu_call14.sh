#!/bin/bash
# Session baseline of HEAD: whole GPU suite, smoke, bench (+ rocprof of the same command), train rocprof, model probe.
mkdir -p gpurun_out; export TMPDIR=/tmp
T=${1:-r02b}
timeout 800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 200 --timeout-method thread > gpurun_out/pytest_$T.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|error|Timeout" gpurun_out/pytest_$T.log | tail -n 5; grep -E "^(FAILED|ERROR)" gpurun_out/pytest_$T.log | head -20
timeout 240 python __graft_entry__.py smoke > gpurun_out/smoke_$T.log 2>&1; echo "smoke rc=$?"; tail -n 3 gpurun_out/smoke_$T.log
timeout 400 python bench.py > gpurun_out/bench_$T.log 2>&1; echo "bench rc=$?"; tail -n 2 gpurun_out/bench_$T.log | cut -c 1-4000
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$T -o bench -- python bench.py --steps 20 --warmup 3 --graph 0 --streams 1 --no-cpu-baseline --no-roofline --no-train-line > gpurun_out/prof_$T.log 2>&1
echo "rocprof rc=$?"
find gpurun_out/prof_$T -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} gpurun_out/kernel_stats_$T.csv
find gpurun_out/prof_$T -name "*kernel_trace.csv" -delete
head -n 32 gpurun_out/kernel_stats_$T.csv | cut -c 1-200
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/proft_$T -o tr -- python bench.py --mode train --steps 5 --warmup 2 > gpurun_out/proft_$T.log 2>&1
find gpurun_out/proft_$T -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} gpurun_out/kernel_stats_train_$T.csv
find gpurun_out/proft_$T -name "*kernel_trace.csv" -delete
head -n 30 gpurun_out/kernel_stats_train_$T.csv | cut -c 1-170
tail -n 1 gpurun_out/proft_$T.log | cut -c 1-1500
VP_BF16_ONLY=1 timeout 300 python tools/model_probe.py 256 > gpurun_out/model_probe_$T.log 2>&1; cat gpurun_out/model_probe_$T.log

#!/bin/bash
# One GPU-box session, parameterised.  Usage (repo root on the GPU box): bash tools/gpu_session.sh <tag> <step> [<step> ...]
#   tests[:<pytest -k expr>]  parity suite with -s (measured values kept), per-test timeout; log -> gpurun_out/pytest_<tag>.log
#   smoke                     __graft_entry__.smoke()
#   bench                     python bench.py (driver defaults, fewer steps)            -> gpurun_out/bench_<tag>.log
#   prof                      rocprofv3 kernel stats of the infer step alone (no train line, one stream, no graph)
#   proftrain[:<batch>]       rocprofv3 kernel stats of the training step
#   gemm[:<scheds>]           tools/gemm_probe.py A/B of the conv256 schedules on the bench shapes (with and without fused sums)
#   run:<command>             anything else, output -> gpurun_out/run_<tag>.log
TAG=${1:-r}; shift
mkdir -p gpurun_out
export TMPDIR=/tmp
for STEP in "$@"; do
  case "$STEP" in
    tests*)
      K="${STEP#tests}"; K="${K#:}"
      if [ -n "$K" ]; then
        timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s --timeout 200 --timeout-method thread -k "$K" > gpurun_out/pytest_$TAG.log 2>&1
      else
        timeout 1100 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s --timeout 200 --timeout-method thread > gpurun_out/pytest_$TAG.log 2>&1
      fi
      echo "pytest rc=$?"; grep -E "passed|failed|error|Timeout" gpurun_out/pytest_$TAG.log | tail -n 5
      grep -E "^(FAILED|ERROR)" gpurun_out/pytest_$TAG.log | head -n 40 ;;
    smoke)
      timeout 240 python __graft_entry__.py smoke > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?"; tail -n 4 gpurun_out/smoke_$TAG.log ;;
    bench)
      timeout 500 python bench.py --steps 100 --warmup 10 > gpurun_out/bench_$TAG.log 2>&1; echo "bench rc=$?"; tail -n 2 gpurun_out/bench_$TAG.log | cut -c 1-3500 ;;
    prof)
      (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --graph 0 --streams 1 --no-cpu-baseline --no-roofline --no-train-line > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1)
      echo "rocprof rc=$?"; find /tmp/prof_$TAG -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} gpurun_out/kernel_stats_$TAG.csv
      head -n 32 gpurun_out/kernel_stats_$TAG.csv | cut -c 1-160 ;;
    proftrain*)
      GB="${STEP#proftrain}"; GB="${GB#:}"; GB=${GB:-256}
      (cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/proft_$TAG -o tr -- python $GRAFT_REPO_ROOT/bench.py --mode train --global-batch $GB --steps 10 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/proftrain_${TAG}_b$GB.log 2>&1)
      echo "rocprof train rc=$?"; find /tmp/proft_$TAG -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} gpurun_out/kernel_stats_train_${TAG}_b$GB.csv
      grep -h '^{"metric"' gpurun_out/proftrain_${TAG}_b$GB.log | cut -c1-500; head -n 24 gpurun_out/kernel_stats_train_${TAG}_b$GB.csv | cut -c 1-150 ;;
    gemm*)
      S="${STEP#gemm}"; S="${S#:}"; S=${S:-4,5,6}
      timeout 300 python tools/gemm_probe.py 5 $S 1536x1536 512x512 > gpurun_out/gemm_$TAG.log 2>&1
      PSUM=1 timeout 300 python tools/gemm_probe.py 5 $S 1536x1536 512x512 >> gpurun_out/gemm_$TAG.log 2>&1
      echo "gemm rc=$?"; cat gpurun_out/gemm_$TAG.log ;;
    run:*)
      CMD="${STEP#run:}"; timeout 600 bash -c "$CMD" > gpurun_out/run_$TAG.log 2>&1; echo "run rc=$?"; tail -n 30 gpurun_out/run_$TAG.log | cut -c 1-400 ;;
  esac
done

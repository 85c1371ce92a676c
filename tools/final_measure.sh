#!/bin/bash
# The round's measurement package on ONE box, PMC pass last (it is the slowest and nothing after it depends on it):
#   bash tools/final_measure.sh <tag>     -> gpurun_out/{bench,kernel_stats_infer,kernel_stats_train,pmc,smoke}_<tag>.*
TAG=${1:-r06}
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_$TAG.log 2>&1; echo "bench rc=$?"; tail -n 1 gpurun_out/bench_$TAG.log | cut -c1-400
timeout 240 python __graft_entry__.py smoke > gpurun_out/smoke_$TAG.log 2>&1; echo "smoke rc=$?"; tail -n 2 gpurun_out/smoke_$TAG.log
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fm_i -o b -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/prof_bench_$TAG.log 2>&1)
find /tmp/fm_i -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} gpurun_out/kernel_stats_bench_$TAG.csv
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fm_s -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --graph 0 --streams 1 --no-cpu-baseline --no-roofline --no-train-line > $GRAFT_REPO_ROOT/gpurun_out/prof_infer_$TAG.log 2>&1)
find /tmp/fm_s -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} gpurun_out/kernel_stats_infer_$TAG.csv
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fm_t -o b -- python3 $GRAFT_REPO_ROOT/bench.py --mode train --global-batch 256 --steps 10 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/prof_train_$TAG.log 2>&1)
find /tmp/fm_t -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} gpurun_out/kernel_stats_train_$TAG.csv
# the split-precision (parity) engine's step: kernel stats + ordered launch list
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/fm_x -o b -- python3 $GRAFT_REPO_ROOT/bench.py --dtype float32x3 --steps 20 --warmup 3 --graph 0 --streams 1 --no-cpu-baseline --no-roofline --no-train-line > $GRAFT_REPO_ROOT/gpurun_out/prof_infer_x3_$TAG.log 2>&1)
find /tmp/fm_x -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} gpurun_out/kernel_stats_infer_x3_$TAG.csv
bash tools/infer_seq.sh 256 float32x3 > gpurun_out/infer_step_sequence_x3_$TAG.log 2>&1
bash tools/infer_seq.sh 256 bfloat16 > gpurun_out/infer_step_sequence_$TAG.log 2>&1
echo "stats done"; head -n 4 gpurun_out/kernel_stats_infer_$TAG.csv | cut -c1-150; head -n 4 gpurun_out/kernel_stats_infer_x3_$TAG.csv | cut -c1-150
bash tools/pmc_step.sh infer_$TAG python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --graph 0 --streams 1 --no-cpu-baseline --no-roofline --no-train-line 2>&1 | tail -n 14
bash tools/pmc_step.sh infer_x3_$TAG python3 $GRAFT_REPO_ROOT/bench.py --dtype float32x3 --steps 6 --warmup 2 --graph 0 --streams 1 --no-cpu-baseline --no-roofline --no-train-line 2>&1 | tail -n 14

#!/bin/bash
# rocprof kernel stats of the training step at global batch 32 (the per-GPU share of the strong-scaled 8-GPU point)
export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/proft32 -o tr -- python $GRAFT_REPO_ROOT/bench.py --mode train --global-batch 32 --steps 10 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/proft_b32.log 2>&1
cp $(find /tmp/proft32 -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_train_b32.csv
grep -h '^{"metric"' $GRAFT_REPO_ROOT/gpurun_out/proft_b32.log | cut -c1-400
head -14 $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_train_b32.csv | cut -c1-150

#!/bin/bash
# per-kernel profile of one 2-D backbone at a batch: gpu_call30.sh <Model> <B>
mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
M=$1; B=$2
VP_BF16_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm -o m -- python $GRAFT_REPO_ROOT/tools/model_probe.py $B $M 2>&1 | grep "$M"
f=$(find /tmp/pm -name "*kernel_stats.csv" | head -n 1); cp $f $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_${M}_B$B.csv; head -n 30 $f | cut -c 1-200

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
for B in 256 64; do
VP_BF16_ONLY=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc$B -o cam -- python $GRAFT_REPO_ROOT/tools/model_probe.py $B CAMPPlus 2>&1 | grep CAMP
f=$(find /tmp/pc$B -name "*kernel_stats.csv" | head -n 1); cp $f $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_campp_B$B.csv; head -n 14 $f | cut -c 1-180
done

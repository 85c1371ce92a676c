#!/bin/bash
# training-path checkpoint: the train tests, then the DP train bench line (graphed) + its kernel stats
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_train.py -m gpu -q -x --tb=short -p no:cacheprovider --timeout 300 2>&1 | tail -n 12
python bench.py --mode train --steps 20 --warmup 5 2>/dev/null | tail -n 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('train', d['ms_per_step'], d['value'], d['loss'], d['config'].get('hip_graph'), d.get('hip_graph_error'))"
python bench.py --mode train --global-batch 32 --steps 20 --warmup 5 2>/dev/null | tail -n 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('train B=32', d['ms_per_step'], d['value'], d['config'].get('hip_graph'), d.get('hip_graph_error'))"
mkdir -p gpurun_out
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/proft -o tr -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 10 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/proft_$1.log 2>&1
cp $(find /tmp/proft -name '*kernel_stats.csv' | head -1) $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_train_$1.csv
head -12 $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_train_$1.csv | cut -c1-150

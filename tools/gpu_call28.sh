#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -q -x --tb=short -p no:cacheprovider --timeout 300 -s -k "graphed" 2>&1 | tail -n 12
for gb in 256 32; do for tg in 1 0; do
echo "global-batch $gb train-graph $tg"; python bench.py --mode train --global-batch $gb --train-graph $tg --steps 20 --warmup 5 2>/dev/null | tail -n 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['loss'], d['config'].get('hip_graph'), d.get('hip_graph_error'))"
done; done

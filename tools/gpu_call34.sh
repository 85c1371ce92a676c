#!/bin/bash
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -q -x --tb=short -p no:cacheprovider --timeout 300 -s -k "asp or ecapa or tdnn_training or resnetse" 2>&1 | grep -v "^$" | tail -n 16
for rep in 1 2; do
python bench.py --mode train --steps 20 --warmup 5 2>/dev/null | tail -n 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('train', d['ms_per_step'], d['value'], d['loss'], d['config'].get('hip_graph'), d.get('hip_graph_error'))"
done

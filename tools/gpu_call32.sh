#!/bin/bash
# A/B of VPMI_TRAIN_BF16_OPS levels on the training bench + the wide tests
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_train.py -m gpu -q -x --tb=short -p no:cacheprovider --timeout 300 -s -k "wide" 2>&1 | grep -v "^$" | tail -n 40
for lv in 1 2 1 2; do
echo -n "level $lv: "; VPMI_TRAIN_BF16_OPS=$lv python bench.py --mode train --steps 20 --warmup 5 2>/dev/null | tail -n 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('train', d['ms_per_step'], d['value'], d['loss'], d['config'].get('hip_graph'), d.get('hip_graph_error'))"
done

#!/bin/bash
# A/B of an env switch on the headline bench inside one box: gpu_call31.sh VAR
export TMPDIR=/tmp
V=$1
for rep in 1 2; do for val in 0 1; do
echo -n "$V=$val: "; env $V=$val python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-train-line 2>/dev/null | tail -n 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['loss'], d['roofline']['avg_launch_ms'] if 'roofline' in d else None)"
done; done

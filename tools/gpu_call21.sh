#!/bin/bash
export TMPDIR=/tmp
for g in 0 1 2 3 8 16 32; do
  echo "GROUP_M=$g"; VPMI_GROUP_M=$g timeout 200 python bench.py --steps 50 --no-cpu-baseline --no-train-line 2>/dev/null | tail -n 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], [ (l['cin'],l['ms']) for l in d['roofline']['launches']])"
done

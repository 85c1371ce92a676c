#!/bin/bash
# streams / schedule sweep of the bench step
mkdir -p gpurun_out; export TMPDIR=/tmp
for s in 1 2 3 4; do
  echo "streams=$s graph=1"; timeout 200 python bench.py --streams $s --steps 200 --no-cpu-baseline --no-roofline --no-train-line 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done
echo "streams=1 graph=0"; timeout 200 python bench.py --streams 1 --graph 0 --steps 200 --no-cpu-baseline --no-roofline --no-train-line 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
for c in 3 5; do
  echo "VPMI_CONV256=$c streams=2"; VPMI_CONV256=$c timeout 200 python bench.py --streams 2 --steps 200 --no-cpu-baseline --no-roofline --no-train-line 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])"
done

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --timeout 200 --timeout-method thread -x 2>&1 | tail -n 4
timeout 200 python bench.py --steps 200 --no-cpu-baseline --no-train-line 2>/dev/null | tail -n 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'])"

#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -q -x --tb=short -p no:cacheprovider --timeout 200 2>&1 | tail -n 3
for c in 4 3; do
  echo "VPMI_CONV256=$c streams=2"; VPMI_CONV256=$c timeout 200 python bench.py --streams 2 --steps 200 --no-cpu-baseline --no-train-line 2>/dev/null | tail -n 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['roofline']['frac'], [ (l['cin'],l['ms']) for l in d['roofline']['launches']])"
done

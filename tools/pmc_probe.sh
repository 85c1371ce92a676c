#!/bin/bash
# PMC passes over tools/gemm_probe.py (one counter group per pass; --kernel-trace only, as gpurun requires).
# Usage (GPU box): bash tools/pmc_probe.sh <tag> <shapes...>   -> gpurun_out/pmc_<tag>_<n>/ + a summary on stdout
TAG=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pmc_${TAG}_$i -o g -- python tools/gemm_probe.py 3 "$@" > gpurun_out/pmc_${TAG}_$i.log 2>&1
  python - "$i" "$grp" gpurun_out/pmc_${TAG}_$i <<'PY'
import csv, glob, sys, collections
i, grp, d = sys.argv[1], sys.argv[2], sys.argv[3]
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
if not f:
    print('pass', i, 'no counter file'); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name']
    if 'conv_gemm' not in k: continue
    acc[(k[:60], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for (k, gs), c in acc.items():
    print('pass', i, k, 'grid', gs, {n: round(sum(v) / len(v)) for n, v in c.items()})
PY
done

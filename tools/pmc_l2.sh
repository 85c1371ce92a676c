#!/bin/bash
# L1 <-> L2 request counters of the 256-wide GEMM, one schedule at a time (is the 64-B-row ring fetching every line twice?)
export TMPDIR=/tmp; mkdir -p gpurun_out
for S in "$@"; do
 i=0
 for grp in "TCP_TCC_READ_REQ_sum TCC_REQ_sum" "TCC_READ_sum TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum" "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d gpurun_out/pl2_${S}_$i -o g -- python tools/gemm_probe.py 2 $S 1536x1536 512x512 > gpurun_out/pl2_${S}_$i.log 2>&1
  python - "$S" "$grp" gpurun_out/pl2_${S}_$i <<'PY'
import csv, glob, sys, collections
s, grp, d = sys.argv[1], sys.argv[2], sys.argv[3]
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
if not f:
    print('sched', s, grp, 'no counter file'); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    k = r['Kernel_Name']
    if 'conv_gemm' not in k: continue
    acc[(k[:50], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for (k, gs), c in acc.items():
    print('sched', s, k, 'grid', gs, {n: round(sum(v) / len(v)) for n, v in c.items()}, 'n', [len(v) for v in c.values()][0])
PY
  tail -n 2 gpurun_out/pl2_${S}_$i.log | grep -i "error\|invalid" | head -2
 done
done

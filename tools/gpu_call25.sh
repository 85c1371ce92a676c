#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
M=${1:-ResNetSE}; B=${2:-64}
VP_BF16_ONLY=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/pr -o m -- python $GRAFT_REPO_ROOT/tools/model_probe.py $B $M 2>&1 | grep "$M"
f=$(find /tmp/pr -name "*kernel_trace.csv" | head -n 1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last forward = kernels after the last 'fcm_conv1' (stem) launch
idx = max(i for i, r in enumerate(rows) if 'fcm_conv1' in r['Kernel_Name'])
agg = collections.OrderedDict(); tot = 0
for r in rows[idx:]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    n = r['Kernel_Name'][:90]
    a = agg.setdefault(n, [0, 0.0]); a[0] += 1; a[1] += d; tot += d
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{d:9.1f} us {c:4d} calls  {n}")
print('sum', tot)
# the 12 longest single launches
big = sorted(rows[idx:], key=lambda r: int(r['Start_Timestamp']) - int(r['End_Timestamp']))[:14]
for r in big:
    print(f"   {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us grid {r.get('Grid_Size_X','?')} {r['Kernel_Name'][:70]}")
PY

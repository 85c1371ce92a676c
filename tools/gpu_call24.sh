#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
cd /tmp
VP_BF16_ONLY=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/pc -o cam -- python $GRAFT_REPO_ROOT/tools/model_probe.py 256 CAMPPlus > /dev/null 2>&1
f=$(find /tmp/pc -name "*kernel_trace.csv" | head -n 1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last forward: find last fcm_conv1 launch
idx = max(i for i, r in enumerate(rows) if 'fcm_conv1' in r['Kernel_Name'] or 'Lb1EEE' in r['Kernel_Name'] or '<2, true>' in r['Kernel_Name'])
tot = 0
for r in rows[idx:]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    n = r['Kernel_Name']
    n = n[:70]
    print(f"{d:8.1f} us  grid {r.get('Grid_Size_X','?'):>8} {r.get('Grid_Size_Y','?'):>5}  {n}")
print('sum', tot)
PY

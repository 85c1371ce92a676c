# ordered kernel list of ONE inference step (the last of the run, one stream, no graph) with durations: bash tools/infer_seq.sh [batch] [dtype]
export TMPDIR=/tmp
GB=${1:-256}
DT=${2:-bfloat16}
(cd /tmp && timeout 500 rocprofv3 --kernel-trace --output-format csv -d /tmp/iseq -o tr -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --dtype $DT --graph 0 --streams 1 --no-cpu-baseline --no-roofline --no-train-line > /tmp/iseq.log 2>&1)
f=$(find /tmp/iseq -name "*kernel_trace.csv" | head -n 1)
python3 - "$f" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# the last step: from the last fbank kernel on
start = max(i for i, n in enumerate(names) if 'fbank_frames' in n)
t0 = int(rows[start]['Start_Timestamp'])
tot = 0
for r in rows[start:]:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    tot += d
    n = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])
    n = re.sub(r'^void ', '', n)
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {d:8.1f}  g={r.get('Grid_Size_X', '?'):>8} {n[:130]}")
print('kernel time of the step', tot, 'us;', len(rows) - start, 'launches')
PY

# SQ instruction / wait counters of asp_utt_kernel under a VPMI_ASP_DBG setting.  Usage (GPU box): bash tools/pmc_asp.sh "<dbg values>"
export TMPDIR=/tmp
for d in $1; do
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_IFETCH SQ_WAVES SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS SQ_BUSY_CYCLES"; do
    (cd /tmp && SHAPES=256x298 VPMI_ASP_DBG=$d timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pq_$d -o g -- python $GRAFT_REPO_ROOT/tools/asp_probe.py dump > /tmp/pq_$d.log 2>&1)
    f=$(find /tmp/pq_$d -name "*counter_collection.csv" | head -1)
    python - "$f" "$d" <<'PY'
import csv, sys, collections
c = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'asp_utt' in r['Kernel_Name']:
        c[r['Counter_Name']].append(float(r['Counter_Value']))
print('dbg', sys.argv[2], {k: round(sum(v) / len(v)) for k, v in c.items()})
PY
    rm -rf /tmp/pq_$d
  done
done

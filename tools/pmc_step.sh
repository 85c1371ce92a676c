#!/bin/bash
# HBM / L2 / MFMA / LDS counters for EVERY kernel of a command, one counter group per pass (rocprofv3 --kernel-trace --pmc only, as gpurun
# requires), aggregated per kernel by tools/pmc_aggregate.py.
# Usage (GPU box, repo root): bash tools/pmc_step.sh <tag> <command ...>      -> gpurun_out/pmc_<tag>.json
TAG=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for grp in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d /tmp/pmc_${TAG}_$i -o g -- "$@" > /tmp/pmc_${TAG}_$i.log 2>&1)
  echo "pass $i ($grp) rc=$?"
done
python tools/pmc_aggregate.py $TAG /tmp/pmc_${TAG}_ > gpurun_out/pmc_$TAG.json
python - gpurun_out/pmc_$TAG.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in sorted(d['kernels'].items(), key=lambda kv: -kv[1].get('total_us', 0))[:28]:
    print(f"{k[:58]:58s} n={v['calls']:4d} avg {v.get('avg_us', 0):8.1f} us  rd {v.get('read_MB', 0):8.1f} MB wr {v.get('write_MB', 0):8.1f} MB  "
          f"HBM {v.get('hbm_TBps', 0):5.2f} TB/s  L2hit {v.get('l2_hit', 0):4.2f}  mfma_busy {v.get('mfma_busy', 0):4.2f}  lds_conf {v.get('lds_conflict_frac', 0):5.3f}")
PY

"""Aggregate the per-pass rocprofv3 counter CSVs of tools/pmc_step.sh into one JSON: per kernel name the launch count, the average
duration (kernel-trace timestamps of the same passes) and per-launch averages of every counter, plus derived figures:
  read_MB   = TCC_EA0_RDREQ_sum x 64 B x 2  (gfx950 tallies the 128-B requests of a 16-B/lane streaming read at 64 B: MI355X_MICROARCH.md,
              HBM section; kernels whose reads are narrower are over-stated by up to 2x -- flagged by rd32_frac, the share of 32-B requests)
  write_MB  = 64 B x WRREQ_64B + 32 B x (WRREQ - WRREQ_64B)
  hbm_TBps  = (read + write) / average duration;  l2_hit = TCC_HIT / (TCC_HIT + TCC_MISS)
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES);  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
Usage: python tools/pmc_aggregate.py <tag> <pass-directory prefix>"""
import collections
import csv
import glob
import json
import re
import sys

tag, prefix = sys.argv[1], sys.argv[2]


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    name = re.sub(r'\((Conv|Res2|Se|Asp|Fbank|Cmn|Mom|Dense|Aam|SeGate)[A-Za-z0-9_<>]*Args[^)]*\)$', '', name)
    return name[:110]


cnt = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in sorted(glob.glob(prefix + '*')):
    if not glob.os.path.isdir(d):
        continue
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            cnt[short(r['Kernel_Name'])][r['Counter_Name']].append(float(r['Counter_Value']))
    for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            dur[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
out = {}
for k, c in cnt.items():
    if k.startswith('at::') or k.startswith('__amd') or 'rocblas' in k or 'elementwise' in k:
        continue
    m = {n: sum(v) / len(v) for n, v in c.items()}
    e = {'calls': max(len(v) for v in c.values()), 'counters_per_launch': {n: round(v, 1) for n, v in m.items()}}
    if dur.get(k):
        ds = sorted(dur[k])
        e['avg_us'] = round(sum(ds) / len(ds), 2)
        e['total_us'] = round(sum(ds), 1)
    if 'TCC_EA0_RDREQ_sum' in m:
        e['read_MB'] = round(m['TCC_EA0_RDREQ_sum'] * 128 / 1e6, 2)
        e['rd32_frac'] = round(m.get('TCC_EA0_RDREQ_32B_sum', 0.0) / max(1.0, m['TCC_EA0_RDREQ_sum']), 3)
    if 'TCC_EA0_WRREQ_sum' in m:
        w64 = m.get('TCC_EA0_WRREQ_64B_sum', 0.0)
        e['write_MB'] = round((64 * w64 + 32 * (m['TCC_EA0_WRREQ_sum'] - w64)) / 1e6, 2)
    if 'read_MB' in e and 'write_MB' in e and e.get('avg_us'):
        e['hbm_TBps'] = round((e['read_MB'] + e['write_MB']) / e['avg_us'], 3)       # MB / us = TB/s
    if 'TCC_HIT_sum' in m:
        e['l2_hit'] = round(m['TCC_HIT_sum'] / max(1.0, m['TCC_HIT_sum'] + m['TCC_MISS_sum']), 3)
    if 'SQ_BUSY_CU_CYCLES' in m:
        e['mfma_busy'] = round(m['SQ_VALU_MFMA_BUSY_CYCLES'] / max(1.0, 4 * m['SQ_BUSY_CU_CYCLES']), 3)
    if 'SQ_LDS_IDX_ACTIVE' in m:
        e['lds_conflict_frac'] = round(m['SQ_LDS_BANK_CONFLICT'] / max(1.0, m['SQ_LDS_IDX_ACTIVE']), 4)
    out[k] = e
import hashlib, os  # noqa: E402
_h = hashlib.sha1()
for _f in ('conv_gemm256.hip',):       # the same hash bench.py (kernel_git_hash) compares before quoting roofline.traffic
    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'voiceprintrecognition-paddlepaddle_amd', 'csrc', _f), 'rb') as _fh:
        _h.update(_fh.read())
json.dump({'csrc_hash': _h.hexdigest()[:12], '_source': f'tools/pmc_step.sh {tag}: rocprofv3 --kernel-trace --pmc <group>, one group per pass, per-launch averages over every launch '
                      'of the kernel in the command; corrections in tools/pmc_aggregate.py', 'kernels': out}, sys.stdout, indent=1)

"""cpu_baseline of bench.py at 16 / 32 / 64 / 128 threads, one process each (the thread pools are sized at first use):
python tools/cpu_threads_sweep.py  -> one line per count; the best count becomes bench.CPU_THREADS."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = ("import sys, json; sys.path.insert(0, %r); import torch, bench; "
        "fz, model, head, state, head_w = bench.build_ecapa(torch.device('cpu'), 'float32'); "
        "print('CPUBASE ' + json.dumps(bench.cpu_baseline(state, head_w, target_s=12.0)))") % ROOT
for n in (16, 32, 64, 128):
    env = dict(os.environ, VPMI_CPU_THREADS=str(n), OMP_NUM_THREADS=str(n))
    r = subprocess.run([sys.executable, '-c', code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith('CPUBASE ')]
    print(n, 'threads:', line[0][8:] if line else ('failed: ' + r.stderr[-400:]), flush=True)

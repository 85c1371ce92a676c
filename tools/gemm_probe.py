"""Micro-benchmark of the conv GEMM (no model around it): correctness vs torch matmul + TFLOP/s.
Usage: python tools/gemm_probe.py [reps]   (run under rocprofv3 --pmc ... for counters)"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import torch  # noqa: E402
from ppvector import _native as N  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
shapes = [(1536, 1536), (512, 512)] if len(sys.argv) < 3 else [tuple(int(v) for v in s.split('x')) for s in sys.argv[2:]]
lib, ctx = N.lib(), N.ctx(0)
B, T = 256, 298
M = B * T
g = torch.Generator(device='cuda').manual_seed(0)
for cin, cout in shapes:
    x = torch.randn((M, cin), device='cuda', generator=g).to(torch.bfloat16)
    w = (torch.randn((cout, cin), device='cuda', generator=g) / cin ** 0.5).to(torch.bfloat16)
    y = torch.empty((M, cout), device='cuda', dtype=torch.bfloat16)
    d = N.Conv1dDesc()
    d.dtype_in = d.dtype_out = N.VP_BF16
    d.B, d.T_in, d.T_out, d.Cin, d.Cout, d.KW, d.dilation, d.stride = B, T, T, cin, cout, 1, 1, 1
    d.pad_mode = N.VP_PAD_REFLECT
    d.x, d.ldx, d.w, d.y, d.ldy = x.data_ptr(), cin, w.data_ptr(), y.data_ptr(), cout
    N.check(lib.vp_conv1d_fwd(ctx, C.byref(d), N.stream_ptr()), ctx)
    torch.cuda.synchronize()
    ref = (x[:4096].float() @ w.float().t())
    err = (y[:4096].float() - ref).abs().max().item() / ref.abs().max().item()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in evs:
        a.record()
        N.check(lib.vp_conv1d_fwd(ctx, C.byref(d), N.stream_ptr()), ctx)
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in evs)[len(evs) // 2]
    print(f'{cin}x{cout}: {ms * 1e3:8.1f} us  {2.0 * M * cin * cout / ms / 1e9:7.1f} TFLOP/s  rel-err {err:.2e}', flush=True)

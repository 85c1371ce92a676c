"""Micro-benchmark of the 256-wide conv GEMM (no model around it): every schedule of the kernel on the bench shapes,
interleaved rounds in ONE process (A/B), full-output check against a torch matmul of the same bf16 operands.
Usage: python tools/gemm_probe.py [rounds] [schedules, e.g. 3,4] [CinxCout ...]
X3=1: the split-precision form of the same layers (hl32 operands and output on the 128 x 256 ring, csrc/conv_gemm256.hip; schedule ids are
ignored), checked against the f32 matmul of the values the planes carry; TF figures then count the algorithmic 2 M N K flops."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import torch  # noqa: E402
from ppvector import _native as N  # noqa: E402
from ppvector.models.utils import pack_hl32, unpack_hl32  # noqa: E402

X3 = os.environ.get('X3') == '1'

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
scheds = [int(v) for v in sys.argv[2].split(',')] if len(sys.argv) > 2 else [3, 4, 5]
shapes = [(1536, 1536), (512, 512)] if len(sys.argv) < 4 else [tuple(int(v) for v in s.split('x')) for s in sys.argv[3:]]
lib, ctx = N.lib(), N.ctx(0)
B, T = int(os.environ.get("GP_B", "256")), 298
M = B * T
g = torch.Generator(device='cuda').manual_seed(0)
for cin, cout in shapes:
    x = torch.randn((M, cin), device='cuda', generator=g).to(torch.bfloat16)
    w = (torch.randn((cout, cin), device='cuda', generator=g) / cin ** 0.5).to(torch.bfloat16)
    bias = torch.randn((cout,), device='cuda', generator=g)
    sc = torch.rand((cout,), device='cuda', generator=g) + 0.5
    sh = torch.randn((cout,), device='cuda', generator=g)
    if X3:
        x = pack_hl32(torch.randn((M, cin), device='cuda', generator=g))
        w = pack_hl32(torch.randn((cout, cin), device='cuda', generator=g) / cin ** 0.5)
        ref = torch.relu(unpack_hl32(x).double() @ unpack_hl32(w).double().t() + bias) * sc + sh
    else:
        ref = torch.relu(x.float() @ w.float().t() + bias) * sc + sh
    y = torch.empty((M, cout), device='cuda', dtype=torch.float32 if X3 else torch.bfloat16)
    d = N.Conv1dDesc()
    d.dtype_in = d.dtype_out = N.VP_HL32 if X3 else N.VP_BF16
    if X3 and os.environ.get('X3OUT') == 'f32':      # A/B: what the hl32 split + double stores of the epilogue cost
        d.dtype_out = N.VP_F32
    d.mfma_bf16 = 2 if X3 else 0
    d.B, d.T_in, d.T_out, d.Cin, d.Cout, d.KW, d.dilation, d.stride = B, T, T, cin, cout, 1, 1, 1
    d.pad_mode = N.VP_PAD_REFLECT
    d.x, d.ldx, d.w, d.y, d.ldy = x.data_ptr(), cin, w.data_ptr(), y.data_ptr(), cout
    d.bias, d.act, d.bn_scale, d.bn_shift = bias.data_ptr(), N.VP_ACT_RELU, sc.data_ptr(), sh.data_ptr()
    if os.environ.get('PSUM'):                      # the fused time sums of tdnn2 / MFA inside the step
        tiles, nseg = lib.vp_conv1d_tiles_m(B, T), lib.vp_conv1d_nseg(T)
        ps = torch.empty((tiles, nseg, cout), device='cuda'); pq = torch.empty((tiles, nseg, cout), device='cuda')
        d.psum, d.psumsq = ps.data_ptr(), pq.data_ptr()
    times = {s: [] for s in scheds}
    worst = {s: 0.0 for s in scheds}
    for r in range(rounds):
        for s in scheds:
            lib.vp_conv256_select(s)
            y.zero_()
            reps = 4
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            for a, b in evs:
                a.record()
                N.check(lib.vp_conv1d_fwd(ctx, C.byref(d), N.stream_ptr()), ctx)
                b.record()
            torch.cuda.synchronize()
            times[s] += [a.elapsed_time(b) for a, b in evs[1:]]
            err = ((unpack_hl32(y) if (X3 and os.environ.get('X3OUT') != 'f32') else y.float()) - ref).abs().max().item() / ref.abs().max().item()      # every element, every round (races show up here)
            worst[s] = max(worst[s], err)
    for s in scheds:
        t = sorted(times[s])
        med, mn = t[len(t) // 2], t[0]
        print(f'{cin}x{cout} sched {s}: median {med * 1e3:7.1f} us ({2.0 * M * cin * cout / med / 1e9:7.1f} TF)  min {mn * 1e3:7.1f} us '
              f'({2.0 * M * cin * cout / mn / 1e9:7.1f} TF)  worst rel-err over {rounds} rounds {worst[s]:.2e}', flush=True)
    del x, w, y, ref
lib.vp_conv256_select(-1)

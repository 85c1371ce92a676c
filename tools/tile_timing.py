"""Phase stamps of the 256-wide conv GEMM (needs the -DVP_TIMING variant: VPMI_LIB=.../libvpmi_timing.so).
X3=1: the split-precision ring (hl32 operands); the default schedule (6) runs 128-row tiles."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import torch  # noqa: E402
from ppvector import _native as N  # noqa: E402
from ppvector.models.utils import pack_hl32  # noqa: E402

X3 = os.environ.get('X3') == '1'

lib, ctx = N.lib(), N.ctx(0)
B, T = 256, 298
M = B * T
for cin, cout in [(1536, 1536), (512, 512)]:
    x = torch.randn((M, cin), device='cuda')
    w = torch.randn((cout, cin), device='cuda') / cin ** 0.5
    x, w = (pack_hl32(x), pack_hl32(w)) if X3 else (x.to(torch.bfloat16), w.to(torch.bfloat16))
    y = torch.empty((M, cout), device='cuda', dtype=torch.float32 if X3 else torch.bfloat16)
    rows = 256 if os.environ.get('VPMI_CONV256') in ('3', '4', '5') else 128
    ntile = ((M + rows - 1) // rows) * ((cout + 255) // 256)
    stamps = torch.zeros((ntile, 8), dtype=torch.int64, device='cuda')
    d = N.Conv1dDesc()
    d.dtype_in = d.dtype_out = N.VP_HL32 if X3 else N.VP_BF16
    d.mfma_bf16 = 2 if X3 else 0
    d.B, d.T_in, d.T_out, d.Cin, d.Cout, d.KW, d.dilation, d.stride = B, T, T, cin, cout, 1, 1, 1
    d.pad_mode = N.VP_PAD_REFLECT
    d.x, d.ldx, d.w, d.y, d.ldy = x.data_ptr(), cin, w.data_ptr(), y.data_ptr(), cout
    d.add_in = stamps.data_ptr()
    for _ in range(3):
        N.check(lib.vp_conv1d_fwd(ctx, C.byref(d), N.stream_ptr()), ctx)
    torch.cuda.synchronize()
    s = stamps.cpu().double() / 100.0          # us
    t0 = s[:, 0].min()
    pro, loop, epi = s[:, 1] - s[:, 0], s[:, 2] - s[:, 1], s[:, 3] - s[:, 2]
    print(f'{cin}x{cout}: tiles {ntile}  span {s[:, 3].max() - t0:.1f} us   prologue {pro.mean():.2f} (p90 {pro.quantile(0.9):.2f})  '
          f'k-loop {loop.mean():.2f} (p90 {loop.quantile(0.9):.2f})  epilogue {epi.mean():.2f} (p90 {epi.quantile(0.9):.2f})  '
          f'sum/tile {(s[:, 3] - s[:, 0]).mean():.2f}')
    print(f'   epilogue: barrier {(s[:, 4] - s[:, 2]).mean():.2f}  to-half0-end {(s[:, 6] - s[:, 4]).mean():.2f}  rest {(s[:, 3] - s[:, 6]).mean():.2f}')
    cyc = stamps[:, 5].cpu().double()
    print(f'   shader clock over the k-loop: {(cyc / (s[:, 2] - s[:, 1])).mean():.0f} MHz   ({cyc.mean():.0f} cycles)')
    raw = stamps[:, 7].cpu()
    print(f'   k-loop waits (wave 0, steps 1..): vmcnt {((raw >> 32).double() / 100.0).mean():.2f} us   barrier {((raw & 0xffffffff).double() / 100.0).mean():.2f} us')
    starts = (s[:, 0] - t0).sort().values
    print('   start-time deciles (us):', ' '.join(f'{starts[int(q * (ntile - 1))]:.0f}' for q in (0, .1, .2, .3, .4, .5, .6, .7, .8, .9, 1)))

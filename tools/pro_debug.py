import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import torch
from ppvector import _native as N
lib, ctx = N.lib(), N.ctx(0)
B, T, Cin, Cout = 1, 128, 192, 128
for tdt in (torch.float32, torch.bfloat16):
    x = torch.arange(Cin, dtype=torch.float32).repeat(B * T, 1).cuda().to(tdt) + 1      # x[m][c] = c + 1
    w = torch.zeros(Cout, Cin); w[torch.arange(Cout), torch.arange(Cout)] = 1.0          # picks channel n
    ps = (torch.arange(Cin, dtype=torch.float32) % 7 + 1).cuda()
    ph = (torch.arange(Cin, dtype=torch.float32) % 5).cuda()
    wd = w.cuda().to(tdt)
    y = torch.zeros(B * T, Cout, dtype=tdt, device='cuda')
    d = N.Conv1dDesc()
    d.dtype_in = d.dtype_out = N.dtype_id(tdt)
    d.B, d.T_in, d.T_out, d.Cin, d.Cout, d.KW, d.dilation, d.stride, d.pad_mode = B, T, T, Cin, Cout, 1, 1, 1, N.VP_PAD_ZERO
    d.x, d.ldx, d.w = x.data_ptr(), Cin, wd.data_ptr()
    d.pro_scale, d.pro_shift = ps.data_ptr(), ph.data_ptr()
    d.y, d.ldy = y.data_ptr(), Cout
    N.check(lib.vp_conv1d_fwd(ctx, C.byref(d), N.stream_ptr()), ctx)
    torch.cuda.synchronize()
    ref = (torch.arange(Cin, dtype=torch.float32) + 1) * ps.cpu() + ph.cpu()
    got = y.float().cpu()
    bad = (got[0] - ref[:Cout]).abs() > 0.05 * ref[:Cout].abs()
    print(tdt, 'row0 bad channels:', bad.nonzero().flatten().tolist()[:40])
    print(' got', got[0, :16].tolist()); print(' ref', ref[:16].tolist())
    rows_bad = ((got - ref[None, :Cout]).abs() > 0.05 * ref[None, :Cout].abs()).any(1).nonzero().flatten().tolist()
    print(' bad rows', rows_bad[:20], len(rows_bad))

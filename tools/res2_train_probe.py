"""Where does the fused Res2Net training kernel's time go at small batches?  vp_res2_train_fwd / _bwd (one workgroup per utterance, a
grid barrier per chunk conv) timed over (B, T): if the time hardly moves with T at B = 32, the chunk chain is bound by its seven grid
barriers and dependent global round trips, and splitting an utterance's rows over several workgroups cannot buy
the 3x it was asked to.  python tools/res2_train_probe.py"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import torch  # noqa: E402
import ppvector  # noqa: E402
from ppvector import _native as N  # noqa: E402
from ppvector.train.functions import Res2Fn  # noqa: E402

ppvector.set_train_amp(True)
lib, ctx = N.lib(), N.ctx(0)
S = 8
g = torch.Generator().manual_seed(3)
params = []
for _ in range(S - 1):
    params += [(torch.randn(64, 64, 3, generator=g) * 0.05).cuda(), torch.zeros(64).cuda(), torch.ones(64).cuda(), torch.zeros(64).cuda(),
               torch.zeros(64).cuda(), torch.ones(64).cuda()]
print('# res2_train kernels, 8 x 64 channels, dilation 2: microseconds per launch (median of 20, HIP events)')
print('#   B     T   rows    fwd_us   bwd_us   fwd_us/chunk')
for B, T in ((32, 298), (32, 150), (32, 75), (8, 298), (64, 298), (128, 298), (256, 298), (256, 75)):
    M = B * T
    x = torch.randn(M, 64 * S, generator=g).cuda().to(torch.bfloat16)
    out16 = torch.empty_like(x)
    z = torch.empty((S - 1, M, 64), dtype=torch.float32, device='cuda')
    inb = torch.empty((S - 1, M, 64), dtype=torch.bfloat16, device='cuda')
    stats = torch.empty((S - 1, 2, 64), dtype=torch.float32, device='cuda')
    cfg = dict(B=B, T=T, dilation=2, momentum=0.9, eps=1e-5)
    d = Res2Fn._fused_desc(x, None, cfg, params, S)
    d.z, d.inb, d.stats, d.out_bf16 = z.data_ptr(), inb.data_ptr(), stats.data_ptr(), out16.data_ptr()
    ws = torch.empty(lib.vp_res2_train_workspace_bytes(B, S), dtype=torch.uint8, device='cuda')
    dout = torch.randn(M, 64 * S, generator=g).cuda()
    dx = torch.empty_like(dout)
    dzb = torch.empty((S - 1, M, 64), dtype=torch.bfloat16, device='cuda')
    dvec = torch.empty((S - 1, 3, 64), dtype=torch.float32, device='cuda')
    db = N.Res2TrainDesc()
    db.B, db.T, db.C, db.scale, db.width, db.dil = B, T, 64 * S, S, 64, 2
    db.momentum, db.eps = 0.9, 1e-5
    db.x, db.out = dout.data_ptr(), dx.data_ptr()
    for i in range(S - 1):
        db.w[i], db.gamma[i] = params[6 * i].data_ptr(), params[6 * i + 2].data_ptr()
    db.z, db.stats, db.dzb, db.dvec = z.data_ptr(), stats.data_ptr(), dzb.data_ptr(), dvec.data_ptr()

    def time(fn, desc):
        ts = []
        for i in range(24):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            N.check(fn(ctx, C.byref(desc), ws.data_ptr(), ws.numel(), N.stream_ptr()), ctx)
            b.record()
            torch.cuda.synchronize()
            if i >= 4:
                ts.append(a.elapsed_time(b) * 1e3)
        return sorted(ts)[len(ts) // 2]

    tf = time(lib.vp_res2_train_fwd, d)
    tb = time(lib.vp_res2_train_bwd, db)
    print(f'  {B:4d} {T:5d} {M:6d}  {tf:8.1f} {tb:8.1f}   {tf / (S - 1):8.1f}')
assert lib.vp_grid_barrier_status(ctx) == 0

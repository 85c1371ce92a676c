"""Weight-gradient kernels of the wide 1x1 layers, timed in isolation: the 256-tile kernel on transposing LDS reads (csrc/wgrad_tr.hip)
against the 128-tile register-transpose kernel (VPMI_WGRAD_TR256_OFF=1, a second process).  Usage: python tools/wgrad_probe.py [reps]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import torch
from ppvector import _native as N
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
lib, ctx = N.lib(), N.ctx(torch.device('cuda', 0))
for (M, Cout, Cin) in ((76288, 512, 512), (76288, 1536, 1536), (9536, 512, 512), (9536, 1536, 1536)):
    x = torch.randn(M, Cin, device='cuda').to(torch.bfloat16)
    dz = torch.randn(M, Cout, device='cuda').to(torch.bfloat16)
    d = N.Conv1dDesc()
    d.dtype_in, d.dtype_out = N.VP_BF16, N.VP_F32
    d.B, d.T_in, d.T_out, d.Cin, d.Cout, d.KW, d.dilation, d.stride = 1, M, M, Cin, Cout, 1, 1, 1
    d.pad_mode, d.pad_left = N.VP_PAD_ZERO, 0
    d.x, d.ldx, d.xoff = x.data_ptr(), Cin, 0
    d.mfma_bf16 = 1
    ws = torch.empty(lib.vp_conv1d_wgrad_workspace_bytes(C.byref(d)), dtype=torch.uint8, device='cuda')
    dW = torch.empty((Cout, Cin), dtype=torch.float32, device='cuda')
    run = lambda: N.check(lib.vp_conv1d_wgrad_bf16_oik(ctx, C.byref(d), dz.data_ptr(), Cout, dW.data_ptr(), ws.data_ptr(), ws.numel(), N.stream_ptr()), ctx)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    ref = dz.float().t() @ x.float()
    rel = float((dW - ref).norm() / ref.norm())
    print(f'M={M} {Cout}x{Cin}: {us:8.1f} us per call (kernel + partial sum)  {2.0 * M * Cout * Cin / us / 1e6:7.1f} TFLOP/s   rel err vs torch f32 {rel:.2e}')

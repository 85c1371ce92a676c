"""Co-run screen: does a kernel corrupt a well-behaved kernel that runs right behind it / beside it?  Canary = se_gate on FIXED
inputs, launched after the candidate on each of four streams; any output that differs from the quiet run is a hit (X).
Found in round 3: conv_gemm_kernel<bf16, bf16, 128, 1x1> (no longer dispatched).  Usage: python tools/stress_corun.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import torch
from ppvector import _native as N
lib, ctx = N.lib(), N.ctx(0)
dev = torch.device('cuda')
B, T = 128, 298
M = B * T
g = torch.Generator(device='cuda').manual_seed(0)
streams = [torch.cuda.Stream() for _ in range(4)]
cout, H = 512, 128
sh = torch.randn((cout,), device=dev, generator=g)
w1 = torch.randn((cout, H), device=dev, generator=g) / cout ** 0.5
b1 = torch.randn((H,), device=dev, generator=g)
w2 = torch.randn((H, cout), device=dev, generator=g) / H ** 0.5
b2 = torch.randn((cout,), device=dev, generator=g)
tiles, nseg = lib.vp_conv1d_tiles_m(B, T), lib.vp_conv1d_nseg(T)
psfix = torch.randn((tiles, nseg, cout), device=dev, generator=g)
def gate(out):
    N.check(lib.vp_se_gate_fwd(ctx, psfix.data_ptr(), sh.data_ptr(), B, T, cout, H, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), out.data_ptr(), N.stream_ptr()), ctx)
ref = torch.zeros((B, cout), device=dev); gate(ref); torch.cuda.synchronize()
def conv_maker(dtype, cin, co, sched, out=None, amp=0):
    tdt = torch.bfloat16 if dtype == 'bf16' else torch.float32
    odt = tdt if out is None else (torch.bfloat16 if out == 'bf16' else torch.float32)
    x = torch.randn((M, cin), device=dev, generator=g).to(tdt)
    w = (torch.randn((co, cin), device=dev, generator=g) / cin ** 0.5).to(tdt)
    def make():
        y = torch.zeros((M, co), device=dev, dtype=odt)
        d = N.Conv1dDesc()
        d.dtype_in, d.dtype_out = N.dtype_id(tdt), N.dtype_id(odt)
        d.mfma_bf16 = amp
        d.B, d.T_in, d.T_out, d.Cin, d.Cout, d.KW, d.dilation, d.stride = B, T, T, cin, co, 1, 1, 1
        d.pad_mode = N.VP_PAD_REFLECT
        d.x, d.ldx, d.w, d.y, d.ldy = x.data_ptr(), cin, w.data_ptr(), y.data_ptr(), co
        d.act = N.VP_ACT_RELU
        return (d, y, x, w)
    def run(s):
        lib.vp_conv256_select(sched)
        N.check(lib.vp_conv1d_fwd(ctx, C.byref(s[0]), N.stream_ptr()), ctx)
    return make, run
os.environ.setdefault('VPMI_BN128', '1')          # the screen needs the suspect instantiation back
cases = {'bf16 128-wide (sched 0) 512->512': conv_maker('bf16', 512, 512, 0), 'bf16 128-wide 512->128 (Cout 128)': conv_maker('bf16', 512, 128, 6),
         'bf16 -> f32 128-wide (sched 0)': conv_maker('bf16', 512, 512, 0, out='f32'), 'f32 tensors, bf16 MFMA (amp) 128-wide': conv_maker('f32', 512, 512, 0, amp=1),
         'f32 tensors, bf16 MFMA (amp) 64-wide': conv_maker('f32', 64, 64, 0, amp=1), 'bf16 -> f32 ring 128x256 (sched 6)': conv_maker('bf16', 512, 512, 6, out='f32'),
         'bf16 64-wide 64->64': conv_maker('bf16', 64, 64, 6),
         'f32 128-wide 512->512': conv_maker('f32', 512, 512, 0), 'bf16 ring 128x256 (sched 6)': conv_maker('bf16', 512, 512, 6),
         'bf16 ring 256 (sched 4)': conv_maker('bf16', 512, 512, 4), 'bf16 two-stage 256 (sched 3)': conv_maker('bf16', 512, 512, 3)}
for name, (make, run) in cases.items():
    sets = [(make(), torch.zeros((B, cout), device=dev)) for _ in range(4)]
    pat = []
    for it in range(20):
        for (s, out), st in zip(sets, streams):
            with torch.cuda.stream(st):
                run(s)
                gate(out)
        torch.cuda.synchronize()
        pat.append(''.join('X' if not torch.equal(out, ref) else '.' for s, out in sets))
    print(f'{name:40s}', ' '.join(pat), flush=True)

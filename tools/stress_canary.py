"""Co-run screen with canaries (the study behind DESIGN.md section 8): which instruction pattern of a VICTIM kernel returns wrong values
when an MFMA-heavy kernel runs beside it?  Four streams, each: neighbour -> victim.  Neighbours: the 128-wide bf16 conv GEMM
(schedule 0), the 128 x 256 ring GEMM, torch matmul (bf16 / f32), nothing.  Victims: se_gate on fixed inputs (libvpmi) and the
canaries of tools/canary.hip -- registers / LDS at rest, load data, and se_gate's inner loop piece by piece (which consumer of a freshly
returned load: v_fma_f32, v_pk_fma_f32 with / without op_sel, behind s_nop, v_mov_b64, DPP, v_lshl_add_u64, MFMA operands).
Result (profiles/r04_corun_canary.log): only packed-f32 consumers fail; a library built without them is clean.
Usage: tools/build_canary.sh && python tools/stress_canary.py [iterations]   (VICTIMS=..., CULPRITS=... select; VPMI_LIB = the build)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd')
sys.path.insert(0, PKG)
import torch
from ppvector import _native as N
IT = int(sys.argv[1]) if len(sys.argv) > 1 else 20
NS = int(os.environ.get('NSTREAMS', '4'))
lib, ctx = N.lib(), N.ctx(0)
can = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bin', 'libcanary.so'))     # bash tools/build_canary.sh
can.canary_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
can.canary_fill_launch.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
can.canary_fill_w_launch.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
can.canary_fill_a_launch.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
dev = torch.device('cuda')
B, T = 128, 298
M = B * T
g = torch.Generator(device='cuda').manual_seed(0)
streams = [torch.cuda.Stream() for _ in range(NS)]
cout, H = 512, 128
sh = torch.randn((cout,), device=dev, generator=g)
w1 = torch.randn((cout, H), device=dev, generator=g) / cout ** 0.5
b1 = torch.randn((H,), device=dev, generator=g)
w2 = torch.randn((H, cout), device=dev, generator=g) / H ** 0.5
b2 = torch.randn((cout,), device=dev, generator=g)
tiles, nseg = lib.vp_conv1d_tiles_m(B, T), lib.vp_conv1d_nseg(T)
psfix = torch.randn((tiles, nseg, cout), device=dev, generator=g)
WORDS = 1 << 16
cbuf = torch.zeros((WORDS,), device=dev, dtype=torch.int32)
can.canary_fill_launch(cbuf.data_ptr(), WORDS, N.stream_ptr())
wbuf = torch.zeros((128, 512), device=dev)
can.canary_fill_w_launch(wbuf.data_ptr(), 512, N.stream_ptr())
abuf = torch.zeros((4096 * 512,), device=dev, dtype=torch.bfloat16)
can.canary_fill_a_launch(abuf.data_ptr(), 4096, N.stream_ptr())
torch.cuda.synchronize()


def gate(out):
    N.check(lib.vp_se_gate_fwd(ctx, psfix.data_ptr(), sh.data_ptr(), B, T, cout, H, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), out.data_ptr(), N.stream_ptr()), ctx)


ref = torch.zeros((B, cout), device=dev); gate(ref); torch.cuda.synchronize()
# victims: name -> (launch(rep_or_out), is_gate)
CAN = {'vgpr40': (0, 128, 1024, 50, 0), 'lds21k': (1, 128, 1024, 60, 5376), 'vmem': (2, 128, 1024, 16, WORDS), 'bcast': (3, 128, 1024, 50, 0),
       'vgpr100': (4, 128, 1024, 30, 0),
       'fma.reg': (5, 128, 1024, 40, 0), 'fma.lds': (6, 128, 1024, 40, 0), 'fma.vmem': (7, 128, 1024, 12, 0),
       'pk.reg': (9, 128, 1024, 40, 0), 'pk.lds': (10, 128, 1024, 40, 0), 'pk.vmem': (11, 128, 1024, 12, 0),
       'pk.nop2': (12, 256, 512, 12, 0), 'pk.nop7': (13, 256, 512, 12, 0), 'mov64': (14, 256, 512, 12, 0), 'dpp': (15, 256, 512, 12, 0),
       'pk.newloads': (16, 256, 512, 12, 0), 'pk.mul': (17, 256, 512, 12, 0), 'lshl_add_u64': (18, 256, 512, 12, 0),
       'mfma': (20, 128, 1024, 12, 4096), 'pk.plain': (19, 256, 512, 12, 0), 'pk.nop0': (21, 256, 512, 12, 0),
       'pkv.noopsel': (22, 128, 1024, 12, 0), 'pkv.nop2': (23, 128, 1024, 12, 0), 'pkv.mov64': (24, 128, 1024, 12, 0), 'pkv.xreg': (25, 128, 1024, 12, 0),
       'pkv.512': (11, 256, 512, 12, 0)}


def canary(kind_name, rep):
    k, blocks, threads, iters, words = CAN[kind_name]
    rc = can.canary_launch(k, blocks, threads, iters, (abuf if k == 20 else wbuf if k >= 5 else cbuf).data_ptr(), words, rep.data_ptr(), N.stream_ptr())
    assert rc == 0, rc


def conv_maker(cin, co, sched):
    x = torch.randn((M, cin), device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn((co, cin), device=dev, generator=g) / cin ** 0.5).to(torch.bfloat16)

    def make():
        y = torch.zeros((M, co), device=dev, dtype=torch.bfloat16)
        d = N.Conv1dDesc()
        d.dtype_in = d.dtype_out = N.VP_BF16
        d.B, d.T_in, d.T_out, d.Cin, d.Cout, d.KW, d.dilation, d.stride = B, T, T, cin, co, 1, 1, 1
        d.pad_mode = N.VP_PAD_REFLECT
        d.x, d.ldx, d.w, d.y, d.ldy = x.data_ptr(), cin, w.data_ptr(), y.data_ptr(), co
        d.act = N.VP_ACT_RELU
        return (d, y, x, w)

    def run(s):
        lib.vp_conv256_select(sched)
        N.check(lib.vp_conv1d_fwd(ctx, C.byref(s[0]), N.stream_ptr()), ctx)
    return make, run


def matmul_maker(dt=torch.bfloat16):
    a = torch.randn((2048, 2048), device=dev, dtype=dt)

    def make():
        return (a,)

    def run(s):
        s[0] @ s[0]
    return make, run


culprits = {
    'conv128 (schedule 0)': conv_maker(512, 512, 0),
    'ring 128x256 (schedule 6)': conv_maker(512, 512, 6),
    'torch matmul bf16 2048': matmul_maker(),
    'torch matmul f32 2048': matmul_maker(torch.float32),
    'nothing': (lambda: (0,), lambda s: None),
}
only = os.environ.get('CULPRITS')
victims = os.environ.get('VICTIMS', 'gate,fma.reg,pk.reg,fma.lds,pk.lds,fma.vmem,pk.vmem').split(',')
for cname, (make, run) in culprits.items():
    if only and not any(o in cname for o in only.split(',')):
        continue
    line = []
    for v in victims:
        sets = [(make(), torch.zeros((B, cout), device=dev), torch.zeros((256,), device=dev, dtype=torch.int32)) for _ in range(NS)]
        hits, first = 0, None
        for it in range(IT):
            for (s, out, rep), st in zip(sets, streams):
                with torch.cuda.stream(st):
                    run(s)
                    if v.startswith('gate'):
                        gate(out)          # VPMI_SE_NOPK (read once per process) picks the flavour of the gate kernel
                    else:
                        canary(v, rep)
            torch.cuda.synchronize()
            for s, out, rep in sets:
                if v.startswith('gate'):
                    if not torch.equal(out, ref):
                        hits += 1
                        if first is None:
                            dd = (out - ref).abs()
                            r = int(torch.nonzero(dd.max(dim=1)[0] > 0).flatten()[0])
                            cols = torch.nonzero(dd[r] > 0).flatten().tolist()
                            first = f'row {r} cols {cols[:3]}..x{len(cols)}'
                else:
                    h = rep.cpu()
                    if int(h[0]):
                        hits += 1
                        if first is None:
                            u = [int(x) & 0xffffffff for x in h[:8]]
                            log = [(int(h[8 + 2 * i]) & 0xffff, (int(h[9 + 2 * i]) >> 16) & 0xffff) for i in range(min(int(h[0]), 12))]
                            first = f'n={u[0]} blk {u[1]} tid {u[2]} (lane {u[2] & 63}) slot {u[3]} it {u[4]} got {u[5]:08x} want {u[6]:08x} log(tid,slot) {log}'
                        rep.zero_()
        line.append(f'{v}: {hits}/{IT * NS}' + (f' [{first}]' if first else ''))
    print(f'{cname:36s} ' + ' | '.join(line), flush=True)

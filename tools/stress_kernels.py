"""Race / uninitialised-read screen, kernel by kernel: the same launch on several streams at once (own outputs, own scratch,
outputs and scratch pre-filled with different garbage), every result compared bit for bit with a quiet single-stream run.
Usage: python tools/stress_kernels.py [iterations]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import torch  # noqa: E402
from ppvector import _native as N  # noqa: E402

it = int(sys.argv[1]) if len(sys.argv) > 1 else 30
lib, ctx = N.lib(), N.ctx(0)
dev = torch.device('cuda')
B, T = 128, 298
M = B * T
g = torch.Generator(device='cuda').manual_seed(0)
streams = [torch.cuda.Stream() for _ in range(4)]


def conv_case(cin, cout, sched, psum, rowbias, tanh):
    x = torch.randn((M, cin), device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn((cout, cin), device=dev, generator=g) / cin ** 0.5).to(torch.bfloat16)
    bias = torch.randn((cout,), device=dev, generator=g)
    sc = torch.rand((cout,), device=dev, generator=g) + 0.5
    sh = torch.randn((cout,), device=dev, generator=g)
    rb = torch.randn((B, cout), device=dev, generator=g)
    tiles, nseg = lib.vp_conv1d_tiles_m(B, T), lib.vp_conv1d_nseg(T)

    def make(fill):
        y = torch.full((M, cout), fill, device=dev, dtype=torch.bfloat16)
        ps = torch.full((tiles, nseg, cout), fill, device=dev)
        pq = torch.full((tiles, nseg, cout), fill, device=dev)
        d = N.Conv1dDesc()
        d.dtype_in = d.dtype_out = N.VP_BF16
        d.B, d.T_in, d.T_out, d.Cin, d.Cout, d.KW, d.dilation, d.stride = B, T, T, cin, cout, 1, 1, 1
        d.pad_mode = N.VP_PAD_REFLECT
        d.x, d.ldx, d.w, d.y, d.ldy = x.data_ptr(), cin, w.data_ptr(), y.data_ptr(), cout
        d.bias, d.act, d.bn_scale, d.bn_shift = bias.data_ptr(), N.VP_ACT_RELU, sc.data_ptr(), sh.data_ptr()
        if psum:
            d.psum, d.psumsq = ps.data_ptr(), pq.data_ptr()
        if rowbias:
            d.rowbias = rb.data_ptr()
        if tanh:
            d.act2 = N.VP_ACT_TANH
        return d, (y, ps, pq)

    lib.vp_conv256_select(sched)
    d0, ref = make(0.0)
    N.check(lib.vp_conv1d_fwd(ctx, C.byref(d0), N.stream_ptr()), ctx)
    torch.cuda.synchronize()
    bad = 0
    sets = [make(float(i + 1) * 7.0) for i in range(4)]
    for _ in range(it):
        for (d, outs), st in zip(sets, streams):
            with torch.cuda.stream(st):
                N.check(lib.vp_conv1d_fwd(ctx, C.byref(d), N.stream_ptr()), ctx)
        torch.cuda.synchronize()
        for d, outs in sets:
            if not torch.equal(outs[0], ref[0]) or (psum and (not torch.equal(outs[1], ref[1]) or not torch.equal(outs[2], ref[2]))):
                bad += 1
    lib.vp_conv256_select(-1)
    print(f'conv {cin}->{cout} sched {sched} psum={psum} rowbias={rowbias} tanh={tanh}: {bad} / {4 * it} concurrent launches differ', flush=True)


conv_case(512, 512, 0, True, False, False)
conv_case(512, 512, 0, False, False, False)
conv_case(1536, 128, 0, False, True, True)
conv_case(512, 512, 6, True, False, False)
conv_case(512, 512, 6, False, False, False)
conv_case(1536, 1536, 6, True, False, False)
conv_case(512, 512, 4, True, False, False)


def pair_case(sched, n_streams=4, own_psum=False):
    """tdnn2 (512 -> 512 with fused time sums) -> se_gate on each stream, as inside the forward: the gate must not depend on what else runs."""
    cin = cout = 512
    H = 128
    x = [torch.randn((M, cin), device=dev, generator=g).to(torch.bfloat16) for _ in range(3)]
    w = (torch.randn((cout, cin), device=dev, generator=g) / cin ** 0.5).to(torch.bfloat16)
    bias = torch.randn((cout,), device=dev, generator=g)
    sc = torch.rand((cout,), device=dev, generator=g) + 0.5
    sh = torch.randn((cout,), device=dev, generator=g)
    w1 = torch.randn((cout, H), device=dev, generator=g) / cout ** 0.5
    b1 = torch.randn((H,), device=dev, generator=g)
    w2 = torch.randn((H, cout), device=dev, generator=g) / H ** 0.5
    b2 = torch.randn((cout,), device=dev, generator=g)
    tiles, nseg = lib.vp_conv1d_tiles_m(B, T), lib.vp_conv1d_nseg(T)
    lib.vp_conv256_select(sched)

    def make():
        y = torch.zeros((M, cout), device=dev, dtype=torch.bfloat16)
        pss = [torch.zeros((tiles, nseg, cout), device=dev) for _ in range(3)]
        ps = pss[0]
        gate = [torch.zeros((B, cout), device=dev) for _ in range(3)]
        ds = []
        for k in range(3):
            if own_psum:
                ps = pss[k]
            d = N.Conv1dDesc()
            d.dtype_in = d.dtype_out = N.VP_BF16
            d.B, d.T_in, d.T_out, d.Cin, d.Cout, d.KW, d.dilation, d.stride = B, T, T, cin, cout, 1, 1, 1
            d.pad_mode = N.VP_PAD_REFLECT
            d.x, d.ldx, d.w, d.y, d.ldy = x[k].data_ptr(), cin, w.data_ptr(), y.data_ptr(), cout
            d.bias, d.act, d.bn_scale, d.bn_shift = bias.data_ptr(), N.VP_ACT_RELU, sc.data_ptr(), sh.data_ptr()
            d.psum = ps.data_ptr()
            ds.append((d, ps))
        return ds, y, ps, gate

    def run(s):
        ds, y, ps, gate = s
        for k in range(3):                      # three blocks reuse ONE psum buffer, as the engine's workspace does
            N.check(lib.vp_conv1d_fwd(ctx, C.byref(ds[k][0]), N.stream_ptr()), ctx)
            N.check(lib.vp_se_gate_fwd(ctx, ds[k][1].data_ptr(), sh.data_ptr(), B, T, cout, H, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                       gate[k].data_ptr(), N.stream_ptr()), ctx)

    ref = make()
    run(ref)
    torch.cuda.synchronize()
    sets = [make() for _ in range(n_streams)]
    bad = [0, 0, 0]
    for _ in range(it):
        for s, st in zip(sets, streams):
            with torch.cuda.stream(st):
                run(s)
        torch.cuda.synchronize()
        for s in sets:
            for k in range(3):
                if not torch.equal(s[3][k], ref[3][k]):
                    bad[k] += 1
    lib.vp_conv256_select(-1)
    print(f'pair conv(psum) -> se_gate x3 sched {sched}, {n_streams} streams, own psum per pair {own_psum}: gates differing per block {bad} / {n_streams * it}', flush=True)


pair_case(0)
pair_case(0, 4, True)
pair_case(6)
pair_case(0, 1)


def visibility_case(sched, n_streams=4):
    """conv (fused time sums) -> torch clone of the sums on the same stream: does a dependent read see the conv's writes?"""
    cin = cout = 512
    x = torch.randn((M, cin), device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn((cout, cin), device=dev, generator=g) / cin ** 0.5).to(torch.bfloat16)
    bias = torch.randn((cout,), device=dev, generator=g)
    tiles, nseg = lib.vp_conv1d_tiles_m(B, T), lib.vp_conv1d_nseg(T)
    lib.vp_conv256_select(sched)

    def make():
        y = torch.zeros((M, cout), device=dev, dtype=torch.bfloat16)
        ps = torch.zeros((tiles, nseg, cout), device=dev)
        d = N.Conv1dDesc()
        d.dtype_in = d.dtype_out = N.VP_BF16
        d.B, d.T_in, d.T_out, d.Cin, d.Cout, d.KW, d.dilation, d.stride = B, T, T, cin, cout, 1, 1, 1
        d.pad_mode = N.VP_PAD_REFLECT
        d.x, d.ldx, d.w, d.y, d.ldy = x.data_ptr(), cin, w.data_ptr(), y.data_ptr(), cout
        d.bias, d.act = bias.data_ptr(), N.VP_ACT_RELU
        d.psum = ps.data_ptr()
        return d, y, ps

    ref = make()
    N.check(lib.vp_conv1d_fwd(ctx, C.byref(ref[0]), N.stream_ptr()), ctx)
    torch.cuda.synchronize()
    sets = [make() for _ in range(n_streams)]
    bad_dep, bad_final = 0, 0
    for _ in range(it):
        clones = []
        for s, st in zip(sets, streams):
            with torch.cuda.stream(st):
                s[2].zero_()
                N.check(lib.vp_conv1d_fwd(ctx, C.byref(s[0]), N.stream_ptr()), ctx)
                clones.append(s[2].clone())
        torch.cuda.synchronize()
        for s, c in zip(sets, clones):
            bad_dep += int(not torch.equal(c, ref[2]))
            bad_final += int(not torch.equal(s[2], ref[2]))
    lib.vp_conv256_select(-1)
    print(f'visibility sched {sched}, {n_streams} streams: dependent clone differs {bad_dep}, final buffer differs {bad_final} / {n_streams * it}', flush=True)


visibility_case(0)
visibility_case(6)


def gate_alone(n_streams=4, with_conv=False):
    cout, H = 512, 128
    tiles, nseg = lib.vp_conv1d_tiles_m(B, T), lib.vp_conv1d_nseg(T)
    ps = torch.randn((tiles, nseg, cout), device=dev, generator=g)
    sh = torch.randn((cout,), device=dev, generator=g)
    w1 = torch.randn((cout, H), device=dev, generator=g) / cout ** 0.5
    b1 = torch.randn((H,), device=dev, generator=g)
    w2 = torch.randn((H, cout), device=dev, generator=g) / H ** 0.5
    b2 = torch.randn((cout,), device=dev, generator=g)
    ref = torch.zeros((B, cout), device=dev)
    N.check(lib.vp_se_gate_fwd(ctx, ps.data_ptr(), sh.data_ptr(), B, T, cout, H, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), ref.data_ptr(), N.stream_ptr()), ctx)
    torch.cuda.synchronize()
    outs = [torch.zeros((B, cout), device=dev) for _ in range(n_streams)]
    a = torch.randn((4096, 4096), device=dev)
    bad = 0
    for _ in range(it):
        for o, st in zip(outs, streams):
            with torch.cuda.stream(st):
                if with_conv:
                    (a @ a)
                N.check(lib.vp_se_gate_fwd(ctx, ps.data_ptr(), sh.data_ptr(), B, T, cout, H, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), o.data_ptr(), N.stream_ptr()), ctx)
        torch.cuda.synchronize()
        bad += sum(int(not torch.equal(o, ref)) for o in outs)
    print(f'se_gate alone, {n_streams} streams, other work {with_conv}: {bad} / {n_streams * it} differ', flush=True)


gate_alone(4, False)
gate_alone(4, True)
gate_alone(1, False)

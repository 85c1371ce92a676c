"""GPU parity tests of the training path (f32 engine): every backward entry point and the whole TDNN training step
against PyTorch autograd over the CPU oracle graph (float64 where cheap).  Run with -m gpu on an MI355X."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import models as om

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def N():
    from ppvector import _native as N
    if not torch.cuda.is_available():
        pytest.fail('no GPU visible: these tests must run on an MI355X (no CPU fallback exists)')
    N.ctx(0)
    return N


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp(min=1e-30)).item()


@pytest.mark.parametrize('case', [
    # B, T, Cin, Cout, kw, dil, pad
    (3, 50, 80, 64, 5, 1, 'none'),        # TDNN layer 1 geometry (Cin % 64 != 0: k-column tiles straddle taps)
    (2, 41, 64, 128, 3, 2, 'none'),
    (2, 37, 128, 64, 3, 3, 'none'),
    (4, 33, 64, 192, 1, 1, 'none'),
    (2, 45, 64, 64, 3, 2, 'zero'),
    (70, 131, 64, 64, 3, 1, 'none'),      # many row splits
    (3, 40, 64, 64, 3, 4, 'reflect'),     # ECAPA Res2 conv: reflect padding (adjoint = full conv + mirror fold)
    (2, 33, 80, 128, 5, 1, 'reflect'),    # ECAPA block0
])
def test_conv_block_grads_vs_autograd(N, case):
    """conv (+bias) -> ReLU -> BatchNorm(batch statistics): output, running statistics and all five gradients."""
    from ppvector.train.functions import ConvBlock
    B, T, Cin, Cout, kw, dil, pad = case
    g = torch.Generator().manual_seed(5 + kw + dil)
    x = torch.randn(B, T, Cin, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Cout, Cin, kw, generator=g, dtype=torch.float64) / (Cin * kw) ** 0.5).requires_grad_()
    b = torch.randn(Cout, generator=g, dtype=torch.float64, requires_grad=True)
    ga = (torch.rand(Cout, generator=g, dtype=torch.float64) + 0.5).requires_grad_()
    be = torch.randn(Cout, generator=g, dtype=torch.float64, requires_grad=True)
    p = dil * (kw - 1) // 2 if pad == 'zero' else 0
    xt = x.transpose(1, 2)
    if pad == 'reflect':
        xt = F.pad(xt, (dil * (kw - 1) // 2,) * 2, mode='reflect')
    z = F.relu(F.conv1d(xt, w, b, dilation=dil, padding=p))
    mean, var = z.mean(dim=(0, 2)), z.var(dim=(0, 2), unbiased=False)
    y = ((z - mean[None, :, None]) / torch.sqrt(var[None, :, None] + 1e-5) * ga[None, :, None] + be[None, :, None]).transpose(1, 2)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    xd = x.detach().float().reshape(B * T, Cin).cuda().requires_grad_()
    wd, bd = w.detach().float().cuda().requires_grad_(), b.detach().float().cuda().requires_grad_()
    gd, hd = ga.detach().float().cuda().requires_grad_(), be.detach().float().cuda().requires_grad_()
    rm, rv = torch.zeros(Cout, device='cuda'), torch.ones(Cout, device='cuda')
    out = ConvBlock.apply(xd, wd, bd, None, gd, hd, rm, rv, dict(B=B, T=T, dilation=dil, pad=pad, relu=True))
    out.backward(dy.float().reshape(-1, Cout).cuda())
    To = y.shape[1]
    assert rel(out.reshape(B, To, Cout), y.detach()) < 2e-6
    assert rel(rm, 0.1 * mean.detach()) < 2e-6 and rel(rv, 0.9 + 0.1 * var.detach()) < 2e-6
    for name, got, ref in (('dx', xd.grad.reshape(B, T, Cin), x.grad), ('dW', wd.grad, w.grad), ('dbias', bd.grad, b.grad),
                           ('dgamma', gd.grad, ga.grad), ('dbeta', hd.grad, be.grad)):
        assert rel(got, ref) < 2e-5, (name, rel(got, ref))


def test_asp_pieces_vs_autograd(N):
    from ppvector.train.functions import AttnStats, BNRows, TimeStats
    B, T, Cc = 3, 47, 96
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, T, Cc, generator=g, dtype=torch.float64, requires_grad=True)
    e = (torch.randn(B, T, Cc, generator=g, dtype=torch.float64) * 2).requires_grad_()
    al = torch.softmax(e, dim=1)
    mu = (al * x).sum(1)
    sd = torch.sqrt(((al * (x - mu[:, None]) ** 2).sum(1)).clamp(min=1e-12))
    m0 = x.mean(1)
    s0 = torch.sqrt((((x - m0[:, None]) ** 2).mean(1)).clamp(min=1e-12))
    dp, ds = torch.randn(B, 2 * Cc, generator=g, dtype=torch.float64), torch.randn(B, 2 * Cc, generator=g, dtype=torch.float64)
    (torch.cat([mu, sd], 1) * dp).sum().backward(retain_graph=True)
    gx_attn, ge = x.grad.clone(), e.grad.clone()
    x.grad = None
    (torch.cat([m0, s0], 1) * ds).sum().backward()
    gx_stats = x.grad.clone()
    xd = x.detach().float().reshape(B * T, Cc).cuda().requires_grad_()
    ed = e.detach().float().reshape(B * T, Cc).cuda().requires_grad_()
    pooled = AttnStats.apply(ed, xd, B, T)
    assert rel(pooled, torch.cat([mu, sd], 1).detach()) < 2e-6
    pooled.backward(dp.float().cuda())
    assert rel(ed.grad.reshape(B, T, Cc), ge) < 2e-5 and rel(xd.grad.reshape(B, T, Cc), gx_attn) < 2e-5
    xd2 = x.detach().float().reshape(B * T, Cc).cuda().requires_grad_()
    st = TimeStats.apply(xd2, B, T)
    assert rel(st, torch.cat([m0, s0], 1).detach()) < 2e-6
    st.backward(ds.float().cuda())
    assert rel(xd2.grad.reshape(B, T, Cc), gx_stats) < 2e-5
    # BatchNorm1D on (B, C) rows
    v = torch.randn(16, 64, generator=g, dtype=torch.float64, requires_grad=True)
    ga = (torch.rand(64, generator=g, dtype=torch.float64) + 0.5).requires_grad_()
    be = torch.randn(64, generator=g, dtype=torch.float64, requires_grad=True)
    yv = (v - v.mean(0)) / torch.sqrt(v.var(0, unbiased=False) + 1e-5) * ga + be
    dv = torch.randn(16, 64, generator=g, dtype=torch.float64)
    yv.backward(dv)
    vd, gd, hd = (t.detach().float().cuda().requires_grad_() for t in (v, ga, be))
    rm, rv = torch.zeros(64, device='cuda'), torch.ones(64, device='cuda')
    yo = BNRows.apply(vd, gd, hd, rm, rv, 0.9, 1e-5)
    yo.backward(dv.float().cuda())
    assert rel(yo, yv.detach()) < 2e-6 and rel(vd.grad, v.grad) < 2e-5 and rel(gd.grad, ga.grad) < 2e-5 and rel(hd.grad, be.grad) < 2e-5


@pytest.mark.parametrize('cfg', [(0.2, 0.0, False), (0.3, 0.1, False), (0.2, 0.0, True), (0.0, 0.0, False)])
def test_head_loss_grads_vs_autograd(N, cfg):
    from ppvector.train.functions import HeadLoss
    margin, ls, easy = cfg
    B, D, Cc = 24, 192, 500
    g = torch.Generator().manual_seed(3)
    emb = torch.randn(B, D, generator=g, dtype=torch.float64, requires_grad=True)
    W = torch.randn(D, Cc, generator=g, dtype=torch.float64, requires_grad=True)
    labels = torch.randint(0, Cc, (B,), generator=g)
    loss = om.aam_loss(om.cosine_head(emb, W), labels, margin, 32.0, easy, ls)
    loss.backward()
    ed, Wd = emb.detach().float().cuda().requires_grad_(), W.detach().float().cuda().requires_grad_()
    lo = HeadLoss.apply(ed, Wd, labels.cuda(), margin, 32.0, ls, easy)[0]
    (lo * 1.0).backward()
    assert abs(lo.item() - loss.item()) < 1e-4 * max(1.0, abs(loss.item()))
    assert rel(ed.grad, emb.grad) < 5e-5 and rel(Wd.grad, W.grad) < 5e-5


def test_adam_matches_formula(N):
    from ppvector.optimizer.adam import Adam
    g = torch.Generator().manual_seed(9)
    ps = [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in ((7, 5), (33,), (4, 3, 2))]
    ref = [p.detach().double().cpu().clone() for p in ps]
    m = [torch.zeros_like(r) for r in ref]
    v = [torch.zeros_like(r) for r in ref]
    opt = Adam(ps, learning_rate=1e-2, weight_decay=1e-3)
    for t in range(1, 4):
        opt.clear_grad()
        grads = [torch.randn(p.shape, generator=g) for p in ps]
        for p, gr in zip(ps, grads):
            p.grad = gr.cuda()                      # as autograd leaves it after clear_grad(): the optimiser packs the tensors itself
        opt.step()
        for i, gr in enumerate(grads):
            gg = gr.double() + 1e-3 * ref[i]
            m[i] = 0.9 * m[i] + 0.1 * gg
            v[i] = 0.999 * v[i] + 0.001 * gg * gg
            ref[i] = ref[i] - 1e-2 * (m[i] / (1 - 0.9 ** t)) / (torch.sqrt(v[i] / (1 - 0.999 ** t)) + 1e-8)
    for p, r in zip(ps, ref):
        assert rel(p.detach(), r) < 1e-6


@pytest.mark.parametrize('kind', ['AdamW', 'Momentum', 'Nesterov', 'SGD'])
def test_other_optimizers_match_their_formulas(N, kind):
    """paddle.optimizer.AdamW (decoupled decay), Momentum (plain and use_nesterov) and SGD on the flat buffers, three steps against
    the update rules in float64 (optimizer/__init__.py:12-18 resolves any of them by name)."""
    from ppvector import optimizer as O
    g = torch.Generator().manual_seed(13)
    ps = [torch.nn.Parameter(torch.randn(s, generator=g).cuda()) for s in ((7, 5), (33,), (4, 3, 2))]
    ref = [p.detach().double().cpu().clone() for p in ps]
    st1, st2 = [torch.zeros_like(r) for r in ref], [torch.zeros_like(r) for r in ref]
    lr, wd = 1e-2, 1e-2
    opt = {'AdamW': lambda: O.AdamW(ps, learning_rate=lr, weight_decay=wd),
           'Momentum': lambda: O.Momentum(ps, learning_rate=lr, momentum=0.9, weight_decay=wd),
           'Nesterov': lambda: O.Momentum(ps, learning_rate=lr, momentum=0.9, use_nesterov=True, weight_decay=wd),
           'SGD': lambda: O.SGD(ps, learning_rate=lr, weight_decay=wd)}[kind]()
    for t in range(1, 4):
        opt.clear_grad()
        grads = [torch.randn(p.shape, generator=g) for p in ps]
        for p, gr in zip(ps, grads):
            p.grad = gr.cuda()
        opt.step()
        for i, gr in enumerate(grads):
            gr = gr.double()
            if kind == 'AdamW':
                st1[i] = 0.9 * st1[i] + 0.1 * gr
                st2[i] = 0.999 * st2[i] + 0.001 * gr * gr
                ref[i] = ref[i] * (1 - lr * wd) - lr * (st1[i] / (1 - 0.9 ** t)) / (torch.sqrt(st2[i] / (1 - 0.999 ** t)) + 1e-8)
            else:
                mu = 0.0 if kind == 'SGD' else 0.9
                gg = gr + wd * ref[i]
                st1[i] = mu * st1[i] + gg
                ref[i] = ref[i] - lr * (gg + mu * st1[i] if kind == 'Nesterov' else st1[i])
    for p, r in zip(ps, ref):
        assert rel(p.detach(), r) < 1e-6


def test_tdnn_training_step_vs_oracle_autograd(N):
    """Whole training step of configs/tdnn.yml's model: loss, every parameter gradient and the running statistics
    against autograd over the oracle graph (train-mode BatchNorm), then one Adam step."""
    from ppvector.models.fc import SpeakerIdentification
    from ppvector.models.tdnn import TDNN
    from ppvector.optimizer.adam import Adam
    from ppvector.train.functions import HeadLoss
    B, T, Cc = 6, 70, 40
    p = om.tdnn_params(80, seed=11)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(B, T, 80, generator=g) * 2
    labels = torch.randint(0, Cc, (B,), generator=g)
    Wh = om.head_params(192, Cc, seed=4)
    pr = {k: v.clone().double().requires_grad_(v.dtype.is_floating_point and not k.endswith(('_mean', '_variance'))) for k, v in p.items()}
    Wr = Wh.clone().double().requires_grad_()
    stats = {}
    emb_ref = om.tdnn_forward(pr, x.double(), training=True, stats_out=stats)
    loss_ref = om.aam_loss(om.cosine_head(emb_ref, Wr), labels, 0.2, 32.0, False, 0.0)
    loss_ref.backward()
    m = TDNN(80)
    m.load_state_dict(p)
    m = m.cuda().train()
    head = SpeakerIdentification(192, Cc).cuda()
    with torch.no_grad():
        head.weight.copy_(Wh.cuda())
    emb = m(x.cuda())
    assert rel(emb, emb_ref.detach()) < 2e-5
    loss = HeadLoss.apply(emb, head.weight, labels.cuda(), 0.2, 32.0, 0.0, False)[0]
    assert abs(loss.item() - loss_ref.item()) < 2e-4 * abs(loss_ref.item())
    loss.backward()
    worst = 0.0
    for k, v in m.named_parameters():
        if pr[k].grad.norm().item() < 1e-9:        # softmax over time ignores its logits' bias: the true gradient is 0
            assert v.grad.abs().max().item() < 1e-5, k
            continue
        r = rel(v.grad, pr[k].grad)
        worst = max(worst, r)
        assert r < 5e-4, (k, r)
    assert rel(head.weight.grad, Wr.grad) < 5e-4
    print(f'[tdnn train] loss {loss.item():.5f} (oracle {loss_ref.item():.5f}); worst parameter-gradient rel-L2 {worst:.2e}')
    mean1, var1 = stats['bn1.']
    assert rel(m.bn1._mean, 0.9 * p['bn1._mean'] + 0.1 * mean1) < 1e-5
    assert rel(m.bn1._variance, 0.9 * p['bn1._variance'] + 0.1 * var1) < 1e-5
    opt = Adam(list(m.parameters()) + list(head.parameters()), learning_rate=1e-3, weight_decay=1e-6)
    before = m.td_layer1.weight.detach().clone()
    opt.step()
    assert (m.td_layer1.weight.detach() - before).abs().max().item() > 0
    m.eval()
    with torch.no_grad():
        assert torch.isfinite(m(x.cuda())).all()


@pytest.mark.parametrize('mode', ['f32', 'x3'])
def test_ecapa_training_step_vs_oracle_autograd(N, mode):
    """ECAPA-TDNN (configs/ecapa_tdnn.yml) training step: loss and every parameter gradient vs autograd over the oracle graph.
    mode x3 = ppvector.set_train_x3: the same f32 step with the conv GEMMs (forward, data gradient, weight gradient) in split precision
    -- held to the SAME bounds as the exact-f32 step."""
    import ppvector
    from ppvector.models.ecapa_tdnn import EcapaTdnn
    from ppvector.train.functions import HeadLoss
    ppvector.set_train_x3(mode == 'x3')
    try:
        _ecapa_step_vs_oracle(mode)
    finally:
        ppvector.set_train_x3(False)


def _ecapa_step_vs_oracle(mode):
    from ppvector.models.ecapa_tdnn import EcapaTdnn
    from ppvector.train.functions import HeadLoss
    # batch statistics over 4 x 50 frames amplify rounding ~1e4 x (the exact-f32 step lands 7e-4 from float64 there); the split-precision
    # step (2^-17 per product against f32's 2^-24) is therefore checked on a better-conditioned batch, 12 x 120 frames
    B, T, Cc = (4, 50, 30) if mode == 'f32' else (12, 120, 30)
    # At random init this graph turns a forward error e into a gradient error of ~150 e whatever the arithmetic (measured at 12 x 120:
    # torch's own float32 autograd of the oracle graph: embeddings 1.7e-6 / whole gradient 7.9e-4 from float64; the exact-f32 engine 5.2e-6 /
    # 7.0e-4; split precision 2.6e-5 / 5.9e-3; enable_amp 1.3e-2 / 1.5e-1).  So the split-precision gradient is held RELATIVE to float32
    # autograd's own deviation (12 x), per parameter to 5e-2 (bias gradients are sums of cancelling terms), and the embeddings to 1e-4.
    e_tol, l_tol, g_tol, w_tol = (5e-5, 2e-4, 2e-3, 1e-3) if mode == 'f32' else (1e-4, 5e-4, 5e-2, 5e-3)
    p = om.ecapa_params(80, seed=21)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, T, 80, generator=g) * 2
    labels = torch.randint(0, Cc, (B,), generator=g)
    Wh = om.head_params(192, Cc, seed=5)
    pr = {k: v.clone().double().requires_grad_(not k.endswith(('_mean', '_variance'))) for k, v in p.items()}
    Wr = Wh.clone().double().requires_grad_()
    emb_ref = om.ecapa_forward(pr, x.double(), training=True)
    loss_ref = om.aam_loss(om.cosine_head(emb_ref, Wr), labels, 0.2, 32.0, False, 0.0)
    loss_ref.backward()
    m = EcapaTdnn(80)
    m.load_state_dict(p)
    m = m.cuda().train()
    Wd = Wh.cuda().requires_grad_()
    emb = m(x.cuda())
    assert rel(emb, emb_ref.detach()) < e_tol
    loss = HeadLoss.apply(emb, Wd, labels.cuda(), 0.2, 32.0, 0.0, False)[0]
    assert abs(loss.item() - loss_ref.item()) < l_tol * abs(loss_ref.item())
    loss.backward()
    worst, wk, num, den = 0.0, '', 0.0, 0.0
    for k, v in m.named_parameters():
        num += (v.grad.double().cpu() - pr[k].grad).pow(2).sum().item()
        den += pr[k].grad.pow(2).sum().item()
        if pr[k].grad.norm().item() < 1e-9:
            assert v.grad.abs().max().item() < 1e-5, k
            continue
        r = rel(v.grad, pr[k].grad)
        if r > worst:
            worst, wk = r, k
        assert r < g_tol, (k, r)
    assert rel(Wd.grad, Wr.grad) < w_tol
    whole = (num / den) ** 0.5
    whole_tol = 2e-3
    if mode != 'f32':                                     # the yardstick: float32 autograd of the same oracle graph against its float64 run
        p32 = {k: v.clone().float().requires_grad_(not k.endswith(('_mean', '_variance'))) for k, v in p.items()}
        W32 = Wh.clone().float().requires_grad_()
        om.aam_loss(om.cosine_head(om.ecapa_forward(p32, x.float(), training=True), W32), labels, 0.2, 32.0, False, 0.0).backward()
        n32 = sum((p32[k].grad.double() - pr[k].grad).pow(2).sum().item() for k, _ in m.named_parameters())
        whole_tol = 12 * (n32 / den) ** 0.5
    print(f'[ecapa train {mode}] loss {loss.item():.5f} (oracle {loss_ref.item():.5f}); worst parameter-gradient rel-L2 {worst:.2e} ({wk}); whole gradient {whole:.2e} (bound {whole_tol:.2e})')
    assert whole < whole_tol, (whole, whole_tol)
    m.eval()


def test_reference_shaped_train_loop_overfits_a_batch(N):
    """The reference's loop body (trainer.py:206-274) over its own object graph -- nn.Sequential(backbone, classifier),
    build_loss, build_optimizer, build_lr_scheduler, MarginScheduler -- on one fixed batch: the first loss equals the
    oracle's, and a few Adam steps drive it down."""
    from ppvector.loss import build_loss
    from ppvector.models import build_model
    from ppvector.models.fc import SpeakerIdentification
    from ppvector.optimizer import MarginScheduler, build_lr_scheduler, build_optimizer
    from ppvector.train.step import TrainStep
    from ppvector.utils.utils import dict_to_object
    configs = dict_to_object(dict(
        model_conf=dict(model='TDNN', model_args=dict(embd_dim=192, pooling_type='ASP'),
                        classifier=dict(classifier_type='Cosine', num_speakers=12, num_blocks=0)),
        loss_conf=dict(loss='AAMLoss', loss_args=dict(margin=0.2, scale=32, easy_margin=False, label_smoothing=0.0)),
        optimizer_conf=dict(optimizer='Adam', optimizer_args=dict(weight_decay=1e-6), scheduler='WarmupCosineSchedulerLR',
                            scheduler_args=dict(learning_rate=2e-3, min_lr=1e-5, warmup_epoch=1)),
        train_conf=dict(max_epoch=4)))
    torch.manual_seed(1000)
    backbone = build_model(input_size=80, configs=configs)
    p = om.tdnn_params(80, seed=3)
    backbone.load_state_dict(p)
    classifier = SpeakerIdentification(input_dim=backbone.embd_dim, **{k: v for k, v in configs.model_conf.classifier.items()})
    model = torch.nn.Sequential(backbone, classifier).cuda()
    criterion = build_loss(configs)
    spe = 5
    scheduler = build_lr_scheduler(step_per_epoch=spe, configs=configs)
    optimizer = build_optimizer(parameters=model.parameters(), learning_rate=scheduler, configs=configs)
    margin = MarginScheduler(criterion, increase_start_epoch=1, fix_epoch=3, step_per_epoch=spe, initial_margin=0.0, final_margin=0.3)
    step = TrainStep(model, criterion, optimizer, scheduler, margin)
    g = torch.Generator().manual_seed(8)
    x = (torch.randn(8, 64, 80, generator=g) * 2).cuda()
    y = torch.randint(0, 12, (8,), generator=g).cuda()
    with torch.no_grad():
        W0 = classifier.weight.detach().cpu().double()
    ref0 = om.aam_loss(om.cosine_head(om.tdnn_forward({k: v.double() for k, v in p.items()}, x.cpu().double(), training=True), W0),
                       y.cpu(), 0.0, 32.0, False, 0.0).item()
    losses = []
    for _ in range(12):
        loss, acc = step(x, y)
        losses.append(loss.item())
    assert abs(losses[0] - ref0) < 2e-4 * abs(ref0), (losses[0], ref0)
    assert scheduler.get_lr() > 0 and criterion.margin > 0          # both schedules moved
    assert losses[1] == losses[0]                                   # warm-up: the table's lr[0] is 0 (scheduler.py:6-40)
    print('[train loop] losses', ' '.join(f'{v:.3f}' for v in losses), ' final acc', float(acc))
    assert losses[4] < losses[0] - 0.05                             # before the margin ramp starts (epoch 1 = step 5)


@pytest.mark.parametrize('case', [
    # B, T, F, Cin, Cout, k, stride, relu
    (2, 13, 10, 8, 16, 3, 1, True),
    (2, 13, 10, 16, 8, 3, 2, True),      # odd T, even F: zero-insertion data gradient
    (3, 8, 9, 8, 12, 1, 2, False),       # strided 1x1 (the bottleneck's downsample branch)
    (2, 6, 7, 12, 8, 1, 1, True),
])
def test_conv2d_block_grads_vs_autograd(N, case):
    """Conv2D -> BatchNorm2D(batch statistics) -> [ReLU] over (B, T, F, C) positions: output and all gradients."""
    from ppvector.train.functions import Conv2dBlock
    B, T, Fq, Cin, Cout, k, s, relu = case
    g = torch.Generator().manual_seed(31 + k + s)
    x = torch.randn(B, Cin, Fq, T, generator=g, dtype=torch.float64, requires_grad=True)          # the reference's (B, C, F, T)
    w = (torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64) / (Cin * k * k) ** 0.5).requires_grad_()
    b = torch.randn(Cout, generator=g, dtype=torch.float64, requires_grad=True)
    ga = (torch.rand(Cout, generator=g, dtype=torch.float64) + 0.5).requires_grad_()
    be = torch.randn(Cout, generator=g, dtype=torch.float64, requires_grad=True)
    z = F.conv2d(x, w, b, stride=s, padding=(k - 1) // 2)
    mean, var = z.mean(dim=(0, 2, 3)), z.var(dim=(0, 2, 3), unbiased=False)
    y = (z - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + 1e-5) * ga[None, :, None, None] + be[None, :, None, None]
    if relu:
        y = F.relu(y)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    to2d = lambda t: t.permute(0, 3, 2, 1).reshape(-1, t.shape[1])                                # (B, C, F, T) -> (B*T*F, C)
    xd = to2d(x.detach()).float().cuda().requires_grad_()
    wd, bd = w.detach().float().cuda().requires_grad_(), b.detach().float().cuda().requires_grad_()
    gd, hd = ga.detach().float().cuda().requires_grad_(), be.detach().float().cuda().requires_grad_()
    rm, rv = torch.zeros(Cout, device='cuda'), torch.ones(Cout, device='cuda')
    out = Conv2dBlock.apply(xd, wd, bd, gd, hd, rm, rv, dict(B=B, T=T, F=Fq, stride=s, relu=relu))
    out.backward(to2d(dy).float().cuda())
    assert rel(out, to2d(y.detach())) < 2e-6
    for name, got, ref in (('dx', xd.grad, to2d(x.grad)), ('dW', wd.grad, w.grad), ('dbias', bd.grad, b.grad),
                           ('dgamma', gd.grad, ga.grad), ('dbeta', hd.grad, be.grad)):
        if ref.norm().item() < 1e-9:
            assert got.abs().max().item() < 1e-5, name        # a bias before BatchNorm has zero gradient
            continue
        assert rel(got, ref) < 3e-5, (name, rel(got, ref))


@pytest.mark.parametrize('case', [(2, 13, 10, 8, 16, 3, 1), (3, 9, 8, 16, 8, 1, 2)])
def test_conv2d_block_hardtanh_folded_vs_autograd(N, case):
    """Conv2D -> BatchNorm2D -> Hardtanh(0, 20) (ERes2Net's ReLU, eres2net.py:14-22): the clamp leaves with the BatchNorm apply pass and
    its mask 0 < y < 20 is re-evaluated inside the two BatchNorm-backward passes (mask_hi of vp_col_sums_masked_f32 /
    vp_bn_relu_bwd_masked_f32) -- no activation pass either way.  gamma is large enough that BOTH clamps are active."""
    from ppvector.train.functions import Conv2dBlock
    B, T, Fq, Cin, Cout, k, s = case
    g = torch.Generator().manual_seed(77 + k + s)
    x = torch.randn(B, Cin, Fq, T, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Cout, Cin, k, k, generator=g, dtype=torch.float64) / (Cin * k * k) ** 0.5).requires_grad_()
    b = torch.randn(Cout, generator=g, dtype=torch.float64, requires_grad=True)
    ga = ((torch.rand(Cout, generator=g, dtype=torch.float64) + 0.5) * 14.0).requires_grad_()
    be = (torch.randn(Cout, generator=g, dtype=torch.float64) * 3 + 6).requires_grad_()
    z = F.conv2d(x, w, b, stride=s, padding=(k - 1) // 2)
    mean, var = z.mean(dim=(0, 2, 3)), z.var(dim=(0, 2, 3), unbiased=False)
    pre = (z - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + 1e-5) * ga[None, :, None, None] + be[None, :, None, None]
    y = F.hardtanh(pre, 0.0, 20.0)
    lo, hi = (pre <= 0).double().mean().item(), (pre >= 20).double().mean().item()
    assert lo > 0.05 and hi > 0.05, (lo, hi)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    to2d = lambda t: t.permute(0, 3, 2, 1).reshape(-1, t.shape[1])
    res = {}
    for unfolded in (False, True):
        if unfolded:
            os.environ['VPMI_BN_RELU_UNFOLDED'] = '1'
        try:
            xd = to2d(x.detach()).float().cuda().requires_grad_()
            wd, bd = w.detach().float().cuda().requires_grad_(), b.detach().float().cuda().requires_grad_()
            gd, hd = ga.detach().float().cuda().requires_grad_(), be.detach().float().cuda().requires_grad_()
            rm, rv = torch.zeros(Cout, device='cuda'), torch.ones(Cout, device='cuda')
            out = Conv2dBlock.apply(xd, wd, bd, gd, hd, rm, rv, dict(B=B, T=T, F=Fq, stride=s, act='hardtanh'))
            out.backward(to2d(dy).float().cuda())
            res[unfolded] = (out.detach(), xd.grad, wd.grad, gd.grad, hd.grad)
        finally:
            os.environ.pop('VPMI_BN_RELU_UNFOLDED', None)
    ref = (to2d(y.detach()), to2d(x.grad), w.grad, ga.grad, be.grad)
    for name, got, per_op, want in zip(('y', 'dx', 'dW', 'dgamma', 'dbeta'), res[False], res[True], ref):
        e, e0 = rel(got, want), rel(per_op, want)
        print(f'[conv2d hardtanh {case}] {name:7s} folded vs float64 {e:.2e}   separate activation passes vs float64 {e0:.2e}   clamped at 0 / 20: {lo:.2f} / {hi:.2f}')
        assert e < 3e-5, (name, e)


def test_resnetse_training_step_vs_oracle_autograd(N):
    """ResNetSE (configs/resnet_se.yml architecture, one bottleneck per stage to keep the float64 oracle quick)."""
    from oracle import resnet_se as orse
    from ppvector.models.resnet_se import ResNetSE
    from ppvector.train.functions import HeadLoss
    B, T, Fdim, Cc = 3, 26, 16, 10
    layers = [1, 1, 1, 1]
    p = orse.resnetse_params(Fdim, 192, layers=layers, seed=9)
    g = torch.Generator().manual_seed(12)
    x = torch.randn(B, T, Fdim, generator=g) * 2
    labels = torch.randint(0, Cc, (B,), generator=g)
    Wh = om.head_params(192, Cc, seed=2)
    pr = {k: v.clone().double().requires_grad_(not k.endswith(('_mean', '_variance'))) for k, v in p.items()}
    Wr = Wh.clone().double().requires_grad_()
    emb_ref = orse.resnetse_forward(pr, x.double(), layers=layers, training=True)
    loss_ref = om.aam_loss(om.cosine_head(emb_ref, Wr), labels, 0.2, 32.0, False, 0.0)
    loss_ref.backward()
    m = ResNetSE(Fdim, layers=layers)
    m.load_state_dict(p)
    m = m.cuda().train()
    Wd = Wh.cuda().requires_grad_()
    emb = m(x.cuda())
    assert rel(emb, emb_ref.detach()) < 1e-4
    loss = HeadLoss.apply(emb, Wd, labels.cuda(), 0.2, 32.0, 0.0, False)[0]
    assert abs(loss.item() - loss_ref.item()) < 5e-4 * abs(loss_ref.item())
    loss.backward()
    worst, wk = 0.0, ''
    for k, v in m.named_parameters():
        if pr[k].grad.norm().item() < 1e-9:
            assert v.grad.abs().max().item() < 1e-4, k
            continue
        r = rel(v.grad, pr[k].grad)
        if r > worst:
            worst, wk = r, k
        assert r < 5e-3, (k, r)
    print(f'[resnetse train] loss {loss.item():.5f} (oracle {loss_ref.item():.5f}); worst parameter-gradient rel-L2 {worst:.2e} ({wk})')
    m.eval()


@pytest.mark.parametrize('two_emb', [False, True])
def test_eres2net_training_step_vs_oracle_autograd(N, two_emb):
    """ERes2Net (configs/eres2net.yml architecture, one block per stage): both block kinds, the AFFs, the stride-2 fusion
    convs and TSTP, against autograd over the oracle graph; two_emb: with the second embedding layer (eres2net.py:255-260)."""
    from oracle import eres2net as oer
    from ppvector.models.eres2net import ERes2Net
    from ppvector.train.functions import HeadLoss
    B, T, Fdim, Cc = (8 if two_emb else 3), 20, 16, 10          # (BatchNorm1D over the batch: more than 3 rows)
    nb = (1, 1, 1, 1)
    p = oer.eres2net_params(Fdim, 192, num_blocks=nb, seed=13, two_emb_layer=two_emb)
    g = torch.Generator().manual_seed(14)
    x = torch.randn(B, T, Fdim, generator=g) * 2
    labels = torch.randint(0, Cc, (B,), generator=g)
    Wh = om.head_params(192, Cc, seed=3)
    pr = {k: v.clone().double().requires_grad_(not k.endswith(('_mean', '_variance'))) for k, v in p.items()}
    Wr = Wh.clone().double().requires_grad_()
    emb_ref = oer.eres2net_forward(pr, x.double(), num_blocks=nb, training=True)
    loss_ref = om.aam_loss(om.cosine_head(emb_ref, Wr), labels, 0.2, 32.0, False, 0.0)
    loss_ref.backward()
    m = ERes2Net(Fdim, num_blocks=list(nb), two_emb_layer=two_emb)
    m.load_state_dict(p)
    m = m.cuda().train()
    Wd = Wh.cuda().requires_grad_()
    emb = m(x.cuda())
    assert rel(emb, emb_ref.detach()) < 1e-4
    loss = HeadLoss.apply(emb, Wd, labels.cuda(), 0.2, 32.0, 0.0, False)[0]
    assert abs(loss.item() - loss_ref.item()) < 5e-4 * abs(loss_ref.item())
    loss.backward()
    worst, wk = 0.0, ''
    for k, v in m.named_parameters():
        if pr[k].grad is None or pr[k].grad.norm().item() < 1e-9:
            assert v.grad is None or v.grad.abs().max().item() < 1e-4, k
            continue
        r = rel(v.grad, pr[k].grad)
        if r > worst:
            worst, wk = r, k
        assert r < 5e-3, (k, r)
    if two_emb:                                                     # and the eval path of the same model: folded BatchNorm + dense
        m.eval()
        p_run = {k: v.detach().double().cpu() for k, v in m.state_dict().items()}
        with torch.no_grad():
            e_eval = m(x.cuda())
        assert rel(e_eval, oer.eres2net_forward(p_run, x.double(), num_blocks=nb)) < 1e-4
    print(f'[eres2net train] loss {loss.item():.5f} (oracle {loss_ref.item():.5f}); worst parameter-gradient rel-L2 {worst:.2e} ({wk})')
    m.eval()


def test_eres2net_large_training_step_vs_oracle_autograd(N):
    """BASELINE configs[4]'s backbone as a TRAINING step: ERes2Net-large (55.2 M parameters at F = 80: m_channels 64, mul_channel 2,
    expansion 4, base_width 24, scale 3, all 16 blocks) on a reduced map (T = 24, F = 16) -- forward, AAM head over 1000 classes and every
    parameter gradient against float64 autograd over the oracle graph.  The widths of this variant (chunks of 24 / 48 / 96 / 192 channels,
    three per block) take code paths the 6.6 M base model never reaches.
    At random init this 16-block graph is ILL-CONDITIONED in f32: torch's own float32 autograd over the same oracle graph lands 2 - 5 % from
    its float64 run on most parameter gradients (Hardtanh / ReLU kinks and batch statistics over maps that shrink to 3 x 2 positions;
    profiles/r06_eres2net_large_gradient_conditioning.log).  So each gradient is held to the larger of 5e-3 and 3 x what float32 autograd
    itself achieves (per tensor, or its median over the tensors) -- a wrong
    kernel (a missing tap, a transposed chunk) is off by O(1), not by a factor on the rounding noise."""
    from oracle import eres2net as oer
    from ppvector.models.eres2net import ERes2Net
    from ppvector.train.functions import HeadLoss
    LARGE = dict(m_channels=64, mul_channel=2, expansion=4, base_width=24, scale=3)
    FW = dict(m_channels=64, expansion=4, base_width=24, scale=3)
    B, T, Fdim, Cc = 4, 24, 16, 1000
    p = oer.eres2net_params(Fdim, 192, seed=23, **LARGE)
    g = torch.Generator().manual_seed(24)
    x = torch.randn(B, T, Fdim, generator=g) * 2
    labels = torch.randint(0, Cc, (B,), generator=g)
    Wh = om.head_params(192, Cc, seed=5)
    ref = {}
    for dt in (torch.float64, torch.float32):
        pr = {k: v.clone().to(dt).requires_grad_(not k.endswith(('_mean', '_variance'))) for k, v in p.items()}
        Wr = Wh.clone().to(dt).requires_grad_()
        e = oer.eres2net_forward(pr, x.to(dt), training=True, **FW)
        lo = om.aam_loss(om.cosine_head(e, Wr), labels, 0.2, 32.0, False, 0.0)
        lo.backward()
        ref[dt] = (e.detach(), lo.detach(), Wr.grad, {k: v.grad for k, v in pr.items() if v.grad is not None})
    emb_ref, loss_ref, gW, gp = ref[torch.float64]
    emb32, _, gW32, gp32 = ref[torch.float32]
    m = ERes2Net(Fdim, embd_dim=192, **LARGE)
    m.load_state_dict(p)
    m = m.cuda().train()
    Wd = Wh.cuda().requires_grad_()
    emb = m(x.cuda())
    assert rel(emb, emb_ref) < max(1e-4, 3 * rel(emb32, emb_ref))
    loss = HeadLoss.apply(emb, Wd, labels.cuda(), 0.2, 32.0, 0.0, False)[0]
    assert abs(loss.item() - loss_ref.item()) < 5e-4 * abs(loss_ref.item())
    loss.backward()
    assert rel(Wd.grad, gW) < max(1e-3, 3 * rel(gW32, gW))
    live = [k for k, _ in m.named_parameters() if k in gp and gp[k].norm().item() >= 1e-9]
    r32s = sorted(rel(gp32[k], gp[k]) for k in live)
    noise = r32s[len(r32s) // 2]                       # what float32 autograd of the oracle graph typically achieves here
    worst, wk, n, ratio = 0.0, '', 0, []
    for k, v in m.named_parameters():
        if k not in live:
            assert v.grad is None or v.grad.abs().max().item() < 1e-4, k
            continue
        r, r32 = rel(v.grad, gp[k]), rel(gp32[k], gp[k])
        n += 1
        ratio.append(r / max(r32, 1e-12))
        if r > worst:
            worst, wk = r, k
        assert r < max(5e-3, 3 * max(r32, noise)), (k, r, r32, noise)
    ratio.sort()
    print(f'[eres2net-large train] loss {loss.item():.5f} (oracle {loss_ref.item():.5f}); {n} parameter tensors, worst gradient rel-L2 vs float64 {worst:.2e} ({wk}); '
          f'engine error / float32-autograd error: median {ratio[len(ratio) // 2]:.2f}, max {ratio[-1]:.2f}')
    assert ratio[len(ratio) // 2] < 2.5
    m.eval()


def test_eres2netv2_training_step_vs_oracle_autograd(N):
    """ERes2NetV2 (base_width 26: chunk widths 13 / 26 / 52 / 104 -- the first two run on zero-padded chunks; one bottom-up
    fusion) against autograd over the oracle graph; running statistics land in the reference-shaped buffers."""
    from oracle import eres2net as oer
    from ppvector.models.eres2net import ERes2NetV2
    from ppvector.train.functions import HeadLoss
    B, T, Fdim, Cc = 3, 20, 16, 10
    nb = (1, 2, 1, 1)
    p = oer.eres2net_params(Fdim, 192, num_blocks=nb, base_width=26, seed=17, v2=True)
    g = torch.Generator().manual_seed(15)
    x = torch.randn(B, T, Fdim, generator=g) * 2
    labels = torch.randint(0, Cc, (B,), generator=g)
    Wh = om.head_params(192, Cc, seed=3)
    pr = {k: v.clone().double().requires_grad_(not k.endswith(('_mean', '_variance'))) for k, v in p.items()}
    Wr = Wh.clone().double().requires_grad_()
    emb_ref = oer.eres2netv2_forward(pr, x.double(), num_blocks=nb, training=True)
    loss_ref = om.aam_loss(om.cosine_head(emb_ref, Wr), labels, 0.2, 32.0, False, 0.0)
    loss_ref.backward()
    m = ERes2NetV2(Fdim, num_blocks=list(nb))
    m.load_state_dict(p)
    m = m.cuda().train()
    assert m.layer1[0].width == 13 and m.layer2[0].width == 26
    Wd = Wh.cuda().requires_grad_()
    emb = m(x.cuda())
    assert rel(emb, emb_ref.detach()) < 1e-4
    loss = HeadLoss.apply(emb, Wd, labels.cuda(), 0.2, 32.0, 0.0, False)[0]
    assert abs(loss.item() - loss_ref.item()) < 5e-4 * abs(loss_ref.item())
    loss.backward()
    worst, wk = 0.0, ''
    for k, v in m.named_parameters():
        if pr[k].grad is None or pr[k].grad.norm().item() < 1e-9:
            assert v.grad is None or v.grad.abs().max().item() < 1e-4, k
            continue
        r = rel(v.grad, pr[k].grad)
        if r > worst:
            worst, wk = r, k
        assert r < 5e-3, (k, r)
    # running statistics of a padded-chunk BatchNorm: Paddle momentum 0.9 on the batch mean of the pre-BN conv output
    xin = torch.nn.functional.relu(oer._bn(oer._c2d(x.transpose(1, 2).unsqueeze(1), p, 'conv1.', padding=1), p, 'bn1.', True))
    z = oer._c2d(xin, p, 'layer1.0.conv1.')
    want = 0.9 * p['layer1.0.bn1._mean'] + 0.1 * z.mean(dim=(0, 2, 3))
    assert rel(m.layer1[0].bn1._mean, want) < 1e-5
    print(f'[eres2netv2 train] loss {loss.item():.5f} (oracle {loss_ref.item():.5f}); worst parameter-gradient rel-L2 {worst:.2e} ({wk})')
    m.eval()


@pytest.mark.parametrize('B,gtol', [(3, 3e-2), (16, 2e-2)])
def test_campplus_training_step_vs_oracle_autograd(N, B, gtol):
    """CAM++ (configs/cam++.yml: embd 192): FCM with stride on the frequency axis, the stride-2 TDNN, 52 CAM dense layers with
    two context segments (115 frames after the stride), transit layers, unbiased statistics pooling.
    Every other backbone matches float64 to <= 7e-4; CAM++ sits at 1.8e-2 (B = 3) / 8e-3 (B = 16).  That is the f32 conditioning
    of this 60-layer train-mode graph, not the backward kernels: PyTorch-CPU f32 autograd over the ORACLE graph deviates from
    its own float64 run by 1.5e-2 (B = 3) / 1.3e-2 (B = 16) worst parameter (head.conv1.weight: 7.2e-3 / 6.7e-3).  The test
    therefore also runs that f32 CPU autograd and requires the HIP engine to be no further from float64 than 2.5 x it."""
    from oracle import campplus as oc
    from ppvector.models.campplus import CAMPPlus
    from ppvector.train.functions import HeadLoss
    T, Cc = 230, 8
    p = oc.campplus_params(80, 192, seed=17)
    g = torch.Generator().manual_seed(18)
    x = torch.randn(B, T, 80, generator=g) * 2
    labels = torch.randint(0, Cc, (B,), generator=g)
    Wh = om.head_params(192, Cc, seed=6)
    pr = {k: v.clone().double().requires_grad_(not k.endswith(('_mean', '_variance'))) for k, v in p.items()}
    Wr = Wh.clone().double().requires_grad_()
    emb_ref = oc.campplus_forward(pr, x.double(), training=True)
    loss_ref = om.aam_loss(om.cosine_head(emb_ref, Wr), labels, 0.2, 32.0, False, 0.0)
    loss_ref.backward()
    m = CAMPPlus(80, embd_dim=192)
    m.load_state_dict(p)
    m = m.cuda().train()
    Wd = Wh.cuda().requires_grad_()
    emb = m(x.cuda())
    # the closing BatchNorm over a batch of 3 divides by a tiny batch variance: f32-vs-f64 noise is amplified there
    assert rel(emb, emb_ref.detach()) < 2e-3
    loss = HeadLoss.apply(emb, Wd, labels.cuda(), 0.2, 32.0, 0.0, False)[0]
    assert abs(loss.item() - loss_ref.item()) < 2e-3 * abs(loss_ref.item())
    loss.backward()
    worst, wk = 0.0, ''
    gscale = max(v.grad.abs().max().item() for v in pr.values() if v.grad is not None)
    for k, v in m.named_parameters():
        if pr[k].grad is None or pr[k].grad.norm().item() < 1e-9:        # a bias in front of a BatchNorm: true gradient 0
            assert v.grad is None or v.grad.abs().max().item() < 1e-3 * gscale, k
            continue
        r = rel(v.grad, pr[k].grad)
        if r > worst:
            worst, wk = r, k
        assert r < gtol, (k, r)
    p32 = {k: v.clone().float().requires_grad_(not k.endswith(('_mean', '_variance'))) for k, v in p.items()}
    W32 = Wh.clone().float().requires_grad_()
    om.aam_loss(om.cosine_head(oc.campplus_forward(p32, x.float(), training=True), W32), labels, 0.2, 32.0, False, 0.0).backward()
    w32 = max(rel(p32[k].grad, pr[k].grad) for k in p32 if pr[k].grad is not None and pr[k].grad.norm().item() >= 1e-9)
    print(f'[cam++ train B={B}] PyTorch-CPU f32 autograd of the oracle graph vs float64: worst rel-L2 {w32:.2e}')
    assert worst < max(2.5 * w32, 1e-3), (worst, w32)
    print(f'[cam++ train B={B}] loss {loss.item():.5f} (oracle {loss_ref.item():.5f}); worst parameter-gradient rel-L2 {worst:.2e} ({wk})')
    m.eval()


@pytest.mark.parametrize('case', [
    # B, T, C, O, k, dil, seg_len
    (5, 115, 128, 32, 3, 1, 100),         # the bench geometry of a CAM dense layer: two segments, the second of 15 frames
    (3, 64, 128, 32, 3, 2, 100),          # one segment
    (2, 301, 64, 32, 3, 2, 100),          # four segments, the last of one frame
    (4, 50, 192, 16, 5, 1, 20),           # C not a power of two, a 5-tap local conv, short segments
])
def test_cam_layer_as_one_tape_entry_vs_float64(N, case):
    """CamLayerFn (local conv + vp_cam_gate_fwd_f32 / vp_cam_gate_bwd_f32 / vp_cam_gate_wgrad_f32) against float64 autograd over
    CAMLayer.forward as the reference writes it (campplus.py:88-106), with the output gradient handed over as a column slice of a wider
    tensor (what the DenseNet concatenation's backward hands it); and against the per-op tape it replaces."""
    from ppvector.train.functions import CamLayerFn, Conv2dBlock, ConvBlock, SegCtx, SegScale
    B, T, Cc, O, k, dil, seg = case
    H = Cc // 2
    g = torch.Generator().manual_seed(31 + T)
    h = torch.randn(B, T, Cc, generator=g, dtype=torch.float64, requires_grad=True)
    mk = lambda *sh, s=1.0: (torch.randn(*sh, generator=g, dtype=torch.float64) * s).requires_grad_()
    wl, bl = mk(O, Cc, k, s=(Cc * k) ** -0.5), mk(O, s=0.1)
    w1, b1 = mk(H, Cc, 1, s=Cc ** -0.5 * 3), mk(H, s=0.3)
    w2, b2 = mk(O, H, 1, s=H ** -0.5 * 3), mk(O, s=0.3)
    x = h.transpose(1, 2)
    y = F.conv1d(x, wl, bl, dilation=dil, padding=(k - 1) // 2 * dil)
    segm = F.avg_pool1d(x, kernel_size=seg, stride=seg, ceil_mode=True)
    segm = segm.unsqueeze(-1).expand(*segm.shape, seg).reshape(B, Cc, -1)[..., :T]
    ctx = x.mean(-1, keepdim=True) + segm
    out = (y * torch.sigmoid(F.conv1d(F.relu(F.conv1d(ctx, w1, b1)), w2, b2))).transpose(1, 2)
    gw = torch.randn(B * T, O + 24, generator=g, dtype=torch.float64)
    out.backward(gw[:, 8:8 + O].reshape(B, T, O))
    ref = [out.detach().reshape(B * T, O), h.grad.reshape(B * T, Cc), wl.grad, bl.grad, w1.grad, b1.grad, w2.grad, b2.grad]

    def run(fused):
        leaves = [t.detach().float().cuda().requires_grad_() for t in (h.reshape(B * T, Cc), wl, bl, w1, b1, w2, b2)]
        hd = leaves[0]
        if fused:
            assert CamLayerFn.usable(hd, leaves[1], leaves[3], leaves[5], T, seg)
            o = CamLayerFn.apply(*leaves, dict(B=B, T=T, seg_len=seg, dilation=dil))
        else:
            yy = Conv2dBlock.apply(hd, leaves[1].unsqueeze(2), leaves[2], None, None, None, None, dict(B=B, T=T, F=1, dilation=dil, act=None))
            nseg = (T + seg - 1) // seg
            c = SegCtx.apply(hd, B, T, seg)
            c = ConvBlock.apply(c, leaves[3], leaves[4], None, None, None, None, None, dict(B=B * nseg, T=1, relu=True))
            mm = ConvBlock.apply(c, leaves[5], leaves[6], None, None, None, None, None, dict(B=B * nseg, T=1, sigmoid=True))
            o = SegScale.apply(yy, mm, B, T, seg)
        o.backward(gw.float().cuda()[:, 8:8 + O])
        return [o.detach()] + [t.grad for t in leaves]

    got, per_op = run(True), run(False)
    names = ['out', 'd h', 'd W_local', 'd b_local', 'd W1', 'd b1', 'd W2', 'd b2']
    for nm, a, b, r in zip(names, got, per_op, ref):
        ea, eb = rel(a.reshape(r.shape), r), rel(b.reshape(r.shape), r)
        print(f'[cam layer {case}] {nm:10s} one tape entry vs float64 {ea:.2e}   per-op tape vs float64 {eb:.2e}')
        assert ea < 1e-4, (nm, ea)
        assert ea < max(3 * eb, 5e-6), (nm, ea, eb)


def test_cam_dense_block_on_one_buffer_vs_concatenations(N):
    """CamDenseBlockFn (a CAMDenseTDNNBlock on one preallocated buffer, one gradient buffer backward) against the per-layer tape with
    torch.cat (VPMI_CAM_BLOCK_UNFUSED=1): the forward runs the same kernels on the same values (embeddings equal), the backward adds
    the layers' input gradients in a different order (f32 rounding only)."""
    from oracle import campplus as oc
    from ppvector.models.campplus import CAMPPlus
    from ppvector.train.functions import HeadLoss
    B, T, Cc = 6, 230, 8
    p = oc.campplus_params(80, 192, seed=23)
    g = torch.Generator().manual_seed(24)
    x = (torch.randn(B, T, 80, generator=g) * 2).cuda()
    labels = torch.randint(0, Cc, (B,), generator=g).cuda()
    Wh = om.head_params(192, Cc, seed=6)
    res = {}
    for unfused in (False, True):
        if unfused:
            os.environ['VPMI_CAM_BLOCK_UNFUSED'] = '1'
        try:
            m = CAMPPlus(80, embd_dim=192)
            m.load_state_dict(p)
            m = m.cuda().train()
            Wd = Wh.clone().cuda().requires_grad_()
            emb = m(x)
            loss = HeadLoss.apply(emb, Wd, labels, 0.2, 32.0, 0.0, False)[0]
            loss.backward()
            res[unfused] = (emb.detach(), {k: v.grad.clone() for k, v in m.named_parameters()}, {k: v.clone() for k, v in m.named_buffers()})
            m.eval()
        finally:
            os.environ.pop('VPMI_CAM_BLOCK_UNFUSED', None)
    assert rel(res[False][0], res[True][0]) < 1e-6
    for k, v in res[True][2].items():                                  # running statistics: the same finalize launches
        assert rel(res[False][2][k].float(), v.float()) < 1e-6, k
    gscale = max(v.abs().max().item() for v in res[True][1].values())
    num = den = 0.0
    worst, wk = 0.0, ''
    for k, ref in res[True][1].items():
        got = res[False][1][k]
        if ref.norm().item() < 1e-6 * gscale * ref.numel() ** 0.5:     # a bias in front of a BatchNorm: rounding noise on both sides
            continue
        num += (got.double() - ref.double()).pow(2).sum().item()
        den += ref.double().pow(2).sum().item()
        r = rel(got, ref)
        if r > worst:
            worst, wk = r, k
    print(f'[cam dense block] one buffer vs torch.cat per layer: embeddings {rel(res[False][0], res[True][0]):.1e}, whole gradient rel-L2 '
          f'{(num / den) ** 0.5:.2e}, worst tensor {worst:.2e} ({wk})')
    assert (num / den) ** 0.5 < 1e-3 and worst < 2e-2


def test_placeholder_read_as_data_is_loud(N):
    """An activation that exists as bf16 only travels on the tape as an f32 placeholder that owns ONE element, its values on the
    `_vp_bf16` attribute (functions._placeholder).  A consumer that forgets _f32c / _only16 and reads the placeholder itself must not
    train on silent zeros: the element is NaN, so any arithmetic on it poisons the loss at the first step; the sanctioned readers get
    the values."""
    from ppvector.train.functions import _f32c, _only16, _placeholder
    dev = torch.device('cuda', 0)
    vals = torch.arange(32, dtype=torch.float32, device=dev).reshape(4, 8).to(torch.bfloat16)
    p = _placeholder((4, 8), dev)
    p._vp_bf16, p._vp_bf16_only = vals, True
    assert p.shape == (4, 8) and p.untyped_storage().nbytes() == 4
    assert bool(torch.isnan(p).all()) and bool(torch.isnan((p * 0.0).sum()))           # a raw read is NaN, even scaled by zero
    assert _only16(p) is vals
    assert torch.equal(_f32c(p), vals.float())


def test_eval_engine_follows_training_updates(N):
    """The packed eval engine (and its HIP graph) must not survive a training step: Adam and the BatchNorm running statistics
    are written through raw pointers, which torch's version counters never see (PPVectorTrainer.train(do_eval=True) evaluates
    after every epoch, trainer.py:340-360).  eval -> train steps -> eval: the second embedding must be the oracle's on the
    UPDATED state dict, for the f32 engine, the bf16 engine and the captured graph."""
    import ppvector
    from ppvector.models.tdnn import TDNN
    from ppvector.optimizer.adam import Adam
    from ppvector.train.functions import HeadLoss
    p = om.tdnn_params(80, seed=31)
    m = TDNN(80)
    m.load_state_dict(p)
    m = m.cuda()
    g = torch.Generator().manual_seed(2)
    x = (torch.randn(6, 70, 80, generator=g) * 2).cuda()
    y = torch.randint(0, 9, (6,), generator=g).cuda()
    W = om.head_params(192, 9, seed=4).cuda().requires_grad_()
    opt = Adam(list(m.parameters()) + [W], learning_rate=5e-3)

    def oracle_emb():
        sd = {k: v.detach().cpu().double() for k, v in m.state_dict().items()}
        return om.tdnn_forward(sd, x.cpu().double())

    m.eval()
    ppvector.set_graph_mode(True)
    try:
        e0 = {dt: m.engine(dt).forward(x).clone() for dt in ('float32', 'bfloat16')}
        eg0 = m(x).clone()
        assert rel(e0['float32'], oracle_emb()) < 1e-5
        m.train()
        for _ in range(3):
            opt.clear_grad()
            loss = HeadLoss.apply(m(x), W, y, 0.2, 32.0, 0.0, False)[0]
            loss.backward()
            opt.step()
        m.eval()
        ref1 = oracle_emb()
        assert rel(ref1, e0['float32']) > 1e-3                              # the weights did move
        e1 = {dt: m.engine(dt).forward(x) for dt in ('float32', 'bfloat16')}
        eg1 = m(x)
    finally:
        ppvector.set_graph_mode(False)
    assert rel(e1['float32'], ref1) < 1e-5, rel(e1['float32'], ref1)
    assert rel(eg1, ref1) < 1e-5 and rel(eg1, eg0) > 1e-3
    assert rel(e1['bfloat16'], ref1) < 2e-2 and rel(e1['bfloat16'], e0['bfloat16']) > 1e-3


# ------------------------------------------------------------------------------- mixed precision (train_conf.enable_amp)
@pytest.fixture
def amp():
    import ppvector
    ppvector.set_train_amp(True)
    yield
    ppvector.set_train_amp(False)


@pytest.mark.parametrize('case', [
    # B, T, Cin, Cout, kw, dil, pad
    (70, 131, 512, 512, 1, 1, 'none'),    # the wide 1x1 layers: the 128 x 128 weight-gradient tiles' fast path (source row = output row)
    (9, 77, 192, 320, 1, 1, 'none'),      # ragged tiles (N, K not multiples of 128), rows not a multiple of the 64-row chunk
    (3, 50, 80, 64, 5, 1, 'none'),        # taps, Cin % 64 != 0
    (3, 40, 64, 64, 3, 4, 'reflect'),     # ECAPA Res2 conv
    (2, 33, 80, 128, 5, 1, 'reflect'),    # ECAPA block0
])
def test_conv_block_grads_mixed_precision(N, amp, case):
    """enable_amp: forward, data gradient and weight gradient on the bf16 matrix cores over f32 tensors.  Forward == float64 conv
    of the bf16-rounded operands (tight); gradients vs float64 autograd of the UN-rounded graph at bf16 tolerances (each GEMM
    operand carries 2^-9 relative rounding: rel-L2 of a few 1e-3 per gradient)."""
    from ppvector.train.functions import ConvBlock
    B, T, Cin, Cout, kw, dil, pad = case
    g = torch.Generator().manual_seed(11 + kw + dil + Cin)
    x = torch.randn(B, T, Cin, generator=g, dtype=torch.float64, requires_grad=True)
    w = (torch.randn(Cout, Cin, kw, generator=g, dtype=torch.float64) / (Cin * kw) ** 0.5).requires_grad_()
    b = torch.randn(Cout, generator=g, dtype=torch.float64, requires_grad=True)
    xt = x.transpose(1, 2)
    if pad == 'reflect':
        xt = F.pad(xt, (dil * (kw - 1) // 2,) * 2, mode='reflect')
    y = F.conv1d(xt, w, b, dilation=dil).transpose(1, 2)
    dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
    y.backward(dy)
    xq, wq = x.detach().float().bfloat16().double(), w.detach().float().bfloat16().double()
    xtq = xq.transpose(1, 2)
    if pad == 'reflect':
        xtq = F.pad(xtq, (dil * (kw - 1) // 2,) * 2, mode='reflect')
    yq = F.conv1d(xtq, wq, b.detach(), dilation=dil).transpose(1, 2)
    xd = x.detach().float().reshape(B * T, Cin).cuda().requires_grad_()
    wd, bd = w.detach().float().cuda().requires_grad_(), b.detach().float().cuda().requires_grad_()
    out = ConvBlock.apply(xd, wd, bd, None, None, None, None, None, dict(B=B, T=T, dilation=dil, pad=pad, relu=False))
    out.backward(dy.float().reshape(-1, Cout).cuda())
    To = y.shape[1]
    assert rel(out.reshape(B, To, Cout), yq) < 2e-6                      # exactly the bf16-operand product, f32 accumulated
    assert rel(out.reshape(B, To, Cout), y.detach()) > 1e-4              # ... and not the f32 one
    for name, got, ref in (('dx', xd.grad.reshape(B, T, Cin), x.grad), ('dW', wd.grad, w.grad), ('dbias', bd.grad, b.grad)):
        r = rel(got, ref)
        print(f'[amp conv {case}] {name} rel-L2 {r:.2e}')
        assert r < (2e-5 if name == 'dbias' else 8e-3), (name, r)


def test_ecapa_training_step_mixed_precision(N, amp):
    """ECAPA-TDNN training step under enable_amp.  Two references over the oracle graph, both float64 autograd:
      exact -- the plain graph;  emulated -- the same graph with every conv GEMM's operands (x, w, and dz in both backward GEMMs)
      rounded to bf16 (oracle.models.AMP), i.e. the arithmetic the engine performs minus its f32 accumulation order.
    Measured on MI355X (B = 6, T = 60): loss 12.1558 (emulated 12.1512, exact 12.1613); whole-gradient rel-L2 engine vs exact
    2.25e-1, emulated vs exact 2.24e-1, engine vs emulated 9.3e-2.  bf16 GEMM operands move this tiny train-mode-BatchNorm graph
    by 22 % in its gradient whoever does the arithmetic (BatchNorm backward subtracts the dominant common mode of dy and so
    amplifies the 2^-9 operand rounding; an f32-vs-f64 accumulation difference flips individual roundings, hence engine and
    emulation differ too) -- a property of mixed precision on this problem, not of the kernels, which are held to 2.4e-3 per
    GEMM in test_conv_block_grads_mixed_precision.  Asserted: the engine is closer to the emulation than the emulation is to
    the exact graph, and no further from the exact graph than 1.5 x the emulation."""
    from ppvector.models.ecapa_tdnn import EcapaTdnn
    from ppvector.train.functions import HeadLoss
    B, T, Cc = 6, 60, 30
    p = om.ecapa_params(80, seed=21)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(B, T, 80, generator=g) * 2
    labels = torch.randint(0, Cc, (B,), generator=g)
    Wh = om.head_params(192, Cc, seed=5)

    def oracle(amp_on):
        pr = {k: v.clone().double().requires_grad_(not k.endswith(('_mean', '_variance'))) for k, v in p.items()}
        Wr = Wh.clone().double().requires_grad_()
        om.AMP = amp_on
        try:
            emb = om.ecapa_forward(pr, x.double(), training=True)
            loss = om.aam_loss(om.cosine_head(emb, Wr), labels, 0.2, 32.0, False, 0.0)
            loss.backward()
        finally:
            om.AMP = False
        return emb.detach(), loss.item(), {k: v.grad for k, v in pr.items() if v.grad is not None and v.grad.norm().item() >= 1e-9}

    emb_x, loss_x, g_x = oracle(False)
    emb_e, loss_e, g_e = oracle(True)
    m = EcapaTdnn(80)
    m.load_state_dict(p)
    m = m.cuda().train()
    Wd = Wh.cuda().requires_grad_()
    emb = m(x.cuda())
    loss = HeadLoss.apply(emb, Wd, labels.cuda(), 0.2, 32.0, 0.0, False)[0]
    loss.backward()
    got = {k: v.grad.double().cpu() for k, v in m.named_parameters()}

    def whole(a, b):
        num = sum((a[k] - b[k]).pow(2).sum().item() for k in b)
        return (num / sum(b[k].pow(2).sum().item() for k in b)) ** 0.5

    w_emu, w_exact, w_inh = whole(got, g_e), whole(got, g_x), whole(g_e, g_x)
    print(f'[ecapa train amp] loss {loss.item():.5f} (emulated {loss_e:.5f}, exact {loss_x:.5f});  emb rel-L2 vs emulated {rel(emb, emb_e):.2e}, '
          f'vs exact {rel(emb, emb_x):.2e};  whole-gradient rel-L2 vs emulated {w_emu:.2e}, vs exact {w_exact:.2e} '
          f'(emulated vs exact: {w_inh:.2e})')
    assert abs(loss.item() - loss_e) < 1e-3 * abs(loss_e) and rel(emb, emb_e) < 2e-2
    assert w_emu < w_inh, (w_emu, w_inh)
    assert abs(loss.item() - loss_x) < 2e-3 * abs(loss_x) and rel(emb, emb_x) < 3e-2
    assert w_exact < 1.5 * w_inh + 2e-2, (w_exact, w_inh)
    m.eval()


def _graphed_vs_eager(make_model, xs, ys, n_classes, margin_at=None, tol=1e-5):
    from ppvector.loss.aamloss import AAMLoss
    from ppvector.optimizer.adam import Adam
    from ppvector.train.step import GraphedTrainStep, TrainStep

    def run(cls):
        torch.manual_seed(0)
        model = make_model()
        crit = AAMLoss(margin=0.2, scale=32)
        opt = Adam(model.parameters(), learning_rate=2e-3, weight_decay=1e-6)
        step = cls(model, crit, opt)
        losses, accs = [], []
        for i, (x, y) in enumerate(zip(xs, ys)):
            if margin_at is not None and i >= margin_at:
                crit.update(0.2 + 0.05 * (i - margin_at + 1))    # MarginScheduler's ramp: a new margin every step
            loss, acc = step(x, y)
            losses.append(loss)                                   # collected as returned: every step must own its scalars
            accs.append(acc)
        torch.cuda.synchronize()
        return [float(v) for v in losses], [float(v) for v in accs], {k: v.detach().clone() for k, v in model.state_dict().items()}, step

    le, ae, se, _ = run(TrainStep)
    lg, ag, sg, st = run(GraphedTrainStep)
    assert st.capture_error is None, st.capture_error
    assert st.n_stages >= 1 and len(st._plans) == 1               # ONE capture: the margin ramp must not re-capture
    print(f'[graphed step, {st.n_stages} backward stage(s)] losses eager', [f'{v:.5f}' for v in le], 'graphed', [f'{v:.5f}' for v in lg])
    for a, b in zip(le, lg):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(a)), (le, lg)
    assert ae == ag, (ae, ag)
    worst = 0.0
    for k in se:
        d = (se[k].double() - sg[k].double()).abs().max().item()
        worst = max(worst, d / max(1.0, se[k].abs().max().item()))
        assert d <= tol * max(1.0, se[k].abs().max().item()), (k, d)
    print(f'[graphed step] worst relative parameter / running-statistic difference after {len(xs)} steps: {worst:.2e}')
    return st


def test_graphed_train_step_equals_the_eager_step(N):
    """GraphedTrainStep (forward + backward replayed from captured HIP graphs; all-reduce, Adam and the schedulers eager) against
    TrainStep from the same initial state on the same batches: losses and accuracies of every step, the trained parameters and
    the BatchNorm running statistics.  Eight steps: three eager sightings of the shape, the capture, replays with NEW inputs (the
    static buffers must be refreshed) and a loss margin that changes on every one of the last three steps -- device data
    (vp_set_margin_table): the graph must follow it WITHOUT a re-capture."""
    from ppvector.models.fc import SpeakerIdentification
    from ppvector.models.tdnn import TDNN
    g = torch.Generator().manual_seed(11)
    xs = [(torch.randn(6, 90, 80, generator=g) * 2).cuda() for _ in range(8)]
    ys = [torch.randint(0, 12, (6,), generator=g).cuda() for _ in range(8)]

    def make():
        m = TDNN(80)
        m.load_state_dict(om.tdnn_params(80, seed=5))
        head = SpeakerIdentification(192, 12)
        head.load_state_dict({'weight': om.head_params(192, 12, seed=6)})
        return torch.nn.Sequential(m, head).cuda()

    st = _graphed_vs_eager(make, xs, ys, 12, margin_at=5)
    assert st.n_stages == 1                                       # TDNN declares no cut points


@pytest.mark.parametrize('name', ['eres2net', 'resnetse'])
def test_graphed_2d_backbone_step_in_backward_stages_equals_the_eager_step(N, name):
    """ERes2Net / ResNetSE declare a cut point after each of their first three stages (train/eres2net_train.py, train/resnetse_train.py):
    the captured step is FOUR graphs (layer4 + fusion / pooling + embedding + head | layer3 | layer2 | stem + layer1).  Same losses,
    parameters and running statistics as the eager step; the first backward stage already holds the LATE parameters -- the bulk of the
    flat gradient buffer -- so the data-parallel all-reduce of BASELINE configs[3] / [4] starts under the remaining stages."""
    from ppvector.models.eres2net import ERes2Net
    from ppvector.models.fc import SpeakerIdentification
    from ppvector.models.resnet_se import ResNetSE
    g = torch.Generator().manual_seed(31)
    xs = [(torch.randn(4, 40, 16, generator=g) * 2).cuda() for _ in range(7)]
    ys = [torch.randint(0, 9, (4,), generator=g).cuda() for _ in range(7)]

    def make():
        torch.manual_seed(3)
        m = ERes2Net(16, num_blocks=[1, 1, 1, 1]) if name == 'eres2net' else ResNetSE(16, layers=[1, 1, 1, 1])
        head = SpeakerIdentification(192, 9)
        head.load_state_dict({'weight': om.head_params(192, 9, seed=22)})
        return torch.nn.Sequential(m, head).cuda()

    # (1) one backward through the cut tape against one through the uncut tape, same weights, same batch: every parameter gradient.  A
    # stage output that feeds the next stage AND the bottom-up fusion collects its two contributions at the cut's leaf in another order
    # than the uncut tape adds them -- f32 rounding apart, nothing more
    from ppvector.loss.aamloss import AAMLoss
    from ppvector.train.segments import Recorder
    grads = {}
    for mode in ('plain', 'cut'):
        model = make().train()
        crit = AAMLoss(margin=0.2, scale=32)
        if mode == 'cut':
            rec = Recorder()
            with rec:
                loss = crit(model(xs[0]), ys[0])
                rec.backward(loss)
            assert rec.n_stages == 4
        else:
            crit(model(xs[0]), ys[0]).backward()
        grads[mode] = {k: v.grad.detach().double().clone() for k, v in model.named_parameters() if v.grad is not None}
    assert grads['plain'].keys() == grads['cut'].keys()
    worst = max(((grads['cut'][k] - g).norm() / g.norm().clamp(min=1e-12)).item() for k, g in grads['plain'].items() if g.norm().item() > 1e-9)
    print(f'[graphed {name}] one backward, cut tape vs uncut tape: worst parameter-gradient rel-L2 {worst:.2e}')
    assert worst < 2e-5, worst
    # (2) the captured step against the eager step over seven Adam steps (lr 2e-3): g / sqrt(v) amplifies the rounding-level gradient
    # differences of (1) where a gradient is tiny -- 1e-3 of a weight's scale for ERes2Net; the plain chain (ResNetSE) stays at 1e-5
    st = _graphed_vs_eager(make, xs, ys, 9, margin_at=5, tol=1e-3 if name == 'eres2net' else 1e-5)
    assert st.n_stages == 4
    plan = next(iter(st._plans.values()))
    n = st.optimizer.grad.numel()
    first = sum(b - a for a, b in plan['spans'][0])
    covered = sorted(sp for stage in plan['spans'] for sp in stage)
    assert covered[0][0] == 0 and covered[-1][1] == n and all(a[1] == b[0] for a, b in zip(covered, covered[1:])), covered   # a tiling
    print(f'[graphed {name}] {st.n_stages} backward stages; the first holds {first / n:.0%} of the flat gradient buffer; chunk ready stages {plan["ready"]}')
    assert first > 0.5 * n


def test_graphed_ecapa_step_in_backward_stages_equals_the_eager_step(N):
    """ECAPA-TDNN declares cut points after every SE-Res2 block (train/ecapa_train.py): the captured step is FOUR graphs (head + ASP +
    MFA | block 3 | block 2 | blocks 0-1), each ending with its gradients gathered into the flat buffer -- the slices a data-parallel
    run all-reduces while the next graph replays.  Same losses, parameters and running statistics as the eager step, and the
    stages' flat-buffer spans tile the gradient buffer exactly, last parameters first."""
    from ppvector.models.ecapa_tdnn import EcapaTdnn
    from ppvector.models.fc import SpeakerIdentification
    g = torch.Generator().manual_seed(12)
    xs = [(torch.randn(5, 140, 80, generator=g) * 2).cuda() for _ in range(7)]
    ys = [torch.randint(0, 9, (5,), generator=g).cuda() for _ in range(7)]

    def make():
        m = EcapaTdnn(80, embd_dim=192, pooling_type='ASP', channels=[512, 512, 512, 512, 1536])
        m.load_state_dict(om.ecapa_params(80, seed=21))
        head = SpeakerIdentification(192, 9)
        head.load_state_dict({'weight': om.head_params(192, 9, seed=22)})
        return torch.nn.Sequential(m, head).cuda()

    st = _graphed_vs_eager(make, xs, ys, 9, margin_at=5)
    assert st.n_stages == 4
    plan = next(iter(st._plans.values()))
    spans = [sp for stage in plan['spans'] for sp in stage]
    assert all(len(stage) == 1 for stage in plan['spans']), plan['spans']        # registration order = forward order: one slice per stage
    assert spans[0][1] == st.optimizer.grad.numel() and spans[-1][0] == 0
    assert all(a[0] == b[1] for a, b in zip(spans, spans[1:])), spans           # later stages = earlier parameters, no gap, no overlap
    print('[graphed ecapa] flat-gradient slices per backward stage (elements):', spans)


def _two_rank_graphed_worker(rank, world, port, q, ragged=False):
    import os
    import sys
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)                                       # both ranks share the one GPU of the test box; gloo carries the sums
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from oracle import models as om
    from ppvector.loss.aamloss import AAMLoss
    from ppvector.models.ecapa_tdnn import EcapaTdnn
    from ppvector.models.fc import SpeakerIdentification
    from ppvector.optimizer.adam import Adam
    from ppvector.train.ddp import shard_batch
    from ppvector.train.step import GraphedTrainStep
    g = torch.Generator().manual_seed(31)
    xs = [(torch.randn(8, 120, 80, generator=g) * 2) for _ in range(6)]
    ys = [torch.randint(0, 9, (8,), generator=g) for _ in range(6)]
    m = EcapaTdnn(80, embd_dim=192, pooling_type='ASP', channels=[512, 512, 512, 512, 1536])
    m.load_state_dict(om.ecapa_params(80, seed=21))
    head = SpeakerIdentification(192, 9)
    head.load_state_dict({'weight': om.head_params(192, 9, seed=22)})
    model = torch.nn.Sequential(m, head).cuda()
    opt = Adam(model.parameters(), learning_rate=2e-3, weight_decay=1e-6)
    step = GraphedTrainStep(model, AAMLoss(margin=0.2, scale=32), opt)
    idx = list(shard_batch(8, rank, world))
    losses = []
    for i, (x, y) in enumerate(zip(xs, ys)):
        xi = x[idx]
        if ragged and rank == 1 and i >= 4:
            xi = xi[:, :96 + 8 * (i - 4)]        # a padded length this rank has never seen: it steps eagerly while rank 0 replays graphs
            import time
            time.sleep(0.05)                     # ... and late: the peer's collective waits for it
        loss, _ = step(xi.cuda(), y[idx].cuda())
        losses.append(float(loss))
    torch.cuda.synchronize()
    if ragged:
        q.put((rank, losses, step.n_stages, step.capture_error, opt.flat.detach().cpu().numpy(), step.faults, len(step._plans)))
        dist.barrier()
        dist.destroy_process_group()
        return
    # numpy (pickled by value): a torch tensor would travel as a file descriptor of this process, which may be gone by the read
    q.put((rank, losses, step.n_stages, step.capture_error, opt.flat.detach().cpu().numpy(),
           {k: v.detach().cpu().numpy() for k, v in model.state_dict().items() if k.endswith(('_mean', '_variance'))}))
    dist.barrier()
    dist.destroy_process_group()


def test_graphed_train_step_under_two_ranks_equals_the_sharded_reference(N):
    """GraphedTrainStep with world size 2 (two processes on this box's one GPU, gloo between them -- the collective is the only
    thing that differs from the RCCL run): every rank replays its four stage graphs on ITS shard and all-reduces each stage's slice
    of the flat gradient buffer.  Reference, in this process: the two shards differentiated one after the other on two replicas
    (BatchNorm statistics are rank-local in the reference: plain BatchNorm under fleet, trainer.py:318-320), gradients averaged,
    one Adam step on each replica.  After six steps (three eager, the capture, two replays) the ranks' parameters equal the
    reference's and each other's; each rank's running statistics equal its replica's."""
    import torch.multiprocessing as mp
    from ppvector.loss.aamloss import AAMLoss
    from ppvector.models.ecapa_tdnn import EcapaTdnn
    from ppvector.models.fc import SpeakerIdentification
    from ppvector.optimizer.adam import Adam
    from ppvector.train.ddp import shard_batch
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 35500 + os.getpid() % 2000
    procs = [ctx.Process(target=_two_rank_graphed_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    # the reference while the ranks run
    g = torch.Generator().manual_seed(31)
    xs = [(torch.randn(8, 120, 80, generator=g) * 2) for _ in range(6)]
    ys = [torch.randint(0, 9, (8,), generator=g) for _ in range(6)]
    reps = []
    for r in range(2):
        m = EcapaTdnn(80, embd_dim=192, pooling_type='ASP', channels=[512, 512, 512, 512, 1536])
        m.load_state_dict(om.ecapa_params(80, seed=21))
        head = SpeakerIdentification(192, 9)
        head.load_state_dict({'weight': om.head_params(192, 9, seed=22)})
        model = torch.nn.Sequential(m, head).cuda().train()
        reps.append((model, Adam(model.parameters(), learning_rate=2e-3, weight_decay=1e-6), AAMLoss(margin=0.2, scale=32)))
    ref_losses = [[], []]
    for x, y in zip(xs, ys):
        for r, (model, opt, crit) in enumerate(reps):
            idx = list(shard_batch(8, r, 2))
            loss = crit(model(x[idx].cuda()), y[idx].cuda())
            loss.backward()
            opt.pack_grads()
            ref_losses[r].append(float(loss.detach()))
        gsum = reps[0][1].grad + reps[1][1].grad
        for model, opt, _ in reps:
            opt.grad.copy_(gsum)
            opt.step(grad_scale=0.5)
            opt.clear_grad()
    torch.cuda.synchronize()
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    flat_ref = reps[0][1].flat.detach().cpu()
    assert torch.equal(flat_ref, reps[1][1].flat.detach().cpu())
    for rank, losses, n_stages, err, flat, stats in res:
        flat, stats = torch.from_numpy(flat), {k: torch.from_numpy(v) for k, v in stats.items()}
        assert err is None and n_stages == 4, (err, n_stages)
        dl = max(abs(a - b) / max(1.0, abs(a)) for a, b in zip(ref_losses[rank], losses))
        dp = ((flat - flat_ref).abs().max() / flat_ref.abs().max()).item()
        sd = reps[rank][0].state_dict()
        ds = max(((stats[k] - sd[k].cpu()).abs().max() / max(1.0, sd[k].abs().max().item())).item() for k in stats)
        print(f'[2-rank graphed step] rank {rank}: losses {[f"{v:.5f}" for v in losses]}  rel loss diff {dl:.2e}  '
              f'parameters vs sharded reference {dp:.2e}  running statistics {ds:.2e}')
        assert dl < 1e-5 and dp < 1e-5 and ds < 1e-5, (dl, dp, ds)
    assert np.array_equal(res[0][4], res[1][4])                   # the ranks hold bit-identical parameters


def test_ranks_in_different_step_modes_issue_the_same_collectives(N):
    """GraphedTrainStep picks eager or graphed from rank-local state, and the graphed step used to all-reduce one
    slice per backward stage while the eager step reduced the whole buffer -- ranks whose padded lengths differ would have enqueued
    mismatched collectives.  The schedule is now `reduce_chunks(n_params)`: fixed chunks, last to first, in every mode.  Here rank 1
    meets two NEW padded lengths at steps 5 and 6 (eager, and 50 ms late) while rank 0 replays its four stage graphs: the job must
    finish, and the ranks must hold bit-identical parameters (they applied the same summed gradients)."""
    import torch.multiprocessing as mp
    from ppvector.train.step import reduce_chunks
    for n in (5, (1 << 20) + 3, 6_700_000, 94_000_000):
        ch = reduce_chunks(n)
        assert ch[0][1] == n and ch[-1][0] == 0 and all(a[0] == b[1] for a, b in zip(ch, ch[1:])) and len(ch) <= 17, (n, len(ch))
    assert len(reduce_chunks(6_700_000)) == 7
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 37500 + os.getpid() % 2000
    procs = [ctx.Process(target=_two_rank_graphed_worker, args=(r, 2, port, q, True)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in range(2)), key=lambda t: t[0])
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for rank, losses, n_stages, err, flat, faults, plans in res:
        assert err is None and faults == 0 and np.all(np.isfinite(losses)), (rank, err, faults, losses)
        print(f'[mixed step modes] rank {rank}: losses {[f"{v:.5f}" for v in losses]}  graphs kept for {plans} shape(s), {n_stages} stages')
    assert res[0][6] == 1 and res[1][6] == 1                      # each rank captured its one repeated shape; rank 1's new lengths ran eagerly
    assert np.array_equal(res[0][4], res[1][4])                   # same summed gradients everywhere -> bit-identical parameters


@pytest.mark.parametrize('M,C,two', [(76288, 512, True), (5000, 64, True), (777, 128, False), (1234, 1536, True), (256, 192, True),
                                      (999, 6, True), (19, 64, True)])
def test_col_sums_kernel_vs_float64(N, M, C, two):
    """vp_col_sums_f32 (four-channels-per-lane kernel; C = 6 takes the scalar one): both BatchNorm-backward reductions."""
    from ppvector.train.functions import col_sums
    g = torch.Generator().manual_seed(M + C)
    a = torch.randn(M, C, generator=g)
    b = torch.randn(M, C, generator=g) * 2 + 0.5
    mu, sc = torch.randn(C, generator=g), torch.rand(C, generator=g) + 0.5
    if two:
        got = col_sums(a.cuda(), b.cuda(), mu.cuda(), sc.cuda())
        want1 = (a.double() * (b.double() - mu.double()) * sc.double()).sum(0)
        assert (got[1].double().cpu() - want1).abs().max().item() <= 2e-6 * (a.abs() * (b - mu).abs() * sc).double().sum(0).max().item()
    else:
        got = col_sums(a.cuda())
    want0 = a.double().sum(0)
    assert (got[0].double().cpu() - want0).abs().max().item() <= 2e-6 * a.abs().double().sum(0).max().item()


@pytest.mark.parametrize('M,Cout,Cin,ldx,xoff', [(4133, 512, 256, 256, 0), (9000, 256, 768, 1032, 264), (76288, 512, 512, 512, 0), (64, 256, 256, 256, 0),
                                                  (2500, 1536, 1536, 1536, 0), (3000, 192, 512, 512, 0),
                                                  # round 5: a 128-column side on the 256-tile kernel (ASP's attention TDNN and logits conv at 256 x 298 frames)
                                                  (76288, 128, 1536, 1536, 0), (76288, 1536, 128, 128, 0), (76289, 128, 1536, 1800, 264)])
def test_wgrad_of_bf16_operands_vs_float64(N, M, Cout, Cin, ldx, xoff):
    """vp_conv1d_wgrad_bf16_oik on the wide 1x1 layers: the 256-tile kernel on transposing LDS reads (csrc/wgrad_tr.hip; the last case has
    Cout % 256 != 0 and runs the 128-tile kernel).  bf16 products are exact in f32, so the only error is the f32 accumulation order:
    1e-5 of sum |dz| |x| per output (measured ~1e-6); twenty launches are bit-identical."""
    import ctypes as C
    lib, ctx = N.lib(), N.ctx(torch.device('cuda', 0))
    g = torch.Generator().manual_seed(M + Cout)
    x = torch.randn(M, ldx, generator=g).to(torch.bfloat16)
    dz = (torch.randn(M, Cout, generator=g) * 0.1).to(torch.bfloat16)
    xd, dzd = x.cuda(), dz.cuda()
    d = N.Conv1dDesc()
    d.dtype_in, d.dtype_out = N.VP_BF16, N.VP_F32
    d.B, d.T_in, d.T_out, d.Cin, d.Cout, d.KW, d.dilation, d.stride = 1, M, M, Cin, Cout, 1, 1, 1
    d.pad_mode, d.pad_left = N.VP_PAD_ZERO, 0
    d.x, d.ldx, d.xoff = xd.data_ptr(), ldx, xoff
    d.mfma_bf16 = 1
    ws = torch.empty(lib.vp_conv1d_wgrad_workspace_bytes(C.byref(d)), dtype=torch.uint8, device='cuda')
    outs = []
    for _ in range(20):
        dW = torch.full((Cout, Cin), float('nan'), dtype=torch.float32, device='cuda')
        N.check(lib.vp_conv1d_wgrad_bf16_oik(ctx, C.byref(d), dzd.data_ptr(), Cout, dW.data_ptr(), ws.data_ptr(), ws.numel(), N.stream_ptr()), ctx)
        torch.cuda.synchronize()
        outs.append(dW)
    assert all(torch.equal(o, outs[0]) for o in outs[1:])
    xs = x[:, xoff:xoff + Cin].double()
    want = dz.double().t() @ xs
    scale = (dz.double().abs().t() @ xs.abs()).max().item()
    err = (outs[0].double().cpu() - want).abs().max().item()
    print(f'[wgrad bf16 M={M} {Cout}x{Cin}] max err {err:.3e} against sum |dz||x| {scale:.3e}')
    assert err <= 1e-5 * scale, (err, scale)


@pytest.mark.parametrize('case', [(2, 77, 64, 64, 3, 2, 'reflect'), (3, 298, 512, 512, 1, 1, 'reflect'), (2, 64, 80, 512, 5, 1, 'reflect'),
                                  (4, 130, 128, 192, 1, 1, 'reflect'), (2, 60, 128, 64, 3, 3, 'none')])
def test_wgrad_split_precision_vs_float64(N, case):
    """vp_conv1d_wgrad_oik_f32 with mfma_bf16 = 2 (conv_wgrad_amp_kernel<false, true>: both operands as transposed bf16 hi / lo planes,
    three MFMAs per fragment pair) against the float64 weight gradient of the unrounded f32 operands, beside the single-bf16-pass form."""
    import ctypes as C
    lib, ctx = N.lib(), N.ctx(0)
    B, T, Cin, Cout, kw, dil, pad = case
    g = torch.Generator().manual_seed(sum(case[:6]))
    x = torch.randn(B, T, Cin, generator=g)
    T_out = T if pad != 'none' else T - dil * (kw - 1)
    dz = torch.randn(B, T_out, Cout, generator=g)
    xt = x.double().transpose(1, 2)
    if pad == 'reflect' and kw > 1:
        xt = F.pad(xt, (dil * (kw - 1) // 2,) * 2, mode='reflect')
    w = torch.zeros(Cout, Cin, kw, dtype=torch.float64, requires_grad=True)
    y = F.conv1d(xt, w, None, dilation=dil).transpose(1, 2)
    (y * dz.double()).sum().backward()
    ref = w.grad                                                       # (Cout, Cin, kw)
    xd, dzd = x.cuda(), dz.cuda()
    res = {}
    for mode in (2, 1, 0):
        d = N.Conv1dDesc()
        d.dtype_in = d.dtype_out = N.VP_F32
        d.B, d.T_in, d.T_out, d.Cin, d.Cout, d.KW, d.dilation, d.stride = B, T, T_out, Cin, Cout, kw, dil, 1
        d.pad_mode = {'none': N.VP_PAD_NONE, 'reflect': N.VP_PAD_REFLECT}[pad]
        d.pad_left = 0 if pad == 'none' else dil * (kw - 1) // 2
        d.x, d.ldx, d.mfma_bf16 = xd.data_ptr(), Cin, mode
        dW = torch.zeros(Cout, Cin, kw, device='cuda')
        ws = torch.empty(int(lib.vp_conv1d_wgrad_workspace_bytes(C.byref(d))), dtype=torch.uint8, device='cuda')
        N.check(lib.vp_conv1d_wgrad_oik_f32(ctx, C.byref(d), dzd.data_ptr(), Cout, dW.data_ptr(), ws.data_ptr(), ws.numel(), N.stream_ptr()), ctx)
        torch.cuda.synchronize()
        res[mode] = ((dW.double().cpu() - ref).norm() / ref.norm()).item()
    print(f'[wgrad {case}] rel-L2 vs float64: exact f32 {res[0]:.2e}, split precision {res[2]:.2e}, one bf16 pass {res[1]:.2e}')
    assert res[2] < 2e-5 and res[1] > 30 * res[2], res


@pytest.mark.parametrize('M,C,relu,gamma', [(76288, 512, 1, True), (4097, 64, 1, True), (300, 128, 0, True), (1000, 1536, 1, False),
                                            (23, 192, 1, True)])
def test_bn_relu_bwd_dbias_kernel_vs_float64(N, M, C, relu, gamma):
    """vp_bn_relu_bwd_dbias_f32: dz through BatchNorm (batch statistics) and the ReLU mask, and its column sums, against the
    float64 formula, and against the unfused kernel."""
    lib, ctx = N.lib(), N.ctx(0)
    g = torch.Generator().manual_seed(M * 7 + C)
    dy, z = torch.randn(M, C, generator=g), torch.randn(M, C, generator=g)
    mean, invstd = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    gam = torch.randn(C, generator=g) if gamma else None
    zh = (z.double() - mean.double()) * invstd.double()
    sums = torch.stack([dy.double().sum(0), (dy.double() * zh).sum(0)]).float()
    want = (gam.double() if gamma else 1.0) * invstd.double() * (dy.double() - sums[0].double() / M - zh * sums[1].double() / M)
    if relu:
        want = want * (z > 0)
    dyc, zc, mc, ic, sc = dy.cuda(), z.cuda(), mean.cuda(), invstd.cuda(), sums.cuda()
    gc = gam.cuda() if gamma else None
    dz, dz0 = torch.empty_like(dyc), torch.empty_like(dyc)
    db = torch.empty(C, device='cuda')
    ws = torch.empty(lib.vp_bn_relu_bwd_dbias_workspace_bytes(M, C), dtype=torch.uint8, device='cuda')
    rc = lib.vp_bn_relu_bwd_dbias_f32(ctx, dyc.data_ptr(), C, zc.data_ptr(), C, mc.data_ptr(), ic.data_ptr(),
                                      gc.data_ptr() if gamma else None, sc.data_ptr(), M, C, relu, dz.data_ptr(), C, db.data_ptr(),
                                      ws.data_ptr(), ws.numel(), N.stream_ptr())
    N.check(rc, ctx)
    rc = lib.vp_bn_relu_bwd_f32(ctx, dyc.data_ptr(), C, zc.data_ptr(), C, mc.data_ptr(), ic.data_ptr(), gc.data_ptr() if gamma else None,
                                sc.data_ptr(), M, C, relu, dz0.data_ptr(), C, N.stream_ptr())
    N.check(rc, ctx)
    torch.cuda.synchronize()
    assert rel(dz, want) < 2e-6
    assert rel(dz, dz0) < 1e-6                 # the two kernels contract their multiply-adds differently: not bit-equal
    assert torch.equal(dz == 0, dz0 == 0)      # ... but the ReLU mask is the same
    wdb = want.sum(0)
    assert (db.double().cpu() - wdb).abs().max().item() <= 2e-6 * want.abs().sum(0).max().item()


def _tdnn_block_run(xs, w, b, gam, bet, dy, B, T, cat):
    """One TDNNBlock (1x1 conv -> ReLU -> BatchNorm, batch statistics) forward + backward; cat: through CatConvBlock over xs."""
    from ppvector.train.functions import CatConvBlock, ConvBlock
    xs = [x.clone().requires_grad_() for x in xs]
    w, b, gam, bet = (t.clone().requires_grad_() for t in (w, b, gam, bet))
    rm, rv = torch.zeros_like(b), torch.ones_like(b)
    cfg = dict(B=B, T=T, pad='reflect', relu=True, momentum=0.9, eps=1e-5)
    if cat:
        y = CatConvBlock.apply(cfg, w, b, gam, bet, rm, rv, *xs)
    else:
        y = ConvBlock.apply(torch.cat(xs, dim=1), w, b, None, gam, bet, rm, rv, cfg)
    y.backward(dy)
    return [y.detach(), rm, rv, w.grad, b.grad, gam.grad, bet.grad] + [x.grad for x in xs]


@pytest.mark.parametrize('cat,level', [(False, 1), (True, 1), (False, 2), (True, 2)])
def test_wide_bf16_operands_equal_f32_operand_amp(N, amp, cat, level, monkeypatch):
    """enable_amp, wide 1x1 TDNN blocks (M >= 4096, C >= 256).  Level 1: x and dz kept as bf16 tensors (bf16 -> f32 conv kernels --
    since round 3 the 128 x 256 LDS-DMA kernel for forward and data gradient --, bf16-input weight gradient, dz written as bf16 by
    the BatchNorm backward) must give what the f32-operand mixed-precision kernels give: the same roundings, only the accumulation
    order differs.  y: 2e-5.  The gradients see the order through the ReLU mask: a pre-activation within an ulp of 0
    lands on the other side in a handful of the 9.8 M elements and moves dz there by a whole gradient value -- 3e-4 (measured 9e-5).  Level 2 (the default): the pre-BatchNorm activation z is
    stored as bf16 too (forward on the 256-wide bf16 -> bf16 kernel, batch statistics from its f32 accumulators): ONE more rounding of
    2^-9 relative per element of z, so outputs and gradients stay within 5e-3 rel-L2 of level 0 (measured ~1e-3).
    cat: the MFA form (CatConvBlock builds the bf16 concatenation)."""
    B, T, C, Cout = 64, 300, 256, 512
    g = torch.Generator().manual_seed(5)
    n_in = 2 if cat else 1
    xs = [torch.randn(B * T, C, generator=g).cuda() for _ in range(n_in)]
    w = (torch.randn(Cout, C * n_in, 1, generator=g) / (C * n_in) ** 0.5).cuda()
    b, gam, bet = (torch.randn(Cout, generator=g).cuda() * s + o for s, o in ((0.3, 0.0), (0.2, 1.0), (0.2, 0.0)))
    dy = torch.randn(B * T, Cout, generator=g).cuda()
    monkeypatch.setenv('VPMI_TRAIN_BF16_OPS', '0')
    ref = _tdnn_block_run(xs, w, b, gam, bet, dy, B, T, False)
    monkeypatch.setenv('VPMI_TRAIN_BF16_OPS', str(level))
    got = _tdnn_block_run(xs, w, b, gam, bet, dy, B, T, cat)
    names = ['y', 'running mean', 'running var', 'dW', 'dbias', 'dgamma', 'dbeta'] + [f'dx{i}' for i in range(n_in)]
    for name, a, r in zip(names, got, ref):
        e = rel(a, r)
        print(f'[wide bf16 operands level {level}, cat={cat}] {name} rel-L2 {e:.2e}')
        if name.startswith('running'):
            assert e < 2e-5, (name, e)               # statistics come from the f32 accumulators at every level
        elif level == 1:
            assert e < (2e-5 if name == 'y' else 3e-4), (name, e)
        else:
            assert e < (2e-2 if name == 'dbias' else 5e-3), (name, e)


def test_asp_bf16_logit_gradient_stays_within_bf16_of_the_f32_path(N, amp, monkeypatch):
    """enable_amp, ASP at M >= 16384 rows: the statistics' backward writes d e as bf16 (vp_attn_stats_bwd_de16) and the logits conv's
    two backward GEMMs read that tensor instead of rounding an f32 one on the fly -- the same rounding, so every gradient must
    match the f32-d e path (VPMI_TRAIN_BF16_OPS=0) to accumulation order; the logits' bias gradient is exactly zero either way."""
    from ppvector.models.pooling import AttentiveStatisticsPooling
    from ppvector.train.tdnn_train import asp_forward
    B, T, C = 64, 300, 256
    torch.manual_seed(3)
    asp = AttentiveStatisticsPooling(C, attention_channels=128, global_context=True).cuda().train()
    x0 = torch.randn(B * T, C, device='cuda')
    dp = torch.randn(B, 2 * C, device='cuda')

    def run(level):
        monkeypatch.setenv('VPMI_TRAIN_BF16_OPS', level)
        for p in asp.parameters():
            p.grad = None
        x = x0.clone().requires_grad_()
        out = asp_forward(asp, x, B, T)
        out.backward(dp)
        return [out.detach(), x.grad] + [p.grad for p in asp.parameters()], [n for n, _ in asp.named_parameters()]

    ref, names = run('0')
    got, _ = run('2')
    for name, a, r in zip(['pooled', 'dx'] + names, got, ref):
        if r.norm().item() < 1e-9:
            assert a.abs().max().item() < 1e-6, name
            continue
        e = rel(a, r)
        print(f'[asp de16] {name} rel-L2 {e:.2e}')
        assert e < 2e-5, (name, e)


def test_ecapa_amp_operand_levels_agree_at_bench_scale(N, monkeypatch):
    """ECAPA-TDNN training step under enable_amp at the bench utterance length (56 x 298 frames = 16 688 rows: the wide layers take
    their bf16-operand paths), against the f32 engine's step on the same batch (itself within 7e-4 of float64 autograd).
    Level 1 (bf16 GEMM operands: the same roundings as level 0's f32 operands rounded on the fly, but since round 3 on the 128 x 256
    LDS-DMA kernel -- another accumulation order, and through ReLU masks / train-mode BatchNorm an ulp is amplified like any other
    perturbation of this graph) must be no further from the f32 step than level 0 is (x 1.1).  Level 2 (the default: the
    pre-BatchNorm activation of the seven wide layers stored as bf16) is one more rounding of 2^-9 per element of those tensors; on
    this random-init graph every bf16 rounding is amplified by the train-mode BatchNorm backward (see
    test_ecapa_training_step_mixed_precision), so the yardstick is the distance to the f32 step: level 2 must be no further from
    it than 1.25 x level 0 is.  Measured on MI355X: printed below."""
    import ppvector
    from ppvector.models.ecapa_tdnn import EcapaTdnn
    from ppvector.train.functions import HeadLoss
    B, T, Cc = 56, 298, 40
    p = om.ecapa_params(80, seed=31)
    g = torch.Generator().manual_seed(8)
    x = (torch.randn(B, T, 80, generator=g) * 2).cuda()
    labels = torch.randint(0, Cc, (B,), generator=g).cuda()
    Wh = om.head_params(192, Cc, seed=6)

    def run(level):
        ppvector.set_train_amp(level is not None)
        try:
            if level is not None:
                monkeypatch.setenv('VPMI_TRAIN_BF16_OPS', level)
            m = EcapaTdnn(80)
            m.load_state_dict(p)
            m = m.cuda().train()
            Wd = Wh.cuda().requires_grad_()
            emb = m(x)
            loss = HeadLoss.apply(emb, Wd, labels, 0.2, 32.0, 0.0, False)[0]
            loss.backward()
            return loss.item(), emb.detach().double().cpu(), {k: v.grad.double().cpu() for k, v in m.named_parameters()}
        finally:
            ppvector.set_train_amp(False)

    def whole(a, b):
        return (sum((a[k] - b[k]).pow(2).sum().item() for k in b) / sum(b[k].pow(2).sum().item() for k in b)) ** 0.5

    lx, ex, gx = run(None)
    l0, e0, g0 = run('0')
    l1, e1, g1 = run('1')
    l2, e2, g2 = run('2')
    print(f'[ecapa amp levels] loss f32 {lx:.5f}, levels 0 / 1 / 2: {l0:.5f} / {l1:.5f} / {l2:.5f};  emb rel-L2 vs f32: {rel(e0, ex):.2e} / '
          f'{rel(e1, ex):.2e} / {rel(e2, ex):.2e};  whole-gradient rel-L2 vs f32: {whole(g0, gx):.2e} / {whole(g1, gx):.2e} / '
          f'{whole(g2, gx):.2e};  level 2 vs level 0: emb {rel(e2, e0):.2e}, gradient {whole(g2, g0):.2e}')
    # (levels 0 and 1 differ by summation order only; through 21 chained Res2 chunks and train-mode BatchNorm that is the same ~15 % as
    # either has to f32 -- the yardstick is 'no further from f32', with 25 % slack for the run-to-run spread of that figure)
    assert abs(l1 - lx) < 2e-3 * abs(lx) and rel(e1, ex) < 1.25 * rel(e0, ex) + 1e-3 and whole(g1, gx) < 1.25 * whole(g0, gx)
    assert abs(l2 - lx) < 2e-3 * abs(lx) and rel(e2, ex) < 1.5 * rel(e0, ex) + 1e-3
    assert whole(g2, gx) < 1.5 * whole(g0, gx)      # (level 2 since round 4: every activation between the GEMMs stored as bf16 -- measured 1.32x / 1.08x of level 0)


@pytest.mark.parametrize('S,B,T,dil', [(2, 6, 100, 2), (2, 7, 298, 3), (2, 64, 298, 4), (3, 5, 67, 4), (3, 9, 298, 2), (8, 6, 100, 2), (8, 256, 298, 4)])
def test_res2_chain_one_launch_per_direction_equals_the_per_chunk_path(N, amp, monkeypatch, S, B, T, dil):
    """enable_amp: Res2NetBlock forward / backward as ONE launch each (csrc/res2_train.hip: a workgroup per utterance, batch statistics
    through an in-kernel grid barrier) against the per-chunk launch sequence (VPMI_RES2_TRAIN_UNFUSED=1).  Same roundings (conv
    operands to bf16, f32 accumulation and statistics); what differs is summation order (1e-7).  A chain AMPLIFIES that: a chunk input
    within 1e-7 of a bf16 rounding boundary rounds the other way (2^-9 on that element), the next chunk sees 1e-5, ... up to the bf16
    noise floor after a few chunks, and in backward every ReLU mask that flips moves dz by a whole gradient value (error = sqrt of the
    flipped share).  So: scale 2 (one conv, no amplification) pins the kernels -- outputs 2e-6, gradients 2e-4; scale 3 adds the
    hand-off -- 1e-4 / 5e-3; scale 8 (ECAPA) is bounded at the noise floor the per-chunk path has against ITSELF under a reordered
    sum -- 3e-3 / 5e-2 (measured values printed).  (8, 256, 298, 4) is the bench batch: 256 co-resident workgroups meet 7 + 8 times."""
    from ppvector.train.functions import Res2Fn
    w = 64
    g = torch.Generator().manual_seed(B * 1000 + T)
    x0 = torch.randn(B * T, S * w, generator=g).cuda()
    dout = torch.randn(B * T, S * w, generator=g).cuda()
    base = []
    for i in range(S - 1):
        base += [(torch.randn(w, w, 3, generator=g) / (3 * w) ** 0.5).cuda(), (torch.randn(w, generator=g) * 0.3).cuda(),
                 (torch.randn(w, generator=g) * 0.2 + 1.0).cuda(), (torch.randn(w, generator=g) * 0.2).cuda(),
                 torch.zeros(w).cuda(), torch.ones(w).cuda()]
    cfg = dict(B=B, T=T, scale=S, dilation=dil, momentum=0.9, eps=1e-5)

    def run(unfused):
        if unfused:
            monkeypatch.setenv('VPMI_RES2_TRAIN_UNFUSED', '1')
        else:
            monkeypatch.delenv('VPMI_RES2_TRAIN_UNFUSED', raising=False)
        params = [p.clone() for p in base]
        for i in range(S - 1):
            for k in range(4):
                params[6 * i + k].requires_grad_()
        x = x0.clone().requires_grad_()
        out = Res2Fn.apply(x, cfg, *params)
        out.backward(dout)
        torch.cuda.synchronize()
        res = {'out': out.detach(), 'dx': x.grad}
        for i in range(S - 1):
            res[f'run_mean{i}'], res[f'run_var{i}'] = params[6 * i + 4], params[6 * i + 5]
            for k, nm in enumerate(('dW', 'dbias', 'dgamma', 'dbeta')):
                res[f'{nm}{i}'] = params[6 * i + k].grad
        return res

    ref, got = run(True), run(False)
    assert N.lib().vp_grid_barrier_status(N.ctx(x0.device)) == 0, 'a grid barrier gave up waiting'
    worst = {}
    for k in ref:
        e = rel(got[k], ref[k])
        kind = k.rstrip('0123456789')
        worst[kind] = max(worst.get(kind, 0.0), e)
    print(f'[res2 train chain scale={S} B={B} T={T} dil={dil}] out / dx per 64-channel slice: ' + ' '.join(f"{rel(got['out'][:, i * w:(i + 1) * w], ref['out'][:, i * w:(i + 1) * w]):.1e}/{rel(got['dx'][:, i * w:(i + 1) * w], ref['dx'][:, i * w:(i + 1) * w]):.1e}" for i in range(S)))
    print(f'[res2 train chain scale={S} B={B} T={T} dil={dil}] worst rel-L2 vs the per-chunk path: ' + ', '.join(f'{k} {v:.1e}' for k, v in worst.items()))
    fwd_tol, bwd_tol = {2: (2e-6, 2e-4), 3: (1e-4, 5e-3)}.get(S, (3e-3, 5e-2))
    for kind, e in worst.items():
        assert e < (fwd_tol if kind in ('out', 'run_mean', 'run_var') else bwd_tol), (kind, e)


@pytest.mark.parametrize('S,B,T,dil', [(2, 6, 100, 2), (2, 7, 298, 3), (3, 5, 67, 4), (3, 9, 298, 2)])
def test_res2_chain_one_launch_per_direction_vs_float64_amp_emulation(N, amp, S, B, T, dil):
    """vp_res2_train_fwd / _bwd against an INDEPENDENT reference: float64 autograd over the oracle's Res2NetBlock (ecapa_tdnn.py:11-47) with
    the mixed-precision roundings emulated (conv operands and the conv's output gradient rounded to bf16, everything else exact:
    oracle/models.py AMP).  Scale 2 and 3, where the chain does not amplify a flipped rounding (see the test above): absolute bounds as a
    fraction of each tensor's largest magnitude -- outputs 3e-4 (measured <= 1.0e-4), gradients 4e-3 (measured <= 1.4e-3; scale 2: 3.6e-4)."""
    from ppvector.train.functions import Res2Fn
    w = 64
    g = torch.Generator().manual_seed(B * 77 + T)
    x0 = torch.randn(B * T, S * w, generator=g)
    dout = torch.randn(B * T, S * w, generator=g)
    base = []
    for i in range(S - 1):
        base += [torch.randn(w, w, 3, generator=g) / (3 * w) ** 0.5, torch.randn(w, generator=g) * 0.3,
                 torch.randn(w, generator=g) * 0.2 + 1.0, torch.randn(w, generator=g) * 0.2, torch.zeros(w), torch.ones(w)]
    # float64 reference, (B, C, T) layout
    pr = {}
    for i in range(S - 1):
        for k, nm in enumerate(('conv.conv.weight', 'conv.conv.bias', 'norm.norm.weight', 'norm.norm.bias')):
            pr[f'blocks.{i}.{nm}'] = base[6 * i + k].double().requires_grad_()
    xr = x0.double().view(B, T, S * w).transpose(1, 2).contiguous().requires_grad_()
    om.AMP = True
    try:
        out_r = om.res2net_block(xr, pr, '', S, dil, training=True)
        out_r.backward(dout.double().view(B, T, S * w).transpose(1, 2))
    finally:
        om.AMP = False
    ref = {'out': out_r.detach().transpose(1, 2).reshape(B * T, S * w), 'dx': xr.grad.transpose(1, 2).reshape(B * T, S * w)}
    for i in range(S - 1):
        for k, nm in enumerate(('conv.conv.weight', 'conv.conv.bias', 'norm.norm.weight', 'norm.norm.bias')):
            ref[f'{("dW", "dbias", "dgamma", "dbeta")[k]}{i}'] = pr[f'blocks.{i}.{nm}'].grad
    # the one-launch-per-direction kernels
    params = [t.clone().cuda() for t in base]
    for i in range(S - 1):
        for k in range(4):
            params[6 * i + k].requires_grad_()
    x = x0.clone().cuda().requires_grad_()
    cfg = dict(B=B, T=T, scale=S, dilation=dil, momentum=0.9, eps=1e-5)
    out = Res2Fn.apply(x, cfg, *params)
    out.backward(dout.cuda())
    torch.cuda.synchronize()
    assert N.lib().vp_grid_barrier_status(N.ctx(x.device)) == 0, 'a grid barrier gave up waiting'
    got = {'out': out.detach(), 'dx': x.grad}
    for i in range(S - 1):
        for k, nm in enumerate(('dW', 'dbias', 'dgamma', 'dbeta')):
            got[f'{nm}{i}'] = params[6 * i + k].grad
    worst = {}
    for k, r in ref.items():
        e = (got[k].double().cpu() - r).abs().max().item() / max(r.abs().max().item(), 1e-30)
        kind = k.rstrip('0123456789')
        worst[kind] = max(worst.get(kind, 0.0), e)
    print(f'[res2 train chain vs float64 AMP emulation, scale={S} B={B} T={T} dil={dil}] max |err| / max |ref|: ' + ', '.join(f'{k} {v:.1e}' for k, v in worst.items()))
    for kind, e in worst.items():
        assert e < (3e-4 if kind == 'out' else 4e-3), (kind, e)


def test_block_outputs_written_as_bf16_by_their_producer_change_nothing(N, amp, monkeypatch):
    """enable_amp, ECAPA at >= 4096 rows: the SE-Res2 block outputs reach the next block's tdnn1 and the MFA layer as bf16 operands.
    By default the kernel that produces them (vp_se_scale_residual_shadow) also writes the bf16 copy into its column slice of the MFA
    operand; VPMI_NO_SHADOW=1 converts afterwards (x.to(bfloat16) per consumer + three strided copies).  Same rounding of the same
    values: loss and every parameter gradient must be bit-identical.  (Both runs with the six VPMI_*_F32* switches set: since round 4 the
    default keeps tdnn1's / the Res2 chain's / tdnn2's outputs, the residual, the block outputs and the MFA output as bf16 ONLY -- a different rounding, compared with
    this form in the second half of the test with the f32 engine's step as the yardstick: the bf16-only form must be no further from it
    than 1.5x what the form with f32 activations is (measured 1.14x on the embeddings, 1.0x on the gradient; test_ecapa_amp_operand_levels_agree_at_bench_scale explains that floor).)"""
    import ppvector
    from ppvector.models.ecapa_tdnn import EcapaTdnn
    from ppvector.train.ecapa_train import ecapa_forward_train
    torch.manual_seed(11)
    m0 = EcapaTdnn(80).cuda().train()
    state = {k: v.clone() for k, v in m0.state_dict().items()}
    x = torch.randn(16, 298, 80, device='cuda')
    g = torch.randn(16, 192, device='cuda')
    for name in ('VPMI_SE_F32', 'VPMI_MFA_F32_OUT', 'VPMI_TDNN1_F32_OUT', 'VPMI_RES2_F32_OUT', 'VPMI_BLOCK0_F32_OUT', 'VPMI_BLOCK0_F32_OPS'):
        monkeypatch.setenv(name, '1')

    def run(no_shadow):
        if no_shadow:
            monkeypatch.setenv('VPMI_NO_SHADOW', '1')
        else:
            monkeypatch.delenv('VPMI_NO_SHADOW', raising=False)
        m0.load_state_dict(state)
        for p in m0.parameters():
            p.grad = None
        emb = ecapa_forward_train(m0, x)
        emb.backward(g)
        torch.cuda.synchronize()
        return emb.detach().clone(), {k: p.grad.clone() for k, p in m0.named_parameters()}

    e0, g0 = run(True)
    e1, g1 = run(False)
    worst = max((g1[k] - g0[k]).abs().max().item() for k in g0)
    print(f'[bf16 shadows] embeddings identical: {torch.equal(e0, e1)}; worst |gradient difference| {worst:.1e}')
    assert torch.equal(e0, e1)
    assert all(torch.equal(g0[k], g1[k]) for k in g0)
    for name in ('VPMI_SE_F32', 'VPMI_MFA_F32_OUT', 'VPMI_TDNN1_F32_OUT', 'VPMI_RES2_F32_OUT', 'VPMI_BLOCK0_F32_OUT', 'VPMI_BLOCK0_F32_OPS'):
        monkeypatch.delenv(name)
    e2, g2 = run(False)                                   # the default: activations between the GEMMs as bf16 only
    ppvector.set_train_amp(False)                         # yardstick: the f32 engine's step on the same batch
    try:
        ex, gx = run(False)
    finally:
        ppvector.set_train_amp(True)

    def whole(ga, gb):
        num = sum(float(((ga[k] - gb[k]).double() ** 2).sum()) for k in gb) ** 0.5
        return num / sum(float((gb[k].double() ** 2).sum()) for k in gb) ** 0.5
    print(f'[bf16-only activations] vs the f32 step: embeddings rel-L2 f32 activations {rel(e1, ex):.2e} / bf16 only {rel(e2, ex):.2e}; '
          f'whole-gradient rel-L2 {whole(g1, gx):.2e} / {whole(g2, gx):.2e}; bf16 only vs f32 activations: {rel(e2, e1):.2e} / {whole(g2, g1):.2e}')
    assert rel(e2, ex) < 1.5 * rel(e1, ex) + 2e-3 and whole(g2, gx) < 1.5 * whole(g1, gx)


def test_time_statistics_from_the_convs_fused_sums_stay_as_close_to_the_f32_step(N, monkeypatch):
    """enable_amp only: SEBlock's squeeze mean (ecapa_tdnn.py:66-71) and ASP's context mean / std (pooling.py:97-104) of a TDNNBlock
    output y = BN(z) come from the conv's fused per-utterance sums of z and the layer's scale / shift (vp_moments_finalize_affine)
    instead of a pass over y (VPMI_NO_TSUMS=1 keeps the pass).  The sums are of the f32 accumulators, the pass reads the bf16-stored z:
    the two differ by the storage rounding (1e-4 on a mean), which the chain of bf16 operand roundings behind it amplifies like any
    other perturbation of a mixed-precision step.  Yardstick: the f32 engine's step on the same batch (exact statistics; it never
    takes the shortcut: measured 3e-5 on its embeddings, too much for the parity instrument) -- the shortcut must be no further
    from it than the pass is (25 % slack for the spread of that figure)."""
    import ppvector
    from ppvector.models.ecapa_tdnn import EcapaTdnn
    from ppvector.train.ecapa_train import ecapa_forward_train
    torch.manual_seed(5)
    m0 = EcapaTdnn(80).cuda().train()
    state = {k: v.clone() for k, v in m0.state_dict().items()}
    B, T = 16, 298
    x = torch.randn(B, T, 80, device='cuda')
    g = torch.randn(B, 192, device='cuda')

    def run(use_amp, no_tsums):
        ppvector.set_train_amp(use_amp)
        try:
            if no_tsums:
                monkeypatch.setenv('VPMI_NO_TSUMS', '1')
            else:
                monkeypatch.delenv('VPMI_NO_TSUMS', raising=False)
            m0.load_state_dict(state)
            for p in m0.parameters():
                p.grad = None
            emb = ecapa_forward_train(m0, x)
            emb.backward(g)
            torch.cuda.synchronize()
            return emb.detach().clone(), {k: p.grad.clone() for k, p in m0.named_parameters()}
        finally:
            ppvector.set_train_amp(False)

    def whole(a, b):
        return (sum((a[k].double() - b[k].double()).pow(2).sum().item() for k in b) / sum(b[k].double().pow(2).sum().item() for k in b)) ** 0.5

    ex, gx = run(False, False)
    ex2, gx2 = run(False, True)
    assert torch.equal(ex, ex2) and all(torch.equal(gx[k], gx2[k]) for k in gx), 'the f32 engine must not take the shortcut'
    e0, g0 = run(True, True)
    e1, g1 = run(True, False)
    print(f'[fused time sums] vs the f32 step: embeddings rel-L2 pass {rel(e0, ex):.2e} / fused sums {rel(e1, ex):.2e}; whole-gradient '
          f'rel-L2 pass {whole(g0, gx):.2e} / fused sums {whole(g1, gx):.2e}; fused sums vs pass: {rel(e1, e0):.2e} / {whole(g1, g0):.2e}')
    assert rel(e1, ex) < 1.25 * rel(e0, ex) + 1e-3 and whole(g1, gx) < 1.25 * whole(g0, gx)


@pytest.mark.parametrize('B,T,C,H', [(5, 40, 512, 128), (32, 120, 256, 64), (3, 33, 64, 16)])
def test_se_block_with_its_dense_layers_in_one_launch_vs_float64_autograd(N, monkeypatch, B, T, C, H):
    """SEBlock + block residual (ecapa_tdnn.py:50-82, 139-141) as SEBlockFn with the two dense layers in csrc/se_train.hip (one launch
    forward, two backward) against float64 autograd of the same formulas, f32 engine: output 2e-6, gradients 2e-5; and against the
    previous form of the layers (two 1x1 ConvBlocks, VPMI_SE_DENSE_UNFUSED=1) under enable_amp: same operand roundings, 2e-3."""
    import ppvector
    from ppvector.train.functions import SEBlockFn
    g = torch.Generator().manual_seed(B * 100 + C)
    h0 = torch.randn(B * T, C, generator=g).cuda()
    r0 = torch.randn(B * T, C, generator=g).cuda()
    w1 = (torch.randn(H, C, 1, generator=g) / C ** 0.5).cuda()
    b1 = (torch.randn(H, generator=g) * 0.1).cuda()
    w2 = (torch.randn(C, H, 1, generator=g) / H ** 0.5).cuda()
    b2 = (torch.randn(C, generator=g) * 0.1).cuda()
    dout = torch.randn(B * T, C, generator=g).cuda()

    def run():
        ts = [t.clone().requires_grad_() for t in (h0, r0, w1, b1, w2, b2)]
        out = SEBlockFn.apply(ts[0], ts[1], ts[2], ts[3], ts[4], ts[5], B, T)
        out.backward(dout)
        torch.cuda.synchronize()
        return [out.detach()] + [t.grad for t in ts]

    got = run()
    hd, rd, w1d, b1d, w2d, b2d = (t.double().clone().requires_grad_() for t in (h0, r0, w1, b1, w2, b2))
    mean = hd.reshape(B, T, C).mean(1)
    a = torch.relu(mean @ w1d[:, :, 0].t() + b1d)
    s = torch.sigmoid(a @ w2d[:, :, 0].t() + b2d)
    ref_out = (hd.reshape(B, T, C) * s[:, None, :]).reshape(B * T, C) + rd
    ref_out.backward(dout.double())
    ref = [ref_out.detach(), hd.grad, rd.grad, w1d.grad, b1d.grad, w2d.grad, b2d.grad]
    names = ['out', 'd h', 'd res', 'd W1', 'd b1', 'd W2', 'd b2']
    errs = {n: rel(x, y) for n, x, y in zip(names, got, ref)}
    print(f'[se dense B={B} T={T} C={C} H={H}] f32 vs float64: ' + ', '.join(f'{k} {v:.1e}' for k, v in errs.items()))
    assert errs['out'] < 2e-6 and all(v < 2e-5 for v in errs.values()), errs
    ppvector.set_train_amp(True)
    try:
        fused = run()
        monkeypatch.setenv('VPMI_SE_DENSE_UNFUSED', '1')
        unfused = run()
    finally:
        ppvector.set_train_amp(False)
        monkeypatch.delenv('VPMI_SE_DENSE_UNFUSED', raising=False)
    errs = {n: rel(x, y) for n, x, y in zip(names, fused, unfused)}
    print(f'[se dense B={B} T={T} C={C} H={H}] enable_amp, one launch vs two 1x1 ConvBlocks: ' + ', '.join(f'{k} {v:.1e}' for k, v in errs.items()))
    assert all(v < 2e-3 for v in errs.values()), errs


@pytest.mark.parametrize('B,T,C', [(3, 5000, 32), (4, 2048, 128), (2, 1500, 256), (5, 1024, 12)])
def test_se_scale_backward_over_many_positions_vs_float64(N, B, T, C):
    """SEScale (out = x * s[b] + res: the SE gate of ResNetSE / ERes2Net on (B, T*F', C) feature maps, resnet_se.py:60-75) with thousands of
    positions per utterance: the backward spreads the positions over the chip (vp_scale_rows_bwd_ws_f32, per-chunk partial sums) instead
    of one workgroup per 64 channels of an utterance.  d x = d y * s, d s = sum_positions d y * x: against float64, 2e-6."""
    from ppvector.train.functions import SEScale
    g = torch.Generator().manual_seed(T + C)
    x = torch.randn(B * T, C, generator=g).cuda().requires_grad_()
    s = torch.rand(B, C, generator=g).cuda().requires_grad_()
    res = torch.randn(B * T, C, generator=g).cuda().requires_grad_()
    dy = torch.randn(B * T, C, generator=g).cuda()
    out = SEScale.apply(x, s, res, B, T)
    out.backward(dy)
    torch.cuda.synchronize()
    xd, sd, dd = x.detach().double().reshape(B, T, C), s.detach().double(), dy.double().reshape(B, T, C)
    e_dx = rel(x.grad, (dd * sd[:, None, :]).reshape(B * T, C))
    e_ds = rel(s.grad, (dd * xd).sum(1))
    print(f'[se scale bwd B={B} T={T} C={C}] d x {e_dx:.1e}, d s {e_ds:.1e}, d res exact: {torch.equal(res.grad, dy)}')
    assert e_dx < 2e-6 and e_ds < 2e-6 and torch.equal(res.grad, dy)


@pytest.mark.parametrize('B,T,C,tstp', [(3, 5000, 32, False), (4, 2048, 128, True), (2, 1100, 256, False)])
def test_time_statistics_over_many_frames_vs_float64(N, B, T, C, tstp):
    """TimeStats ([mean | std] over the frames) on utterances of thousands of positions (the SE squeeze of ResNetSE on (B, T*F', C) maps):
    the frames are spread over the chip (vp_time_stats_ws_f32).  Against float64: 2e-6; backward unchanged."""
    from ppvector.train.functions import TimeStats
    g = torch.Generator().manual_seed(T)
    x = (torch.randn(B * T, C, generator=g) * 2 + 3).cuda().requires_grad_()
    st = TimeStats.apply(x, B, T, tstp)
    xd = x.detach().double().reshape(B, T, C)
    mean = xd.mean(1)
    std = (xd.var(1, unbiased=True) + 1e-8).sqrt() if tstp else xd.var(1, unbiased=False).clamp_min(1e-12).sqrt()
    e1, e2 = rel(st[:, :C], mean), rel(st[:, C:], std)
    print(f'[time stats B={B} T={T} C={C} tstp={tstp}] mean {e1:.1e}, std {e2:.1e}')
    assert e1 < 2e-6 and e2 < 2e-6


@pytest.mark.parametrize('B,T,Cc', [(3, 47, 96), (5, 298, 192), (2, 160, 64)])
def test_statistics_kernels_on_bf16_stored_tensors_vs_float64(N, B, T, Cc):
    """Round 4: the training kernels that read activations STORED as bf16 (the values are exactly the bf16 numbers, so the float64
    reference over those numbers is exact up to the kernels' own f32 arithmetic): vp_asp_softmax_stats_l16 and vp_attn_stats_bwd_e16 (bf16
    logits, x f32 or bf16; d e written as bf16: compared at bf16 resolution), vp_time_stats_bwd_add_x16, vp_utt_sums_b16, vp_utt_dot_x16,
    vp_affine_rows_b16_b16 / _f32_b16."""
    lib, ctx = N.lib(), N.ctx(torch.device('cuda', 0))
    g = torch.Generator().manual_seed(B * 31 + T)
    bf = lambda t: t.to(torch.bfloat16)
    x16 = bf(torch.randn(B * T, Cc, generator=g))
    e16 = bf(torch.randn(B * T, Cc, generator=g) * 2)
    dp = torch.randn(B, 2 * Cc, generator=g)
    dpd, x16d = dp.cuda(), x16.cuda()                    # (device copies kept alive in names: a temporary's memory may be reused before its kernel runs)
    for x_is16 in (False, True):
        xs = x16 if x_is16 else torch.randn(B * T, Cc, generator=g)
        x = xs.double().view(B, T, Cc).requires_grad_()
        e = e16.double().view(B, T, Cc).requires_grad_()
        al = torch.softmax(e, dim=1)
        mu = (al * x).sum(1)
        sd = torch.sqrt(((al * (x - mu[:, None]) ** 2).sum(1)).clamp(min=1e-12))
        pooled_ref = torch.cat([mu, sd], 1)
        (pooled_ref * dp.double()).sum().backward()
        xd, ed = xs.cuda(), e16.cuda()
        pooled = torch.empty(B, 2 * Cc, device='cuda')
        dt = N.VP_BF16 if x_is16 else N.VP_F32
        N.check(lib.vp_asp_softmax_stats_l16(ctx, ed.data_ptr(), xd.data_ptr(), dt, Cc, 0, B, T, Cc, 1e-12, pooled.data_ptr(), N.stream_ptr()), ctx)
        assert rel(pooled, pooled_ref.detach()) < 5e-6
        de = torch.empty(B * T, Cc, dtype=torch.bfloat16, device='cuda')
        dx = torch.empty(B * T, Cc, device='cuda')
        N.check(lib.vp_attn_stats_bwd_e16(ctx, ed.data_ptr(), xd.data_ptr(), dt, Cc, pooled.data_ptr(), dpd.data_ptr(), B, T, Cc, 1e-12,
                                          de.data_ptr(), dx.data_ptr(), Cc, N.stream_ptr()), ctx)
        assert rel(dx.view(B, T, Cc), x.grad) < 3e-5
        assert rel(de.float().view(B, T, Cc), e.grad) < 4e-3           # d e leaves as bf16: 2^-9 per element
    # context statistics' backward over a bf16 x, added to another gradient
    x = x16.double().view(B, T, Cc).requires_grad_()
    m0 = x.mean(1)
    s0 = torch.sqrt((((x - m0[:, None]) ** 2).mean(1)).clamp(min=1e-12))
    ds = torch.randn(B, 2 * Cc, generator=g)
    (torch.cat([m0, s0], 1) * ds.double()).sum().backward()
    add = torch.randn(B * T, Cc, generator=g)
    stats = torch.cat([m0, s0], 1).detach().float().cuda()
    out, dsd_ = add.clone().cuda(), ds.cuda()
    N.check(lib.vp_time_stats_bwd_add_x16(ctx, x16d.data_ptr(), Cc, stats.data_ptr(), dsd_.data_ptr(), B, T, Cc, 1e-12, 0,
                                          out.data_ptr(), Cc, out.data_ptr(), Cc, N.stream_ptr()), ctx)
    assert rel(out.view(B, T, Cc), x.grad + add.double().view(B, T, Cc)) < 2e-6
    # per-utterance sums of a bf16 tensor, per-utterance dot of an f32 gradient with a bf16 tensor
    sums = torch.empty(B, Cc, device='cuda')
    N.check(lib.vp_utt_sums_b16(ctx, x16d.data_ptr(), Cc, B, T, Cc, sums.data_ptr(), N.stream_ptr()), ctx)
    assert (sums.double().cpu() - x16.double().view(B, T, Cc).sum(1)).abs().max().item() < 2e-6 * x16.double().abs().view(B, T, Cc).sum(1).max().item()
    dy = torch.randn(B * T, Cc, generator=g)
    dsd, dyd = torch.empty(B, Cc, device='cuda'), dy.cuda()
    N.check(lib.vp_utt_dot_x16(ctx, dyd.data_ptr(), x16d.data_ptr(), B, T, Cc, dsd.data_ptr(), N.stream_ptr()), ctx)
    want = (dy.double() * x16.double()).view(B, T, Cc).sum(1)
    assert (dsd.double().cpu() - want).abs().max().item() < 2e-6 * (dy.double() * x16.double()).abs().view(B, T, Cc).sum(1).max().item()
    # BatchNorm apply pass with a bf16 result: z bf16 or f32 in, bf16 out = the f32 result rounded once
    scale, shift = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
    zf = torch.randn(B * T, Cc, generator=g)
    scd, shd = scale.cuda(), shift.cuda()
    for z in (x16, zf):
        y, zd = torch.empty(B * T, Cc, dtype=torch.bfloat16, device='cuda'), z.cuda()
        fn = lib.vp_affine_rows_b16_b16 if z.dtype == torch.bfloat16 else lib.vp_affine_rows_f32_b16
        N.check(fn(ctx, zd.data_ptr(), Cc, scd.data_ptr(), shd.data_ptr(), B * T, Cc, y.data_ptr(), Cc, 0, N.stream_ptr()), ctx)
        want = (z.float() * scale + shift)                                # the kernel's f32 fma may differ from this by one f32 ulp before rounding
        assert (y.float().cpu() - want).abs().max().item() <= 2 ** -8 * want.abs().max().item()
        assert (y.cpu() != want.to(torch.bfloat16)).float().mean().item() < 2e-3     # same bf16 number except where the f32 values straddle a rounding boundary


def test_enable_amp_trains_like_f32(N, tmp_path):
    """VERDICT r04 ("parity first", item 1): each kernel of the enable_amp step is tight against float64, but the whole step sits 17 % (gradient
    rel-L2) from the f32 step at random init -- does it TRAIN?  PPVectorTrainer (the reference-shaped trainer, trainer.py:202-274 upstream)
    on synthetic separable speakers (tools/amp_convergence.py: pitch + formant voices, noise, random crops, SpecAugment), the reference's
    ecapa_tdnn.yml settings, enable_amp False vs True from the same seed and lists.  Shortened here (120 steps of 128 utterances); the
    300 x 256 run is `python tools/amp_convergence.py` -> profiles/r05_amp_convergence.log.  Asserted band: both runs learn (tail
    accuracy >= 0.9, tail loss below a third of the first logged loss), the AMP tail loss within 25 % + 0.05 of the f32 one, EERs within
    0.03 of each other."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools'))
    import amp_convergence as ac
    import ppvector
    root = str(tmp_path)
    n_spk, batch, epochs = 32, 128, 6
    spe = ac.build_dataset(root, n_spk, train_files=8, steps_per_epoch=20, batch=batch)
    try:
        res = {amp: ac.run(root, n_spk, batch, epochs, amp) for amp in (False, True)}
    finally:
        ppvector.set_train_amp(False)
        ppvector.set_fused_grid_kernels(True)
    c = ac.compare(res[False], res[True])
    for amp in (False, True):
        r = res[amp]
        print(f'[amp trains] enable_amp={amp}: {r["steps"]} steps ({epochs} x {spe}), {r["seconds"]:.1f} s, EER {r["eer"]:.4f}, minDCF {r["min_dcf"]:.4f}; '
              'loss curve ' + ' '.join(f'{p[1]:.3f}' for p in r['curve']))
        assert r['capture_error'] is None and r['barrier_faults'] == 0, (r['capture_error'], r['barrier_faults'])
    print('[amp trains] summary', c)
    for amp in (False, True):
        first = res[amp]['curve'][0][1]
        tl, ta = (c['tail_loss_amp'], c['tail_acc_amp']) if amp else (c['tail_loss_f32'], c['tail_acc_f32'])
        assert ta >= 0.9 and tl < first / 3, (amp, first, tl, ta)
    assert c['tail_loss_amp'] <= 1.25 * c['tail_loss_f32'] + 0.05, c
    assert abs(c['eer_amp'] - c['eer_f32']) <= 0.03, c


def test_weight_prep_launch_writes_every_bf16_panel(N, monkeypatch):
    """vp_prep_weights_bf16 (one launch for all wide layers of a step: W bf16 and W^T bf16, column slices allowed) against torch's own
    conversion, bit for bit; and the registry ConvBlock reads them from goes stale with the weights epoch (an optimiser step)."""
    from ppvector.train import functions as Fn
    g = torch.Generator().manual_seed(9)
    ws = [torch.randn(512, 512, 1, generator=g).cuda(), torch.randn(1536, 1536, 1, generator=g).cuda(), torch.randn(128, 4608, 1, generator=g).cuda(),
          torch.randn(1536, 128, 1, generator=g).cuda(), torch.randn(70, 200, 1, generator=g).cuda()]
    items = [(ws[0], 0, 512), (ws[1], 0, 1536), (ws[2], 0, 1536), (ws[3], 0, 128), (ws[4], 8, 100)]
    Fn.prep_weights_bf16(items)
    torch.cuda.synchronize()
    for w, c0, nc in items:
        p = Fn._panels16(w, c0, nc)
        assert p is not None
        ref = w[:, c0:c0 + nc, 0].to(torch.bfloat16)
        assert torch.equal(p[0], ref) and torch.equal(p[1], ref.t().contiguous()), (tuple(w.shape), c0, nc)
    assert Fn._panels16(ws[2], 0, 4608) is None                      # another slice of the same parameter was never prepared
    N.bump_weights_epoch()
    assert all(Fn._panels16(w, c0, nc) is None for w, c0, nc in items)


def test_conv_se_tail_as_one_tape_entry_changes_nothing(N, amp, monkeypatch):
    """ConvSEFn (tdnn2 + SE gate + residual of an SE-Res2 block as one tape entry: the BatchNorm output h and the SE block's input gradient
    dh are formed on the fly inside the passes that read them, never stored) against the two entries it replaces (ConvBlock + SEBlockFn,
    VPMI_SE_TAIL_UNFUSED=1): loss, embeddings, every parameter gradient and every running statistic of an ECAPA-TDNN training
    forward + backward at 16 x 298 frames, bit for bit."""
    from ppvector.loss.aamloss import AAMLoss
    from ppvector.models.ecapa_tdnn import EcapaTdnn
    from ppvector.models.fc import SpeakerIdentification
    from ppvector.train import functions as Fn
    g = torch.Generator().manual_seed(41)
    x = (torch.randn(16, 298, 80, generator=g) * 2).cuda()
    y = torch.randint(0, 11, (16,), generator=g).cuda()

    def run(unfused):
        if unfused:
            monkeypatch.setenv('VPMI_SE_TAIL_UNFUSED', '1')
        else:
            monkeypatch.delenv('VPMI_SE_TAIL_UNFUSED', raising=False)
        calls = {'n': 0}
        orig = Fn.ConvSEFn.usable

        def counted(*a, **k):
            r = orig(*a, **k)
            calls['n'] += int(bool(r))
            return r
        monkeypatch.setattr(Fn.ConvSEFn, 'usable', staticmethod(counted))
        m = EcapaTdnn(80, embd_dim=192, pooling_type='ASP', channels=[512, 512, 512, 512, 1536])
        m.load_state_dict(om.ecapa_params(80, seed=21))
        head = SpeakerIdentification(192, 11)
        head.load_state_dict({'weight': om.head_params(192, 11, seed=22)})
        model = torch.nn.Sequential(m, head).cuda().train()
        out = model(x)
        loss = AAMLoss(margin=0.2, scale=32)(out, y)
        loss.backward()
        torch.cuda.synchronize()
        monkeypatch.setattr(Fn.ConvSEFn, 'usable', staticmethod(orig))
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        bufs = {k: b.detach().clone() for k, b in model.named_buffers()}
        return float(loss), out['features'].detach().clone(), grads, bufs, calls['n']

    la, ea, ga, ba, na = run(False)
    lb, eb, gb, bb, nb = run(True)
    assert na == 3 and nb == 0, (na, nb)                      # the one-entry form ran for all three blocks / not at all
    assert la == lb and torch.equal(ea, eb)
    diff = [k for k in ga if not torch.equal(ga[k], gb[k])]
    assert not diff, diff[:5]
    assert all(torch.equal(ba[k], bb[k]) for k in ba)
    print(f'[conv + SE tail as one tape entry] loss {la:.6f}: {len(ga)} parameter gradients and {len(ba)} buffers bit-identical to the two-entry form')


def test_mfa_asp_as_one_tape_entry_stays_within_rounding_noise(N, amp, monkeypatch):
    """MfaAspFn (MFA TDNNBlock + AttentiveStatisticsPooling as one tape entry: the pooling layer's context-statistics gradient
    alpha[b, c] + beta[b, c] * y is added inside the MFA layer's two BatchNorm-backward passes instead of by a pass over the (B*T, 1536)
    tensors) against the two entries it replaces (VPMI_MFA_ASP_UNFUSED=1) at 56 x 298 frames: the forward is the same launches (loss,
    embeddings and running statistics bit-identical); the gradients differ by the rounding of alpha + beta * y against
    dmean / T + dstd / std * (y - mean) / T and by the bf16 roundings of dz that flip with it: measured 2.6e-6 on the parameters the term
    reaches first (MFA / ASP / head; bound 2e-4), 3.2e-3 on the whole gradient once three Res2 chains have amplified it (bound 1e-2; the
    mixed-precision step sits ~1e-1 from the f32 step)."""
    from ppvector.loss.aamloss import AAMLoss
    from ppvector.models.ecapa_tdnn import EcapaTdnn
    from ppvector.models.fc import SpeakerIdentification
    from ppvector.train import functions as Fn
    g = torch.Generator().manual_seed(43)
    x = (torch.randn(56, 298, 80, generator=g) * 2).cuda()
    y = torch.randint(0, 11, (56,), generator=g).cuda()

    def run(unfused):
        if unfused:
            monkeypatch.setenv('VPMI_MFA_ASP_UNFUSED', '1')
        else:
            monkeypatch.delenv('VPMI_MFA_ASP_UNFUSED', raising=False)
        calls = {'n': 0}
        orig = Fn._asp_backward

        def counted(ctx, dp, defer_ctx=False):
            calls['n'] += int(bool(defer_ctx))
            return orig(ctx, dp, defer_ctx)
        monkeypatch.setattr(Fn, '_asp_backward', counted)
        m = EcapaTdnn(80, embd_dim=192, pooling_type='ASP', channels=[512, 512, 512, 512, 1536])
        m.load_state_dict(om.ecapa_params(80, seed=21))
        head = SpeakerIdentification(192, 11)
        head.load_state_dict({'weight': om.head_params(192, 11, seed=22)})
        model = torch.nn.Sequential(m, head).cuda().train()
        out = model(x)
        loss = AAMLoss(margin=0.2, scale=32)(out, y)
        loss.backward()
        torch.cuda.synchronize()
        monkeypatch.setattr(Fn, '_asp_backward', orig)
        grads = {k: p.grad.detach().clone() for k, p in model.named_parameters()}
        bufs = {k: b.detach().clone() for k, b in model.named_buffers()}
        return float(loss), out['features'].detach().clone(), grads, bufs, calls['n']

    la, ea, ga, ba, na = run(False)
    lb, eb, gb, bb, nb = run(True)
    assert na == 1 and nb == 0, (na, nb)
    assert la == lb and torch.equal(ea, eb) and all(torch.equal(ba[k], bb[k]) for k in ba)
    worst, wk = 0.0, None
    for k in ga:
        d = rel(ga[k], gb[k])
        if d > worst:
            worst, wk = d, k
    whole = rel(torch.cat([ga[k].reshape(-1) for k in ga]), torch.cat([gb[k].reshape(-1) for k in ga]))
    tail = {k: rel(ga[k], gb[k]) for k in ga if k.startswith(('0.mfa.', '0.asp', '0.fc', '1.'))}      # first-order: the layers the term reaches directly
    print(f'[MFA + ASP as one tape entry] loss {la:.6f} (identical); whole gradient rel-L2 against the two-entry form {whole:.2e}; MFA / ASP / '
          f'head parameters worst {max(tail.values()):.2e}; worst tensor {worst:.2e} ({wk})')
    # the term itself is exact to f32 rounding (test_context_statistics_gradient_folded_into_the_bn_backward_vs_float64); what is bounded here
    # is how far that rounding travels: the MFA layer's own parameters see it first-order, the blocks upstream through bf16 roundings of dz
    # and ReLU masks that flip with it (the Res2 chain amplifies a 1e-7 change of a statistic to 1e-2 on a 64-element bias: DESIGN.md 7c)
    assert max(tail.values()) < 2e-4 and whole < 1e-2 and worst < 3e-2, (max(tail.values()), whole, worst, wk)


@pytest.mark.parametrize('B,T,C', [(56, 298, 1536), (5, 40, 512), (3, 23, 64)])
def test_context_statistics_gradient_folded_into_the_bn_backward_vs_float64(N, B, T, C):
    """vp_time_stats_bwd_coeffs + vp_col_sums_f32_b16_ctx + vp_bn_relu_bwd_dbias_b16_ctx (the pooling layer's context-statistics gradient
    alpha[b, c] + beta[b, c] * y added to d y inside the producing TDNNBlock's two BatchNorm-backward passes, y = bf16(z * scale + shift)
    re-formed from z) against float64: the coefficients against d/dy of [mean_t y | sqrt(max(var_t y, eps))] (pooling.py:97-104), the
    sums and d z against the formula over dy + that gradient."""
    lib, ctx = N.lib(), N.ctx(0)
    g = torch.Generator().manual_seed(B * 1000 + T)
    M = B * T
    z = torch.randn(M, C, generator=g).to(torch.bfloat16)
    dy = torch.randn(M, C, generator=g)
    bsc, bsh = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    mean, invstd = torch.randn(C, generator=g) * 0.1, torch.rand(C, generator=g) + 0.5
    gam = torch.randn(C, generator=g)
    dstats = torch.randn(B, 2 * C, generator=g)
    y = (z.float() * bsc + bsh).to(torch.bfloat16).double().reshape(B, T, C)        # (one f32 multiply-add, rounded to bf16: as the apply pass stores it)
    mu = y.mean(1)
    var = ((y - mu[:, None]) ** 2).mean(1)
    sd = var.clamp(min=1e-12).sqrt()
    stats = torch.cat([mu, sd], dim=1).float()
    be = torch.where(var > 1e-12, dstats[:, C:].double() / stats[:, C:].double() / T, torch.zeros_like(var))
    al = dstats[:, :C].double() / T - be * stats[:, :C].double()
    ab = torch.empty((2, B, C), device='cuda')
    sc_, dc_ = stats.cuda(), dstats.cuda()
    N.check(lib.vp_time_stats_bwd_coeffs(ctx, sc_.data_ptr(), dc_.data_ptr(), B, T, C, 1e-12, ab.data_ptr(), N.stream_ptr()), ctx)
    assert rel(ab[0], al) < 1e-6 and rel(ab[1], be) < 1e-6
    dye = dy.double().reshape(B, T, C) + al[:, None] + be[:, None] * y
    dye = dye.reshape(M, C)
    zh = (z.double() - mean.double()) * invstd.double()
    wsum = torch.stack([dye.sum(0), (dye * zh).sum(0)])
    zc, dyc, bscc, bshc, mc, ic, gc = z.cuda(), dy.cuda(), bsc.cuda(), bsh.cuda(), mean.cuda(), invstd.cuda(), gam.cuda()
    sums = torch.empty((2, C), device='cuda')
    ws = torch.empty(lib.vp_col_sums_workspace_bytes(M, C), dtype=torch.uint8, device='cuda')
    N.check(lib.vp_col_sums_f32_b16_ctx(ctx, dyc.data_ptr(), C, ab[0].data_ptr(), ab[1].data_ptr(), T, bscc.data_ptr(), bshc.data_ptr(),
                                        zc.data_ptr(), C, mc.data_ptr(), ic.data_ptr(), M, C, sums.data_ptr(), ws.data_ptr(), ws.numel(),
                                        N.stream_ptr()), ctx)
    torch.cuda.synchronize()
    scale = dye.abs().sum(0).max().item()
    assert (sums.double().cpu() - wsum).abs().max().item() <= 3e-6 * scale * (invstd.max().item() * 4), (sums.double().cpu() - wsum).abs().max().item()
    want = gam.double() * invstd.double() * (dye - wsum[0] / M - zh * wsum[1] / M) * (z.double() > 0)
    dz = torch.empty((M, C), dtype=torch.bfloat16, device='cuda')
    db = torch.empty(C, device='cuda')
    ws2 = torch.empty(lib.vp_bn_relu_bwd_dbias_workspace_bytes(M, C), dtype=torch.uint8, device='cuda')
    N.check(lib.vp_bn_relu_bwd_dbias_b16_ctx(ctx, dyc.data_ptr(), C, ab[0].data_ptr(), ab[1].data_ptr(), T, bscc.data_ptr(), bshc.data_ptr(),
                                             zc.data_ptr(), C, mc.data_ptr(), ic.data_ptr(), gc.data_ptr(), wsum.float().cuda().data_ptr(), M, C, 1,
                                             dz.data_ptr(), C, db.data_ptr(), ws2.data_ptr(), ws2.numel(), N.stream_ptr()), ctx)
    torch.cuda.synchronize()
    assert rel(dz.float(), want) < 3e-3                           # bf16 storage of dz: 2^-9 per element
    assert torch.equal(dz.cpu() == 0, (want == 0) | (dz.cpu() == 0)) and bool(((dz.cpu() == 0) | (z > 0)).all())
    assert (db.double().cpu() - want.sum(0)).abs().max().item() <= 1e-5 * want.abs().sum(0).max().item()
    print(f'[context-statistics gradient in the BN backward] B {B} T {T} C {C}: coeffs {rel(ab[0], al):.1e} / {rel(ab[1], be):.1e}, dz rel-L2 {rel(dz.float(), want):.1e}')


def test_statistics_pooling_backward_of_a_constant_channel_is_finite(N):
    """CAM++'s statistics pooling (campplus.py:24-30: [mean_t x | std_t x], unbiased, NO epsilon) over a ReLU output: a channel that is zero
    over all frames of an utterance has std = 0 and d sqrt(var) / d x is 0 / 0 there.  Round 5 found this NaN reaching Adam ~200 steps into
    a CAM++ training run (f32 and enable_amp alike; tools/train_dynamics_ab.py, profiles/r05_campplus_dynamics.log) while the oracle graph
    trained on: torch.std masks the singular point.  The kernel now returns that limit -- no contribution from the std -- and equals float64
    autograd of torch.std everywhere."""
    from ppvector.train.functions import TimeStats
    g = torch.Generator().manual_seed(17)
    B, T, C = 3, 37, 64
    x = torch.relu(torch.randn(B, T, C, generator=g))
    x[1, :, 5] = 0.0                      # dead for the whole utterance
    x[2, :, 9] = 1.25                     # constant, non-zero
    x[0, :, :4] = 0.0
    xd = x.double().clone().requires_grad_(True)
    want = torch.cat([xd.mean(1), xd.std(1, unbiased=True)], dim=-1)
    gs = torch.randn(B, 2 * C, generator=g)
    want.backward(gs.double())
    xc = x.reshape(B * T, C).cuda().requires_grad_(True)
    got = TimeStats.apply(xc, B, T, True, 0.0)
    got.backward(gs.cuda())
    torch.cuda.synchronize()
    assert bool(torch.isfinite(xc.grad).all())
    assert rel(got, want.detach()) < 1e-6
    assert rel(xc.grad.reshape(B, T, C), xd.grad) < 1e-5
    assert torch.allclose(xc.grad.reshape(B, T, C)[1, :, 5].cpu(), (gs[1, 5] / T).expand(T).float(), rtol=1e-6, atol=0)   # only the mean's share reaches a dead channel

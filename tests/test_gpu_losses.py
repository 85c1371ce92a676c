"""GPU parity tests of the loss family next to AAMLoss (csrc/losses.hip through the reference-named classes):
golden values / gradients from the reference's own loss files (tests/golden/losses_ref.npz), then larger seeded cases
against autograd over the oracle in float64.  Run with -m gpu on an MI355X."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import losses as ol
from oracle import models as om

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def N():
    from ppvector import _native as N
    if not torch.cuda.is_available():
        pytest.fail('no GPU visible: these tests must run on an MI355X (no CPU fallback exists)')
    N.ctx(0)
    return N


def _build(name, **kw):
    from ppvector.loss import build_loss
    conf = types.SimpleNamespace(loss_conf={'loss': name, 'loss_args': kw})
    return build_loss(conf)


GOLDEN_CASES = [
    ('AMLoss', 'AMLoss', dict(margin=0.2, scale=30, label_smoothing=0.0), False),
    ('AMLoss_ls', 'AMLoss', dict(margin=0.35, scale=30, label_smoothing=0.1), False),
    ('ARMLoss', 'ARMLoss', dict(margin=0.2, scale=30, label_smoothing=0.0), False),
    ('ARMLoss_ls', 'ARMLoss', dict(margin=0.1, scale=20, label_smoothing=0.1), False),
    ('CELoss', 'CELoss', dict(label_smoothing=0.0), False),
    ('CELoss_ls', 'CELoss', dict(label_smoothing=0.2), False),
    ('SubCenterLoss', 'SubCenterLoss', dict(margin=0.2, scale=32, K=3), True),
    ('SubCenterLoss_easy_ls', 'SubCenterLoss', dict(margin=0.3, scale=32, easy_margin=True, K=3, label_smoothing=0.1), True),
    ('SphereFace2_C', 'SphereFace2', dict(margin=0.2, scale=32.0, lanbuda=0.7, t=3, margin_type='C'), False),
    ('SphereFace2_A', 'SphereFace2', dict(margin=0.15, scale=32.0, lanbuda=0.7, t=3, margin_type='A'), False),
]


@pytest.mark.parametrize('case', GOLDEN_CASES, ids=[c[0] for c in GOLDEN_CASES])
def test_losses_match_reference_golden(N, golden_dir, case):
    key, cls, kw, use_k = case
    g = np.load(os.path.join(golden_dir, 'losses_ref.npz'), allow_pickle=False)
    crit = _build(cls, **kw).cuda()
    labels = torch.from_numpy(g['labels']).cuda()
    ref_l, ref_g = float(g[key + '_loss']), g[key + '_dlogits']
    lg = torch.from_numpy(g['logits_k' if use_k else 'logits']).cuda()
    with torch.no_grad():                                                    # eval-time entry point
        l0 = crit({'features': None, 'logits': lg}, labels)
    assert abs(l0.item() - ref_l) < 2e-5 * max(1.0, abs(ref_l))
    lg = lg.clone().requires_grad_(True)                                     # training entry point: value + gradient
    l1 = crit({'features': None, 'logits': lg}, labels)
    (l1 * 2.0).backward()
    assert abs(l1.item() - ref_l) < 2e-5 * max(1.0, abs(ref_l))
    assert np.max(np.abs(lg.grad.cpu().numpy() / 2.0 - ref_g)) < 2e-5 * max(1.0, np.max(np.abs(ref_g)))


BIG_CASES = [
    ('AMLoss', dict(margin=0.25, scale=30, label_smoothing=0.05), lambda l, y, b: ol.am_loss(l, y, 0.25, 30.0, 0.05), 1),
    ('ARMLoss', dict(margin=0.25, scale=30, label_smoothing=0.05), lambda l, y, b: ol.arm_loss(l, y, 0.25, 30.0, 0.05), 1),
    ('CELoss', dict(label_smoothing=0.1), lambda l, y, b: ol.ce_loss(l, y, 0.1), 1),
    ('SubCenterLoss', dict(margin=0.3, scale=32, K=3, label_smoothing=0.1), lambda l, y, b: ol.subcenter_loss(l, y, 0.3, 32.0, False, 3, 0.1), 3),
    ('SubCenterLoss', dict(margin=0.2, scale=32, K=2, easy_margin=True), lambda l, y, b: ol.subcenter_loss(l, y, 0.2, 32.0, True, 2, 0.0), 2),
    ('SphereFace2', dict(margin=0.2, scale=32.0, lanbuda=0.7, t=3, margin_type='C'), lambda l, y, b: ol.sphereface2_loss(l, y, b, 0.2, 32.0, 0.7, 3, 'C'), 1),
    ('SphereFace2', dict(margin=0.15, scale=30.0, lanbuda=0.6, t=2, margin_type='A'), lambda l, y, b: ol.sphereface2_loss(l, y, b, 0.15, 30.0, 0.6, 2, 'A'), 1),
]


@pytest.mark.parametrize('case', BIG_CASES, ids=[f'{c[0]}-{i}' for i, c in enumerate(BIG_CASES)])
def test_losses_through_cosine_head_vs_oracle_autograd(N, case):
    """(B, 192) embeddings -> SpeakerIdentification(K) -> criterion, 2796 speakers (ragged against the 256-wide row walk):
    loss, d emb, d W and the SphereFace2 bias gradient against float64 autograd over the oracle."""
    from ppvector.models.fc import SpeakerIdentification
    cls, kw, orc, K = case
    B, D, Cc = 37, 192, 2796
    gen = torch.Generator().manual_seed(11)
    emb = torch.randn(B, D, generator=gen, dtype=torch.float64, requires_grad=True)
    W = torch.randn(D, Cc * K, generator=gen, dtype=torch.float64, requires_grad=True)
    labels = torch.randint(0, Cc, (B,), generator=gen)
    bias = torch.full((), 0.3, dtype=torch.float64, requires_grad=True)
    loss = orc(om.cosine_head(emb, W), labels, bias)
    loss.backward()
    head = SpeakerIdentification(D, Cc, K=K).cuda().train()
    with torch.no_grad():
        head.weight.copy_(W.detach().float())
    crit = _build(cls, **kw).cuda()
    if cls == 'SphereFace2':
        with torch.no_grad():
            crit.bias.fill_(0.3)
    ed = emb.detach().float().cuda().requires_grad_()
    lo = crit(head(ed), labels.cuda())
    lo.backward()

    def rel(a, b):
        a, b = a.double().cpu(), b.double().cpu()
        return ((a - b).norm() / b.norm().clamp(min=1e-30)).item()
    assert abs(lo.item() - loss.item()) < 1e-4 * max(1.0, abs(loss.item()))
    assert rel(ed.grad, emb.grad) < 1e-4 and rel(head.weight.grad, W.grad) < 1e-4
    if cls == 'SphereFace2':
        assert abs(crit.bias.grad.item() - bias.grad.item()) < 1e-4 * max(1.0, abs(bias.grad.item()))


def test_loss_argument_errors(N):
    lib, ctx = N.lib(), N.ctx(0)
    lg = torch.zeros(2, 6, device='cuda')
    lab = torch.zeros(2, dtype=torch.int64, device='cuda')
    out = torch.zeros(4, device='cuda')
    # K > 1 is only meaningful for the sub-centre loss; unknown kinds are refused
    assert lib.vp_margin_ce_fwd(ctx, lg.data_ptr(), lab.data_ptr(), 2, 3, 2, N.VP_LOSS_AM, 0.2, 30.0, 0.0, 0, out.data_ptr(),
                                out[1:].data_ptr(), None) == N.VP_EINVAL
    assert lib.vp_margin_ce_fwd(ctx, lg.data_ptr(), lab.data_ptr(), 2, 6, 1, 9, 0.2, 30.0, 0.0, 0, out.data_ptr(),
                                out[1:].data_ptr(), None) == N.VP_EINVAL
    with pytest.raises(NotImplementedError):
        _build('TripletAngularMarginLoss')
    with pytest.raises(AttributeError):
        _build('NoSuchLoss')
    with pytest.raises(N.VpmiError):
        _build('AMLoss')({'features': None, 'logits': torch.zeros(2, 6)}, torch.zeros(2, dtype=torch.int64))


@pytest.mark.parametrize('case', [('cos_b2', 'Cosine', 2), ('lin_b0', 'Linear', 0), ('lin_b1', 'Linear', 1)], ids=lambda c: c[0])
def test_head_variants_match_reference_golden_and_autograd(N, golden_dir, case):
    """SpeakerIdentification with DenseLayer('batchnorm') blocks / the 'Linear' output (fc.py:6-87): the reference's state
    dict loads unchanged; eval and train-mode logits against the golden file; gradients and BatchNorm running statistics
    against float64 autograd over the oracle."""
    from ppvector.models.fc import SpeakerIdentification
    tag, ctype, nb = case
    g = np.load(os.path.join(golden_dir, 'head_variants_ref.npz'), allow_pickle=False)
    p = om.classifier_params(24, 9, ctype, 1, nb, 16, seed=1002)
    head = SpeakerIdentification(24, 9, classifier_type=ctype, num_blocks=nb, inter_dim=16)
    head.load_state_dict(p)                                              # strict: same keys as the reference's module
    head = head.cuda().eval()
    emb = torch.from_numpy(g['emb']).cuda()
    with torch.no_grad():
        ev = head(emb)['logits']
    assert np.max(np.abs(ev.cpu().numpy() - g[tag + '_eval'])) < 2e-5 * max(1.0, np.max(np.abs(g[tag + '_eval'])))
    head.train()
    ed = emb.clone().requires_grad_()
    out = head(ed)
    assert out['features'] is ed
    assert np.max(np.abs(out['logits'].detach().cpu().numpy() - g[tag + '_train'])) < 5e-5 * max(1.0, np.max(np.abs(g[tag + '_train'])))
    gl = torch.randn(out['logits'].shape, generator=torch.Generator().manual_seed(5), dtype=torch.float64)
    out['logits'].backward(gl.float().cuda())
    pd = {k: v.double().requires_grad_(v.dtype.is_floating_point and not k.endswith(('_mean', '_variance'))) for k, v in p.items()}
    e64 = torch.from_numpy(g['emb']).double().requires_grad_()
    stats = {}
    om.classifier_head(e64, pd, ctype, nb, training=True, stats_out=stats).backward(gl)

    def rel(a, b):
        a, b = a.double().cpu(), b.double().cpu()
        return ((a - b).norm() / b.norm().clamp(min=1e-30)).item()
    assert rel(ed.grad, e64.grad) < 1e-4
    for k, v in head.named_parameters():
        if pd[k].grad is not None and pd[k].grad.abs().max() > 1e-9:
            assert rel(v.grad, pd[k].grad) < 2e-4, k
    for i in range(nb):                                                   # Paddle momentum 0.9, biased batch variance
        pre = f'blocks.{i}.nonlinear.batchnorm.'
        mean, var = stats[pre]
        bn = head.blocks[i].nonlinear.batchnorm
        assert rel(bn._mean, 0.9 * p[pre + '_mean'].double() + 0.1 * mean) < 1e-5
        assert rel(bn._variance, 0.9 * p[pre + '_variance'].double() + 0.1 * var) < 1e-5

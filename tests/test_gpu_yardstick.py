"""Yardstick, not parity: the SAME network run by an eager framework on the SAME GPU -- the oracle's PyTorch graphs (oracle/models.py,
the restatement of ppvector/models/ecapa_tdnn.py:245-276, fc.py:41-53, aamloss.py:28-47) moved to the MI355X and driven by
PyTorch-ROCm (MIOpen / hipBLASLt kernels, bf16 autocast) -- beside this engine on the same inputs.  The reference itself runs on
PaddlePaddle, which cannot be installed here; an eager framework over vendor kernels is the closest stand-in for "the reference on
this GPU".  The numbers are printed (kept in profiles/r04_gpu_parity.log); the assertions only say the engine is not slower.
Inputs are features (the oracle's Fbank is NumPy on the host and stays out of both sides)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
from oracle import models as om  # noqa: E402

pytestmark = pytest.mark.gpu
B, T, F, NCLS = 256, 298, 80, 2796


def _time(fn, warm=3, reps=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def test_eager_framework_forward_beside_the_engine():
    from ppvector.loss.aamloss import AAMLoss
    from ppvector.models.ecapa_tdnn import EcapaTdnn
    from ppvector.models.fc import SpeakerIdentification
    p = om.ecapa_params(F, seed=1000)
    Wh = om.head_params(192, NCLS, seed=1001)
    g = torch.Generator().manual_seed(3)
    feats = torch.randn(B, T, F, generator=g).cuda()
    labels = (torch.arange(B) * 7 % NCLS).cuda()
    pc = {k: v.cuda() for k, v in p.items()}
    Wc = Wh.cuda()

    def eager(dtype):
        with torch.no_grad(), torch.autocast('cuda', dtype=dtype, enabled=dtype is not None):
            emb = om.ecapa_forward(pc, feats)
            return om.aam_loss(om.cosine_head(emb.float(), Wc), labels, 0.2, 32.0)

    ms_f32 = _time(lambda: eager(None))
    ms_bf16 = _time(lambda: eager(torch.bfloat16))
    m = EcapaTdnn(F)
    m.load_state_dict(p)
    m = m.cuda().eval()
    head = SpeakerIdentification(192, NCLS)
    head.load_state_dict({'weight': Wh})
    head = head.cuda().eval()
    crit = AAMLoss(margin=0.2, scale=32)
    res = {}
    for dt in ('float32', 'bfloat16'):
        eng = m.engine(dt)
        res[dt] = _time(lambda: crit(head(eng.forward(feats)), labels))
    loss_e = float(eager(None))
    loss_v = float(crit(head(m.engine('float32').forward(feats)), labels))
    print(f'[yardstick forward B={B}] eager PyTorch-ROCm over the oracle graph: f32 {ms_f32:.2f} ms, bf16 autocast {ms_bf16:.2f} ms;  '
          f'this engine: f32 {res["float32"]:.2f} ms, bf16 {res["bfloat16"]:.2f} ms (one launch sequence, no graph, features in);  '
          f'loss {loss_v:.5f} vs {loss_e:.5f}')
    assert abs(loss_v - loss_e) < 1e-3 * abs(loss_e)
    assert res['bfloat16'] < ms_bf16 and res['float32'] < ms_f32


def test_eager_framework_training_step_beside_the_engine():
    import ppvector
    from ppvector.loss.aamloss import AAMLoss
    from ppvector.models.ecapa_tdnn import EcapaTdnn
    from ppvector.models.fc import SpeakerIdentification
    from ppvector.optimizer.adam import Adam
    from ppvector.train.step import GraphedTrainStep
    p = om.ecapa_params(F, seed=1000)
    Wh = om.head_params(192, NCLS, seed=1001)
    g = torch.Generator().manual_seed(4)
    feats = torch.randn(B, T, F, generator=g).cuda()
    labels = (torch.arange(B) * 7 % NCLS).cuda()
    pr = {k: v.clone().cuda().requires_grad_(not k.endswith(('_mean', '_variance'))) for k, v in p.items()}
    Wr = Wh.clone().cuda().requires_grad_()
    params = [v for v in pr.values() if v.requires_grad] + [Wr]
    opt = torch.optim.Adam(params, lr=1e-4, weight_decay=1e-6)

    def eager_step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast('cuda', dtype=torch.bfloat16):
            emb = om.ecapa_forward(pr, feats, training=True)
        loss = om.aam_loss(om.cosine_head(emb.float(), Wr), labels, 0.2, 32.0)
        loss.backward()
        opt.step()
        return loss

    ms_eager = _time(eager_step, warm=2, reps=4)
    del opt, pr, Wr, params
    torch.cuda.empty_cache()
    old = ppvector.get_train_amp()
    ppvector.set_train_amp(True)
    try:
        m = EcapaTdnn(F)
        m.load_state_dict(p)
        head = SpeakerIdentification(192, NCLS)
        head.load_state_dict({'weight': Wh})
        model = torch.nn.Sequential(m, head).cuda().train()
        crit = AAMLoss(margin=0.2, scale=32)
        opt2 = Adam(model.parameters(), learning_rate=1e-4, weight_decay=1e-6)
        step = GraphedTrainStep(model, crit, opt2)
        ms_engine = _time(lambda: step(feats, labels), warm=6, reps=10)
        loss = float(step(feats, labels)[0])
    finally:
        ppvector.set_train_amp(old)
    assert np.isfinite(loss)
    print(f'[yardstick training step B={B}] eager PyTorch-ROCm (bf16 autocast, autograd over the oracle graph, torch Adam): {ms_eager:.1f} ms;  '
          f'this engine (enable_amp, staged HIP graphs, features in): {ms_engine:.2f} ms')
    assert ms_engine < ms_eager

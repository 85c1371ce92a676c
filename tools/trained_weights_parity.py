"""north_star's parity bar -- cosine scores within 1e-4 of the f32 reference -- at TRAINED weights.

The parity tests hold the engines to the oracle at random-init weights with randomised BatchNorm statistics.  A trained network is a
different operating point (sparser ReLU outputs, larger dynamic range, a peaked attention softmax), so this trains the backbone for a few
hundred steps on the synthetic speakers of tools/amp_convergence.py (this package's own training step, enable_amp), then scores 96
held-out utterances all-pairs with
    the CPU oracle (f32; oracle/*.py, the restatement pinned on the reference's model files) -- the yardstick,
    the f32 engine, the split-precision ('float32x3') engine and the bf16 engine of this package, eval mode, the same features,
and prints the largest cosine-score difference of each engine from the oracle.
    python tools/trained_weights_parity.py [EcapaTdnn|TDNN|CAMPPlus|ResNetSE|ERes2Net] [steps] [batch]"""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'), os.path.join(ROOT, 'tools')):
    if p not in sys.path:
        sys.path.insert(0, p)
import warnings  # noqa: E402

import numpy as np  # noqa: E402
import torch  # noqa: E402

import amp_convergence as ac  # noqa: E402
import ppvector  # noqa: E402
from oracle import campplus as oc  # noqa: E402
from oracle import eres2net as oer  # noqa: E402
from oracle import models as om  # noqa: E402
from oracle import resnet_se as orse  # noqa: E402
from ppvector.data_utils.featurizer import AudioFeaturizer  # noqa: E402
from ppvector.loss.aamloss import AAMLoss  # noqa: E402
from ppvector.models.campplus import CAMPPlus  # noqa: E402
from ppvector.models.ecapa_tdnn import EcapaTdnn  # noqa: E402
from ppvector.models.eres2net import ERes2Net  # noqa: E402
from ppvector.models.fc import SpeakerIdentification  # noqa: E402
from ppvector.models.resnet_se import ResNetSE  # noqa: E402
from ppvector.models.tdnn import TDNN  # noqa: E402
from ppvector.optimizer.adam import Adam  # noqa: E402
from ppvector.train.step import TrainStep  # noqa: E402

n_spk, epochs = 64, 10
FWD = {'CAMPPlus': oc.campplus_forward, 'TDNN': om.tdnn_forward, 'EcapaTdnn': om.ecapa_forward, 'ResNetSE': orse.resnetse_forward,
       'ERes2Net': oer.eres2net_forward}
MAKE = {'CAMPPlus': lambda: CAMPPlus(80, embd_dim=192), 'TDNN': lambda: TDNN(80), 'EcapaTdnn': lambda: EcapaTdnn(80),
        'ResNetSE': lambda: ResNetSE(80, embd_dim=192), 'ERes2Net': lambda: ERes2Net(80, embd_dim=192, m_channels=32)}


def scores(e):
    e = e.double()
    e = e / e.norm(dim=1, keepdim=True)
    return e @ e.t()


def run(name, steps=240, B=64, n_eval=96, verbose=True):
    """Train `name` for `steps` steps of B under enable_amp, then score n_eval held-out utterances all-pairs with the CPU oracle and both
    engines -> dict(loss, acc, err_{f32,x3,bf16}, rel_{f32,x3,bf16}, eer_oracle, eer_{f32,x3,bf16})."""
    from ppvector.metric.metrics import evaluate_trials
    spe = max(1, steps // epochs)
    table = ac.speaker_table(n_spk, 1000)
    fz = AudioFeaturizer('Fbank', dict(sr=16000, n_mels=80))

    def batch(seed, n):
        rng = np.random.RandomState(seed)
        lab = rng.randint(0, n_spk, n)
        wav = np.stack([ac.synth_utterance(table, int(s), 48000, np.random.RandomState(seed * 1000 + k)) for k, s in enumerate(lab)]).astype(np.float32)
        with torch.no_grad():
            return fz(torch.from_numpy(wav).cuda()).contiguous(), torch.from_numpy(lab).cuda()

    pool = [batch(200 + i, B) for i in range(16)]
    torch.manual_seed(7)
    m = MAKE[name]()
    model = torch.nn.Sequential(m, SpeakerIdentification(192, n_spk)).cuda()
    was = ppvector.get_train_amp()
    ppvector.set_train_amp(True)
    cur = {'i': 0}

    def lr_at(i):
        if i < spe:
            return 1e-3 * (i + 1) / spe
        return 1e-5 + 0.5 * (1e-3 - 1e-5) * (1.0 + math.cos(math.pi * (i - spe) / max(1, steps - spe)))

    crit = AAMLoss(margin=0.0, scale=32)
    opt = Adam(model.parameters(), learning_rate=lambda: lr_at(cur['i']), weight_decay=1e-6)
    step = TrainStep(model, crit, opt, overlap_allreduce=False)
    try:
        for i in range(steps):
            cur['i'] = i
            crit.update(margin=om.margin_schedule(i, spe, epochs, 0.0, 0.3))
            loss, acc = step(*pool[i % len(pool)])
        torch.cuda.synchronize()
    finally:
        ppvector.set_train_amp(was)
    out = dict(loss=float(loss), acc=float(acc))
    if verbose:
        print(f'# {name}: trained {steps} steps of {B} under enable_amp: last loss {out["loss"]:.4f}, accuracy {out["acc"]:.3f}')
    feats, lab = batch(999, n_eval)
    m.eval()
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    if os.environ.get('VP_TWP_SAVE'):                     # the trained operating point, for offline studies on the CPU oracle
        torch.save(dict(name=name, state=sd, feats=feats.cpu(), labels=lab.cpu()), os.environ['VP_TWP_SAVE'])
    with torch.no_grad():
        e_or = FWD[name](sd, feats.cpu())
    s_or = scores(e_or)
    labc = lab.cpu()
    same = labc[:, None] == labc[None, :]
    off = ~torch.eye(n_eval, dtype=torch.bool)
    h = n_eval // 2

    def eer_of(e):
        e = e.float()
        return float(evaluate_trials(e[:h].cuda(), labc[:h].numpy(), e[h:].cuda(), labc[h:].numpy())[0])

    out['eer_oracle'] = eer_of(e_or)
    if verbose:
        print(f'  oracle scores: same-speaker pairs mean {s_or[same & off].mean():.3f}, different-speaker pairs mean {s_or[~same].mean():.3f}; '
              f'EER of the second half scored against the first {out["eer_oracle"]:.4f}')
    for dt, tag in (('float32', 'f32'), ('float32x3', 'x3'), ('bfloat16', 'bf16')):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            eng = m.engine(dt)
        with torch.no_grad():
            e = eng.forward(feats.to(torch.bfloat16) if dt == 'bfloat16' else feats).float().cpu()
        out['err_' + tag] = (scores(e) - s_or).abs().max().item()
        out['rel_' + tag] = ((e.double() - e_or.double()).norm() / e_or.double().norm()).item()
        out['eer_' + tag] = eer_of(e)
        if verbose:
            print(f'  {dt:9s} engine vs CPU oracle at the trained weights: max |cosine score difference| over {n_eval} x {n_eval} pairs {out["err_" + tag]:.2e} '
                  f'(north_star: 1e-4), embeddings rel-L2 {out["rel_" + tag]:.2e}, EER {out["eer_" + tag]:.4f}')
    return out


if __name__ == '__main__':
    run(sys.argv[1] if len(sys.argv) > 1 else 'EcapaTdnn', int(sys.argv[2]) if len(sys.argv) > 2 else 240, int(sys.argv[3]) if len(sys.argv) > 3 else 64)

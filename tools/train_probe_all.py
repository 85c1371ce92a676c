"""Training-step time of every built backbone at the per-GPU batch of its BASELINE config (forward + loss + backward + Adam, synthetic
3 s features), eager and replayed from captured HIP graphs (GraphedTrainStep), f32 engine and enable_amp.
Usage: python tools/train_probe_all.py [name ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import torch  # noqa: E402
import ppvector  # noqa: E402
from ppvector.loss.aamloss import AAMLoss  # noqa: E402
from ppvector.models.campplus import CAMPPlus  # noqa: E402
from ppvector.models.ecapa_tdnn import EcapaTdnn  # noqa: E402
from ppvector.models.eres2net import ERes2Net  # noqa: E402
from ppvector.models.fc import SpeakerIdentification  # noqa: E402
from ppvector.models.resnet_se import ResNetSE  # noqa: E402
from ppvector.models.tdnn import TDNN  # noqa: E402
from ppvector.optimizer.adam import Adam  # noqa: E402
from ppvector.train.step import GraphedTrainStep, TrainStep  # noqa: E402

CASES = [  # name, constructor, feature dim, per-GPU batch, classes
    ('TDNN', lambda: TDNN(80), 80, 64, 2796),
    ('EcapaTdnn', lambda: EcapaTdnn(80), 80, 32, 2796),                      # configs[1] strong-scaled over 8 GPUs
    ('CAMPPlus', lambda: CAMPPlus(80, embd_dim=192), 80, 64, 7205),          # configs[2]: 512 over 8 GPUs
    ('ResNetSE', lambda: ResNetSE(64), 64, 32, 2796),                        # configs[3]: 128 over 4 GPUs, MelSpectrogram(64)
    ('ERes2Net', lambda: ERes2Net(80), 80, 32, 2796),
    # configs[4] AT ITS NAMED SHAPE: ERes2Net-large (55.2 M parameters) + 200 000-class head, 1024 over 8 GPUs = 128 per GPU
    ('ERes2Net-large-200k', lambda: ERes2Net(80, embd_dim=192, m_channels=64, mul_channel=2, expansion=4, base_width=24, scale=3), 80, 128, 200000),
]
want = sys.argv[1:]
for name, make, F, B, ncls in CASES:
    if (want and name not in want) or (not want and name.endswith('200k')):      # the 55 M / 200 k case only when asked for (minutes)
        continue
    for amp in (False, True):
        ppvector.set_train_amp(amp)
        try:
            torch.manual_seed(0)
            m = make()
            emb = getattr(m, 'embd_dim', None) or getattr(m, 'emb_size', None) or 192
            model = torch.nn.Sequential(m, SpeakerIdentification(emb, ncls)).cuda()
            x = torch.randn(B, 298, F, device='cuda') * 3
            y = torch.randint(0, ncls, (B,), device='cuda')
            res = {}
            for kind, cls in (('eager', TrainStep), ('graphs', GraphedTrainStep)):
                crit = AAMLoss()
                opt = Adam(model.parameters(), learning_rate=1e-5, weight_decay=1e-6)
                step = cls(model, crit, opt)
                for _ in range(5):
                    step(x, y)
                torch.cuda.synchronize()
                t0 = time.time()
                n = 10
                for _ in range(n):
                    loss, acc = step(x, y)
                torch.cuda.synchronize()
                res[kind] = (time.time() - t0) / n * 1e3
                if kind == 'graphs' and getattr(step, 'capture_error', None):
                    res[kind] = float('nan')
            nparam = sum(q.numel() for q in model.parameters())
            print(f'{name:10s} B={B:3d} {"enable_amp" if amp else "f32       "}: eager {res["eager"]:8.2f} ms | staged graphs {res["graphs"]:8.2f} ms '
                  f'({B / min(res.values()) * 1e3:7.0f} utt/s); {nparam / 1e6:.1f} M parameters, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB',
                  flush=True)
            del step, opt, model, m
            torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
        except Exception as e:  # noqa: BLE001
            print(f'{name:10s} B={B} amp={amp}: {type(e).__name__}: {str(e)[:160]}', flush=True)
        finally:
            ppvector.set_train_amp(False)

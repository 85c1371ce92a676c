"""Training-step time of the built backbones on the f32 engine (forward + backward + Adam), synthetic 3 s features."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import torch  # noqa: E402
from oracle import models as om  # noqa: E402
from ppvector.loss.aamloss import AAMLoss  # noqa: E402
from ppvector.models.ecapa_tdnn import EcapaTdnn  # noqa: E402
from ppvector.models.fc import SpeakerIdentification  # noqa: E402
from ppvector.models.tdnn import TDNN  # noqa: E402
from ppvector.optimizer.adam import Adam  # noqa: E402
from ppvector.train.step import TrainStep  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for name, cls, params, gf in (('TDNN', TDNN, om.tdnn_params(80), 1.47), ('EcapaTdnn', EcapaTdnn, om.ecapa_params(80), 2.857)):
    if len(sys.argv) > 2 and name not in sys.argv[2:]:
        continue
    m = cls(80)
    m.load_state_dict(params)
    model = torch.nn.Sequential(m, SpeakerIdentification(192, 2796)).cuda()
    crit = AAMLoss()
    opt = Adam(model.parameters(), learning_rate=1e-4, weight_decay=1e-6)
    step = TrainStep(model, crit, opt)
    x = torch.randn(B, 298, 80, device='cuda') * 3
    y = torch.randint(0, 2796, (B,), device='cuda')
    for _ in range(2):
        step(x, y)
    torch.cuda.synchronize()
    t0 = time.time()
    n = 5
    for _ in range(n):
        loss, acc = step(x, y)
    torch.cuda.synchronize()
    ms = (time.time() - t0) / n * 1e3
    print(f'{name:10s} train step f32 B={B}: {ms:8.2f} ms  {B / ms * 1e3:8.0f} utt/s  {3 * gf * B / ms:7.1f} TFLOP/s (3x forward flops)  loss {loss.item():.3f}', flush=True)

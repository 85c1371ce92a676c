"""Experiment: capture forward + backward + Adam of one training step in a HIP graph (torch.cuda.CUDAGraph) and replay it.
Scalars (lr, margin, Adam step count) are frozen into the captured launches -- this measures launch overhead only."""
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
from ppvector.optimizer.adam import Adam  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
m = EcapaTdnn(80)
m.load_state_dict(om.ecapa_params(80))
model = torch.nn.Sequential(m, SpeakerIdentification(192, 2796)).cuda().train()
crit = AAMLoss()
opt = Adam(model.parameters(), learning_rate=1e-4, weight_decay=1e-6)
x = torch.randn(B, 298, 80, device='cuda') * 3
y = torch.randint(0, 2796, (B,), device='cuda')


def one_step():
    loss = crit(model(x), y)
    loss.backward()
    opt.step()
    opt.clear_grad()
    return loss


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        one_step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(5):
    one_step()
torch.cuda.synchronize()
eager = (time.time() - t0) / 5 * 1e3
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    static_loss = one_step()
torch.cuda.synchronize()
for _ in range(2):
    g.replay()
torch.cuda.synchronize()
t0 = time.time()
for _ in range(10):
    g.replay()
torch.cuda.synchronize()
graph = (time.time() - t0) / 10 * 1e3
print(f'ECAPA train step f32 B={B}: eager {eager:.2f} ms ({B / eager * 1e3:.0f} utt/s)   graph replay {graph:.2f} ms ({B / graph * 1e3:.0f} utt/s)   loss {static_loss.item():.4f}')

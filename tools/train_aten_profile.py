"""Which PyTorch-side (aten) kernels a training step launches, and from where: torch.profiler over three steps, grouped by
operator and by the two innermost Python frames.  (The libvpmi launches do not appear here: they go through ctypes.)
python tools/train_aten_profile.py [batch] [EcapaTdnn|CAMPPlus]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402
import ppvector  # noqa: E402
from oracle import models as om  # noqa: E402
from ppvector.loss.aamloss import AAMLoss  # noqa: E402
from ppvector.models.ecapa_tdnn import EcapaTdnn  # noqa: E402
from ppvector.models.fc import SpeakerIdentification  # noqa: E402
from ppvector.optimizer.adam import Adam  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ppvector.set_train_amp(True)
if len(sys.argv) > 2 and sys.argv[2] == 'CAMPPlus':
    from ppvector.models.campplus import CAMPPlus
    torch.manual_seed(0)
    m = CAMPPlus(80, embd_dim=192)
else:
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


for _ in range(3):
    one_step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    for _ in range(3):
        one_step()
torch.cuda.synchronize()
rows = {}
for e in prof.events():
    n = e.name
    if not n.startswith('aten::') or n in ('aten::empty', 'aten::empty_like', 'aten::empty_strided', 'aten::view', 'aten::as_strided', 'aten::slice',
                                          'aten::select', 'aten::reshape', 'aten::permute', 'aten::t', 'aten::transpose', 'aten::detach', 'aten::alias',
                                          'aten::narrow', 'aten::split', 'aten::chunk', 'aten::_unsafe_view', 'aten::expand', 'aten::unsqueeze', 'aten::squeeze',
                                          'aten::contiguous', 'aten::to', 'aten::_to_copy', 'aten::result_type', 'aten::item', 'aten::_local_scalar_dense'):
        continue
    st = [s for s in (e.stack or []) if 'ppvector' in s or 'train_aten' in s or 'autograd' in s]
    key = (n, ' <- '.join(s.split('/')[-1] for s in st[:2]))
    rows[key] = rows.get(key, 0) + 1
print(f'aten operators that launch kernels, per step (B = {B}), by call site:')
for (n, site), c in sorted(rows.items(), key=lambda kv: -kv[1])[:45]:
    print(f'{c / 3:7.1f}  {n:28s} {site}')
print('total', sum(rows.values()) / 3)

"""Eight eager enable_amp training steps of CAM++ (cam), ResNetSE (res) or ERes2Net (eres) at the per-GPU batch of their BASELINE config,
for rocprofv3 --kernel-trace --stats (that is how the ResNetSE SE-gate backward was found).
Usage: rocprofv3 --kernel-trace --stats -- python tools/train_prof_other.py cam|res|eres"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'voiceprintrecognition-paddlepaddle_amd'))
import torch, ppvector
from ppvector.loss.aamloss import AAMLoss
from ppvector.models.campplus import CAMPPlus
from ppvector.models.resnet_se import ResNetSE
from ppvector.models.eres2net import ERes2Net
from ppvector.models.fc import SpeakerIdentification
from ppvector.optimizer.adam import Adam
from ppvector.train.step import TrainStep
name = sys.argv[1]
ppvector.set_train_amp(True)
torch.manual_seed(0)
if name == 'cam':
    m, F, B, ncls = CAMPPlus(80, embd_dim=192), 80, 64, 7205
elif name == 'eres':
    m, F, B, ncls = ERes2Net(80), 80, 32, 2796
else:
    m, F, B, ncls = ResNetSE(64), 64, 32, 2796
model = torch.nn.Sequential(m, SpeakerIdentification(192 if name == 'cam' else getattr(m, 'embd_dim', 192), ncls)).cuda()
x = torch.randn(B, 298, F, device='cuda') * 3
y = torch.randint(0, ncls, (B,), device='cuda')
step = TrainStep(model, AAMLoss(), Adam(model.parameters(), learning_rate=1e-5, weight_decay=1e-6))
for _ in range(8):
    step(x, y)
torch.cuda.synchronize()

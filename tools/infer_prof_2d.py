"""A few eval-mode forwards of CAM++ (cam), ResNetSE (res) or ERes2Net (eres) at B = 64 x 3 s on one engine, for rocprofv3 --kernel-trace --stats.
Usage: rocprofv3 --kernel-trace --stats -- python tools/infer_prof_2d.py cam|res|eres [bfloat16|float32x3|float32] [B]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import warnings
import torch
from ppvector.models.campplus import CAMPPlus
from ppvector.models.eres2net import ERes2Net
from ppvector.models.resnet_se import ResNetSE
name = sys.argv[1]
dt = sys.argv[2] if len(sys.argv) > 2 else 'bfloat16'
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
torch.manual_seed(0)
m = {'cam': lambda: CAMPPlus(80, embd_dim=192), 'res': lambda: ResNetSE(80), 'eres': lambda: ERes2Net(80)}[name]().cuda().eval()
x = torch.randn(B, 298, 80, device='cuda') * 3
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    eng = m.engine(dt)
xin = x.to(torch.bfloat16) if dt == 'bfloat16' else x
with torch.no_grad():
    for _ in range(6):
        eng.forward(xin)
torch.cuda.synchronize()

"""Forward-time probe of the built backbones at a given batch (3 s utterances, T = 298; bf16, split-precision 'float32x3' and f32), with
each engine's fraction of ITS matrix-core peak (MI355X_MICROARCH.md: 2.5 PFLOP/s dense bf16 -- a third of it for x3 --, 157 TFLOP/s f32 MFMA).  The f32 engine is the one that meets
north_star's 1e-4 on ResNetSE / ERes2Net (bf16: 2.1e-4 / 1.8e-4, tests/test_gpu_models.py), so ITS throughput is the parity-clean
number for BASELINE configs 4 and 5.   python tools/model_probe.py [B] [model ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import torch  # noqa: E402
from oracle import campplus as oc  # noqa: E402
from oracle import eres2net as oer  # noqa: E402
from oracle import models as om  # noqa: E402
from oracle import resnet_se as orse  # noqa: E402
from ppvector.models.campplus import CAMPPlus  # noqa: E402
from ppvector.models.ecapa_tdnn import EcapaTdnn  # noqa: E402
from ppvector.models.eres2net import ERes2Net  # noqa: E402
from ppvector.models.resnet_se import ResNetSE  # noqa: E402
from ppvector.models.tdnn import TDNN  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
x = torch.randn(B, 298, 80, device='cuda') * 3
PEAK = {'bfloat16': 2500.0, 'float32': 157.3, 'float32x3': 2500.0 / 3}       # x3: three bf16 MFMAs per product
GF = {'EcapaTdnn': 2.857, 'TDNN': 1.47, 'CAMPPlus': 3.20, 'ResNetSE': 11.07, 'ERes2Net': 10.2}
for name, cls, params in (('EcapaTdnn', EcapaTdnn, om.ecapa_params(80)), ('TDNN', TDNN, om.tdnn_params(80)),
                          ('CAMPPlus', lambda f: CAMPPlus(f, embd_dim=192), oc.campplus_params(80, 192)),
                          ('ResNetSE', ResNetSE, orse.resnetse_params(80, 192)),
                          ('ERes2Net', ERes2Net, oer.eres2net_params(80, 192))):
    if len(sys.argv) > 2 and name not in sys.argv[2:]:
        continue
    m = cls(80)
    m.load_state_dict(params)
    m = m.cuda().eval()
    dts = os.environ.get('VP_DTYPES')
    for dt in (tuple(dts.split(',')) if dts else ('bfloat16',) if os.environ.get('VP_BF16_ONLY') == '1' else ('bfloat16', 'float32x3', 'float32')):
        eng = m.engine(dt)
        xin = x.to(torch.bfloat16) if dt == 'bfloat16' else x
        for _ in range(2):
            eng.forward(xin)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
        for a, b in evs:
            a.record(); eng.forward(xin); b.record()
        torch.cuda.synchronize()
        ms = sorted(a.elapsed_time(b) for a, b in evs)[2]
        import ppvector
        gms = None
        if os.environ.get('VP_GRAPH') == '1':
            ppvector.set_compute_dtype(dt); ppvector.set_graph_mode(True)
            for _ in range(2):
                m(xin)
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
            for a, b in evs:
                a.record(); m(xin); b.record()
            torch.cuda.synchronize()
            gms = sorted(a.elapsed_time(b) for a, b in evs)[2]
            ppvector.set_graph_mode(False); ppvector.set_compute_dtype('float32')
        if gms is not None:
            print(f'{name:10s} {dt:9s} B={B}: graph replay {gms:8.3f} ms  {B / gms * 1e3:10.0f} utt/s', flush=True)
        tf = B * GF[name] / ms
        print(f'{name:10s} {dt:9s} B={B}: {ms:8.3f} ms  {B / ms * 1e3:10.0f} utt/s  {tf:8.1f} TFLOP/s (algorithmic) = {tf / PEAK[dt]:.3f} of the '
              f'{dt} matrix-core peak ({PEAK[dt]:.0f} TF)', flush=True)

"""Per-conv phase stamps of res2_chain_kernel inside an ECAPA forward (needs the -DVP_TIMING variant via VPMI_LIB)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import torch  # noqa: E402
from oracle import models as om  # noqa: E402
from ppvector import _native as N  # noqa: E402
from ppvector.models.ecapa_tdnn import EcapaTdnn  # noqa: E402

B = 256
m = EcapaTdnn(80)
m.load_state_dict(om.ecapa_params(80))
m = m.cuda().eval()
x = (torch.randn(B, 298, 80, device='cuda') * 3).to(torch.bfloat16)
eng = m.engine('bfloat16')
for _ in range(2):
    eng.forward(x)
dbg = torch.zeros((B, 12), dtype=torch.int64, device='cuda')
raw = C.CDLL(os.environ['VPMI_LIB'])
raw.vp_dbg_res2_buffer.argtypes = [C.c_void_p]
raw.vp_dbg_res2_buffer(dbg.data_ptr())
eng.forward(x)
torch.cuda.synchronize()
s = dbg.cpu().double() / 100.0
d = s[:, 1:9] - s[:, 0:8]
print('res2 chain (last block), us per phase, mean over workgroups: stage-in', f'{d[:, 0].mean():.2f}', ' convs', ' '.join(f'{d[:, i].mean():.2f}' for i in range(1, 8)),
      ' total', f'{(s[:, 8] - s[:, 0]).mean():.2f}', ' span', f'{s[:, 8].max() - s[:, 0].min():.2f}')

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
dbg = torch.zeros((B, 32), dtype=torch.int64, device='cuda')
raw = C.CDLL(os.environ['VPMI_LIB'])
raw.vp_dbg_res2_buffer.argtypes = [C.c_void_p]
raw.vp_dbg_res2_buffer(dbg.data_ptr())
eng.forward(x)
torch.cuda.synchronize()
s = dbg.cpu().double() / 100.0
# stamps: 0 start, 1 staged; per conv: MFMAs done, DMAs landed + barrier, epilogue + barrier, copy-out + barrier
print('res2 chain (last block), us, mean over workgroups: stage-in', f'{(s[:, 1] - s[:, 0]).mean():.2f}')
for j in range(7):
    q = s[:, 1 + 4 * j: 6 + 4 * j]
    d = (q[:, 1:] - q[:, :-1]).mean(0)
    print(f'  conv {j}: mfma {d[0]:.2f}  dma-wait+barrier {d[1]:.2f}  epilogue+barrier {d[2]:.2f}  copy-out+barrier {d[3]:.2f}   total {d.sum():.2f}')
print('  total', f'{(s[:, 29] - s[:, 0]).mean():.2f}', ' span', f'{s[:, 29].max() - s[:, 0].min():.2f}')

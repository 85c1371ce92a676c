"""Per-phase time of cam_block_kernel (last dense block) inside a CAM++ forward (needs the -DVP_TIMING variant via VPMI_LIB)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import torch  # noqa: E402
from oracle import campplus as oc  # noqa: E402
from ppvector.models.campplus import CAMPPlus  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = CAMPPlus(80, embd_dim=192)
m.load_state_dict(oc.campplus_params(80, 192))
m = m.cuda().eval()
x = (torch.randn(B, 298, 80, device='cuda') * 3).to(torch.bfloat16)
eng = m.engine('bfloat16')
for _ in range(2):
    eng.forward(x)
dbg = torch.zeros((B, 5), dtype=torch.int64, device='cuda')
raw = C.CDLL(os.environ['VPMI_LIB'])
raw.vp_dbg_cam_buffer.argtypes = [C.c_void_p]
raw.vp_dbg_cam_buffer(dbg.data_ptr())
eng.forward(x)
torch.cuda.synchronize()
s = dbg.cpu().double() / 100.0
names = ['stage tables', 'phase 1 GEMM', 'epilogue h + sums', 'gate', 'local conv + store']
print('cam_block (last block, 16 layers), us per workgroup, mean:', '  '.join(f'{n} {s[:, i].mean():.1f}' for i, n in enumerate(names)), ' total', f'{s.sum(1).mean():.1f}')

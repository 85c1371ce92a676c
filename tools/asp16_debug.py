"""enable_amp ECAPA training forward + backward with the MFA output as bf16 only (default) against VPMI_MFA_F32_OUT=1: per-parameter
gradient differences.  python tools/asp16_debug.py [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'voiceprintrecognition-paddlepaddle_amd'))
import torch
import ppvector
from ppvector.models.ecapa_tdnn import EcapaTdnn
from ppvector.train.ecapa_train import ecapa_forward_train
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
if os.environ.get('TRACE_CALLS'):
    from ppvector import _native as N
    real = N.lib()
    class Proxy:
        def __getattr__(self, name):
            f = getattr(real, name)
            if not callable(f):
                return f
            def call(*a):
                if Proxy.on:
                    print('  call', name, flush=True)
                return f(*a)
            return call
    Proxy.on = False
    N.lib = lambda: Proxy()
ppvector.set_train_amp(True)
torch.manual_seed(3)
m = EcapaTdnn(80).cuda().train()
state = {k: v.clone() for k, v in m.state_dict().items()}
x = torch.randn(B, 298, 80, device='cuda')
g = torch.randn(B, 192, device='cuda')
res = {}
for mode in ('f32out', 'bf16out'):
    if mode == 'f32out':
        os.environ['VPMI_MFA_F32_OUT'] = '1'
    else:
        os.environ.pop('VPMI_MFA_F32_OUT', None)
    m.load_state_dict(state)
    for p in m.parameters():
        p.grad = None
    emb = ecapa_forward_train(m, x)
    torch.cuda.synchronize(); print(mode, 'forward ok', flush=True)
    if os.environ.get('TRACE_CALLS') and mode == 'bf16out':
        Proxy.on = True
    emb.backward(g)
    torch.cuda.synchronize(); print(mode, 'backward ok', flush=True)
    res[mode] = (emb.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters()})
e0, g0 = res['f32out']; e1, g1 = res['bf16out']
print('emb rel', float((e1 - e0).norm() / e0.norm()))
for k in g0:
    r = float((g1[k] - g0[k]).norm() / (g0[k].norm() + 1e-30))
    if r > 2e-2:
        print(f'{k:50s} rel {r:.3e}  |g| {float(g0[k].norm()):.3e}')

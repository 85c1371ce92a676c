import sys, os
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT,'voiceprintrecognition-paddlepaddle_amd'))
import torch
from oracle import eres2net as oer
from oracle import models as om
from ppvector.models.eres2net import ERes2Net
from ppvector.train.functions import HeadLoss
def rel(a,b): a=a.double().cpu(); b=b.double().cpu(); return ((a-b).norm()/b.norm()).item()
LARGE = dict(m_channels=64, mul_channel=2, expansion=4, base_width=24, scale=3)
FW = dict(m_channels=64, expansion=4, base_width=24, scale=3)
Cc=1000
for (B,T,Fd) in ((4,24,16),(8,48,32)):
    nb=(3,4,6,3)
    p = oer.eres2net_params(Fd, 192, seed=23, num_blocks=nb, **LARGE)
    g = torch.Generator().manual_seed(24)
    x = torch.randn(B, T, Fd, generator=g) * 2
    labels = torch.randint(0, Cc, (B,), generator=g)
    Wh = om.head_params(192, Cc, seed=5)
    grads={}
    for dt in (torch.float64, torch.float32):
        pr = {k: v.clone().to(dt).requires_grad_(not k.endswith(('_mean', '_variance'))) for k, v in p.items()}
        Wr = Wh.clone().to(dt).requires_grad_()
        e = oer.eres2net_forward(pr, x.to(dt), num_blocks=nb, training=True, **FW)
        l = om.aam_loss(om.cosine_head(e, Wr), labels, 0.2, 32.0, False, 0.0)
        l.backward()
        grads[dt]={k:v.grad for k,v in pr.items() if v.grad is not None}
    m = ERes2Net(Fd, embd_dim=192, num_blocks=list(nb), **LARGE); m.load_state_dict(p); m=m.cuda().train()
    Wd = Wh.cuda().requires_grad_()
    emb = m(x.cuda())
    loss = HeadLoss.apply(emb, Wd, labels.cuda(), 0.2, 32.0, 0.0, False)[0]
    loss.backward()
    rows=[]
    for k,v in m.named_parameters():
        gr=grads[torch.float64].get(k)
        if gr is None or gr.norm().item()<1e-9: continue
        rows.append((rel(v.grad,gr), rel(grads[torch.float32][k],gr), k))
    rows.sort(reverse=True)
    print(f'B={B} T={T} F={Fd}: worst engine-vs-f64 gradient errors (engine, oracle-in-f32, name):')
    for r in rows[:6]: print('   %.3e  %.3e  %s'%r)
    import statistics
    print('   median engine %.3e  median f32-oracle %.3e'%(statistics.median(r[0] for r in rows), statistics.median(r[1] for r in rows)), flush=True)

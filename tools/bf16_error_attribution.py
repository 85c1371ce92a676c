"""Where does the bf16 engine's score error at trained weights come from?  Offline, on the CPU oracle:
    VP_TWP_SAVE=gpurun_out/ecapa_trained.pt python tools/trained_weights_parity.py EcapaTdnn 240 64      # on the GPU box: saves weights + features
    python tools/bf16_error_attribution.py gpurun_out/ecapa_trained.pt                                   # anywhere: no GPU needed
An instrumented copy of oracle/models.py: ecapa_forward rounds to bf16 (or fp16) at selected sites only -- the weights of every conv, the
activations between the layers as the bf16 engine stores them, the MFA output, the operands of ASP's two convs, the x of ASP's weighted
statistics -- and reports the largest all-pairs cosine-score change against the unrounded forward.  Round 5 (48 held-out utterances, ECAPA
after 240 steps; the engine itself measures 1.9e-3 on 96): everything 2.0e-3 | weights only 1.6e-3 | block activations only 1.1e-3 | MFA
output only 3.2e-4 | all of ASP's operands 3.3e-4 | everything as fp16 2.9e-4.  The error is the bf16 rounding of the WEIGHTS and of the
activations of every layer, not one sensitive stage: no local higher-precision patch brings the bf16 engine under north_star's 1e-4."""
import sys, torch, torch.nn.functional as F
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import models as om
d = torch.load(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/ecapa_trained.pt')
p, feats = d['state'], d['feats'][:48]
torch.set_num_threads(16)
def scores(e):
    e=e.double(); e=e/e.norm(dim=1,keepdim=True); return e@e.t()
def rb(x): return x.float().to(torch.bfloat16).to(x.dtype)
def rh(x): return x.float().to(torch.float16).to(x.dtype)
# instrumented forward: rounding function per site
def fwd(p, x, R):
    c=om.ECAPA_DEFAULTS
    W=lambda k: R('w', p[k])
    def conv_same(x,wk,bk,d):
        w=W(wk); k=w.shape[-1]; pad=d*(k-1)//2
        if pad>0: x=F.pad(x,(pad,pad),mode='reflect')
        return F.conv1d(x,w,p[bk],dilation=d)
    def tdnn(x,pre,d,site):
        y=conv_same(R(site+'.in',x),pre+'conv.conv.weight',pre+'conv.conv.bias',d)
        return om.batchnorm(F.relu(y),p,pre+'norm.norm.')
    x=R('feat',x.transpose(1,2))
    x=R('act',tdnn(x,'blocks.0.',1,'b0'))
    xl=[]
    for i in range(1,4):
        pre=f'blocks.{i}.'
        res=x
        h=R('act',tdnn(x,pre+'tdnn1.',1,'t1'))
        # res2
        chunks=torch.chunk(h,8,dim=1); ys=[chunks[0]]
        y=None
        for j in range(1,8):
            inp=chunks[j] if j==1 else chunks[j]+y
            y=R('act',tdnn(inp,f'{pre}res2net_block.blocks.{j-1}.',c['dilations'][i],'r2'))
            ys.append(y)
        h=torch.cat(ys,1)
        h=R('act',tdnn(h,pre+'tdnn2.',1,'t2'))
        s=h.mean(2,keepdim=True)
        s=F.relu(F.conv1d(s,p[pre+'se_block.conv1.conv.weight'],p[pre+'se_block.conv1.conv.bias']))
        s=torch.sigmoid(F.conv1d(s,p[pre+'se_block.conv2.conv.weight'],p[pre+'se_block.conv2.conv.bias']))
        x=R('act',s*h+res)
        xl.append(x)
    x=torch.cat(xl,1)
    x=R('mfa_out',tdnn(x,'mfa.',1,'mfa'))
    B,C,L=x.shape
    mean=x.mean(2); std=torch.sqrt(((x-mean.unsqueeze(2))**2).mean(2).clamp(min=1e-12))
    attn=torch.cat([x,mean.unsqueeze(2).expand(B,C,L),std.unsqueeze(2).expand(B,C,L)],1)
    a=F.conv1d(R('asp_x',attn),R('asp_w1',p['asp.tdnn.conv.conv.weight']),p['asp.tdnn.conv.conv.bias'])
    a=om.batchnorm(F.relu(a),p,'asp.tdnn.norm.norm.')
    a=F.conv1d(R('asp_h',torch.tanh(a)),R('asp_w2',p['asp.conv.conv.weight']),p['asp.conv.conv.bias'])
    a=F.softmax(a,dim=2)
    xs=R('asp_xs',x)
    mean=(a*xs).sum(2); std=torch.sqrt((a*(xs-mean.unsqueeze(2))**2).sum(2).clamp(min=1e-12))
    x=torch.cat([mean,std],1)
    x=om.batchnorm(x,p,'asp_bn.norm.')
    return F.conv1d(x.unsqueeze(2),p['fc.conv.weight'],p['fc.conv.bias']).squeeze(-1)
with torch.no_grad():
    ref=scores(fwd(p,feats,lambda s,x:x))
    print('check vs oracle', (scores(om.ecapa_forward(p,feats))-ref).abs().max().item())
    def run(name, sites, rf=rb):
        R=lambda s,x: rf(x) if (s in sites or (s.endswith('.in') and 'in' in sites)) else x
        e=(scores(fwd(p,feats,R))-ref).abs().max().item()
        print(f'{name:60s} max score err {e:.2e}')
    ALL={'w','feat','act','mfa_out','asp_x','asp_w1','asp_h','asp_w2','asp_xs','in'}
    run('everything bf16 (operands + stored activations)', ALL)
    run('only weights', {'w','asp_w1','asp_w2'})
    run('only block activations + features', {'feat','act','in'})
    run('only MFA output (as stored, used by ASP)', {'mfa_out'})
    run('only ASP tdnn input x', {'asp_x'})
    run('only ASP h (tanh output)', {'asp_h'})
    run('only ASP w2', {'asp_w2'})
    run('only ASP w1', {'asp_w1'})
    run('only ASP statistics x', {'asp_xs'})
    run('ASP all (x,h,w1,w2,xs)', {'asp_x','asp_h','asp_w1','asp_w2','asp_xs'})
    run('everything except ASP + mfa_out', ALL-{'asp_x','asp_h','asp_w1','asp_w2','asp_xs','mfa_out'})
    run('everything fp16', ALL, rh)
    run('everything bf16 but ASP h,w2 fp16-free (f32)', ALL-{'asp_h','asp_w2'})
    run('everything bf16 but whole ASP + mfa_out f32', ALL-{'asp_x','asp_h','asp_w1','asp_w2','asp_xs','mfa_out'})

"""Oracle (CPU, PyTorch fp32/fp64): ECAPA-TDNN / TDNN forward, cosine head, AAMLoss.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Functional restatement over a flat state
dict whose keys are the reference's Paddle parameter names, so one set of weights feeds the
oracle, the shim-executed reference files and the HIP engine.

Reference lines restated (all under /root/reference/ppvector/):
  models/utils.py:65-93     Conv1d: "same" conv = reflect pad d*(k-1)/2 each side, then valid conv
  models/utils.py:96-119    BatchNorm1d: eps 1e-5; Paddle momentum 0.9 (running = .9*run + .1*batch)
  models/utils.py:122-148   TDNNBlock: BN(ReLU(Conv1d(x)))  -- ReLU BEFORE BN
  models/ecapa_tdnn.py:11-47   Res2NetBlock (y0=x0, y1=f1(x1), yi=fi(xi+y_{i-1}))
  models/ecapa_tdnn.py:50-82   SEBlock (lengths=None branch: plain mean over time)
  models/ecapa_tdnn.py:85-142  SERes2NetBlock
  models/ecapa_tdnn.py:245-276 EcapaTdnn.forward
  models/pooling.py:69-125     AttentiveStatisticsPooling (global context, all-ones mask)
  models/tdnn.py:46-68         TDNN.forward
  models/fc.py:41-53           SpeakerIdentification.forward (Cosine / Linear)
  loss/aamloss.py:28-53        AAMLoss.forward / update
Paddle op semantics assumed [3P-memory]: paddle.var/std unbiased by default; F.normalize
eps 1e-12 (x / max(||x||, eps)); F.linear(x, W) = x @ W with W [in, out]; nn.Linear.weight is
[in, out]; BatchNorm in train mode normalises with the biased batch variance;
CrossEntropyLoss(label_smoothing) = mean over the batch of -sum(q * log_softmax).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5

# Mixed-precision emulation for the tests of the HIP training engine under enable_amp: when AMP is True every convolution
# GEMM sees its operands rounded to bf16 -- x and w in the forward and in the weight / data gradient, the incoming gradient
# dz in both backward GEMMs -- while biases, BatchNorm, activations and accumulation keep the tensor dtype (float64 in the
# tests): the arithmetic libvpmi performs with vp_conv1d_desc.mfma_bf16, minus its f32 accumulation order.
AMP = False


class _RoundBf16(torch.autograd.Function):
    """bf16 rounding forward, straight-through backward (the GEMM reads a rounded copy; the tensor itself is untouched)."""
    @staticmethod
    def forward(ctx, x):
        return x.float().to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundGradBf16(torch.autograd.Function):
    """identity forward, bf16-rounded gradient backward (both backward GEMMs read dz rounded)."""
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.float().to(torch.bfloat16).to(g.dtype)


def _conv1d(x, w, b=None, **kw):
    if not AMP:
        return F.conv1d(x, w, b, **kw)
    y = _RoundGradBf16.apply(F.conv1d(_RoundBf16.apply(x), _RoundBf16.apply(w), None, **kw))
    return y if b is None else y + b.view(1, -1, 1)


# ----------------------------------------------------------------------------- layers
def conv1d_same(x, w, b, dilation=1):
    """models/utils.py:65-93 with stride 1: reflect pad, then valid conv. x (B,C,T)."""
    k = w.shape[-1]
    pad = dilation * (k - 1) // 2
    if pad > 0:
        x = F.pad(x, (pad, pad), mode='reflect')
    return _conv1d(x, w, b, dilation=dilation)


def batchnorm(x, p, prefix, training=False, stats_out=None):
    """Channel dim 1; x (B,C) or (B,C,T).  Keys: prefix + weight|bias|_mean|_variance."""
    w, b = p[prefix + 'weight'], p[prefix + 'bias']
    if training:
        dims = [0] + list(range(2, x.dim()))
        mean = x.mean(dim=dims)
        var = x.var(dim=dims, unbiased=False)
        if stats_out is not None:
            stats_out[prefix] = (mean.detach().clone(), var.detach().clone())
    else:
        mean, var = p[prefix + '_mean'], p[prefix + '_variance']
    shape = [1, -1] + [1] * (x.dim() - 2)
    return (x - mean.view(shape)) / torch.sqrt(var.view(shape) + BN_EPS) * w.view(shape) + b.view(shape)


def tdnn_block(x, p, prefix, dilation=1, training=False, stats_out=None):
    y = conv1d_same(x, p[prefix + 'conv.conv.weight'], p[prefix + 'conv.conv.bias'], dilation)
    return batchnorm(F.relu(y), p, prefix + 'norm.norm.', training, stats_out)


def res2net_block(x, p, prefix, scale=8, dilation=1, training=False, stats_out=None):
    ys = []
    y_i = None
    for i, x_i in enumerate(torch.chunk(x, scale, dim=1)):
        if i == 0:
            y_i = x_i
        elif i == 1:
            y_i = tdnn_block(x_i, p, f'{prefix}blocks.{i - 1}.', dilation, training, stats_out)
        else:
            y_i = tdnn_block(x_i + y_i, p, f'{prefix}blocks.{i - 1}.', dilation, training, stats_out)
        ys.append(y_i)
    return torch.cat(ys, dim=1)


def se_block(x, p, prefix):
    s = x.mean(dim=2, keepdim=True)
    s = F.relu(_conv1d(s, p[prefix + 'conv1.conv.weight'], p[prefix + 'conv1.conv.bias']))
    s = torch.sigmoid(_conv1d(s, p[prefix + 'conv2.conv.weight'], p[prefix + 'conv2.conv.bias']))
    return s * x


def seres2net_block(x, p, prefix, scale=8, dilation=1, training=False, stats_out=None):
    residual = x
    if (prefix + 'shortcut.conv.weight') in p:
        residual = _conv1d(x, p[prefix + 'shortcut.conv.weight'], p[prefix + 'shortcut.conv.bias'])
    x = tdnn_block(x, p, prefix + 'tdnn1.', 1, training, stats_out)
    x = res2net_block(x, p, prefix + 'res2net_block.', scale, dilation, training, stats_out)
    x = tdnn_block(x, p, prefix + 'tdnn2.', 1, training, stats_out)
    x = se_block(x, p, prefix + 'se_block.')
    return x + residual


def asp(x, p, prefix, global_context=True, training=False, stats_out=None, eps=1e-12):
    """pooling.py:86-125 with lengths=None (mask of ones, total = L)."""
    B, C, L = x.shape

    def stats(x, m):
        mean = (m * x).sum(2)
        std = torch.sqrt((m * (x - mean.unsqueeze(2)).pow(2)).sum(2).clamp(min=eps))
        return mean, std

    if global_context:
        m = torch.full((B, 1, L), 1.0 / L, dtype=x.dtype, device=x.device)
        mean, std = stats(x, m)
        attn = torch.cat([x, mean.unsqueeze(2).expand(B, C, L), std.unsqueeze(2).expand(B, C, L)], dim=1)
    else:
        attn = x
    attn = tdnn_block(attn, p, prefix + 'tdnn.', 1, training, stats_out)
    attn = _conv1d(torch.tanh(attn), p[prefix + 'conv.conv.weight'], p[prefix + 'conv.conv.bias'])
    attn = F.softmax(attn, dim=2)
    mean, std = stats(x, attn)
    return torch.cat([mean, std], dim=1)


# ----------------------------------------------------------------------------- models
ECAPA_DEFAULTS = dict(embd_dim=192, channels=[512, 512, 512, 512, 1536], kernel_sizes=[5, 3, 3, 3, 1],
                      dilations=[1, 2, 3, 4, 1], attention_channels=128, res2net_scale=8,
                      se_channels=128, global_context=True)


def ecapa_forward(p, x, prefix='', training=False, stats_out=None, taps=None, **cfg):
    """EcapaTdnn.forward (ecapa_tdnn.py:245-276), pooling_type ASP.  x: (B, T, F) -> (B, embd)."""
    c = dict(ECAPA_DEFAULTS)
    c.update(cfg)
    x = x.transpose(1, 2)
    xl = []
    x = tdnn_block(x, p, prefix + 'blocks.0.', c['dilations'][0], training, stats_out)
    xl.append(x)
    for i in range(1, len(c['channels']) - 1):
        x = seres2net_block(x, p, f'{prefix}blocks.{i}.', c['res2net_scale'], c['dilations'][i],
                            training, stats_out)
        xl.append(x)
    x = torch.cat(xl[1:], dim=1)
    x = tdnn_block(x, p, prefix + 'mfa.', c['dilations'][-1], training, stats_out)
    if taps is not None:
        taps['blocks'] = xl
        taps['mfa'] = x
    x = asp(x, p, prefix + 'asp.', c['global_context'], training, stats_out)
    if taps is not None:
        taps['asp'] = x
    x = batchnorm(x, p, prefix + 'asp_bn.norm.', training, stats_out)
    x = _conv1d(x.unsqueeze(2), p[prefix + 'fc.conv.weight'], p[prefix + 'fc.conv.bias']).squeeze(-1)
    return x


def tdnn_forward(p, x, prefix='', training=False, stats_out=None):
    """TDNN.forward (tdnn.py:46-68), pooling_type ASP: five un-padded Conv1D + ReLU + BN."""
    x = x.transpose(1, 2)
    for i, d in zip(range(1, 5), (1, 2, 3, 1)):
        x = F.relu(_conv1d(x, p[f'{prefix}td_layer{i}.weight'], p[f'{prefix}td_layer{i}.bias'], dilation=d))
        x = batchnorm(x, p, f'{prefix}bn{i}.', training, stats_out)
    x = F.relu(_conv1d(x, p[prefix + 'td_layer5.weight'], p[prefix + 'td_layer5.bias']))
    x = asp(x, p, prefix + 'pooling.', True, training, stats_out)
    x = batchnorm(x, p, prefix + 'bn5.norm.', training, stats_out)
    x = x @ p[prefix + 'linear.weight'] + p[prefix + 'linear.bias']        # Paddle Linear: [in, out]
    return batchnorm(x, p, prefix + 'bn6.norm.', training, stats_out)


def cosine_head(emb, weight):
    """fc.py:49: logits = normalize(x, axis=1) @ normalize(W, axis=0); W is [D, C]."""
    xn = emb / emb.norm(dim=1, keepdim=True).clamp(min=1e-12)
    wn = weight / weight.norm(dim=0, keepdim=True).clamp(min=1e-12)
    return xn @ wn


def classifier_head(emb, p, classifier_type='Cosine', num_blocks=0, training=False, stats_out=None, prefix=''):
    """SpeakerIdentification.forward (fc.py:41-53) with its DenseLayer('batchnorm') stages (fc.py:27-29, :56-71):
    x -> [Conv1D(k=1) -> BatchNorm1D] * num_blocks -> cosine / linear logits.  p: the head's state dict."""
    x = emb
    for i in range(num_blocks):
        w = p[f'{prefix}blocks.{i}.linear.weight']
        x = x @ w.reshape(w.shape[0], -1).t() + p[f'{prefix}blocks.{i}.linear.bias']
        x = batchnorm(x, p, f'{prefix}blocks.{i}.nonlinear.batchnorm.', training, stats_out)
    if classifier_type == 'Cosine':
        return cosine_head(x, p[prefix + 'weight'])
    return x @ p[prefix + 'output.weight'] + p[prefix + 'output.bias']


def classifier_params(input_dim=192, num_speakers=20, classifier_type='Cosine', K=1, num_blocks=0, inter_dim=512, seed=1002,
                      dtype=torch.float32):
    rng = np.random.RandomState(seed)
    p, d = {}, input_dim
    for i in range(num_blocks):
        p.update(_conv_keys(f'blocks.{i}.linear.', inter_dim, d, 1, rng))
        p.update(_bn_keys(f'blocks.{i}.nonlinear.batchnorm.', inter_dim, rng, True))
        d = inter_dim
    if classifier_type == 'Cosine':
        p['weight'] = rng.standard_normal((d, num_speakers * K)) / np.sqrt(d)
    else:
        p['output.weight'] = rng.standard_normal((d, num_speakers)) / np.sqrt(d)
        p['output.bias'] = 0.1 * rng.standard_normal(num_speakers)
    return {k: torch.as_tensor(np.asarray(v), dtype=dtype) for k, v in p.items()}


def aam_margins(margin):
    """aamloss.py:22-25 / :49-53."""
    return dict(cos_m=math.cos(margin), sin_m=math.sin(margin), th=math.cos(math.pi - margin),
                mmm=1.0 + math.cos(math.pi - margin))


def aam_loss(logits, labels, margin=0.2, scale=32.0, easy_margin=False, label_smoothing=0.0):
    """aamloss.py:28-47.  logits (B, C) cosines; labels (B,) int64.  Returns scalar mean loss."""
    m = aam_margins(margin)
    sine = torch.sqrt(1.0 - logits.pow(2))
    phi = logits * m['cos_m'] - sine * m['sin_m']
    if easy_margin:
        phi = torch.where(logits > 0, phi, logits)
    else:
        phi = torch.where(logits > m['th'], phi, logits - m['mmm'])
    one_hot = F.one_hot(labels, logits.shape[1]).to(logits.dtype)
    out = (one_hot * phi + (1.0 - one_hot) * logits) * scale
    logp = F.log_softmax(out, dim=1)
    C = logits.shape[1]
    q = one_hot * (1.0 - label_smoothing) + label_smoothing / C
    return -(q * logp).sum(1).mean()


def margin_schedule(step, step_per_epoch, max_epoch, initial_margin=0.0, final_margin=0.3,
                    increase_type='exp'):
    """optimizer/scheduler.py:79-99 with trainer.py:182-190 defaults."""
    start = int(max_epoch * 0.3) * step_per_epoch
    fix = int(max_epoch * 0.7) * step_per_epoch
    if step < start:
        return initial_margin
    if step >= fix:
        return final_margin
    cur = step - start
    if increase_type == 'exp':
        ratio = 1.0 - math.exp((cur / (fix - start)) * math.log(1e-3 / (1.0 + 1e-6))) * 1.0
    else:
        ratio = 1.0 * cur / (fix - start)
    return initial_margin + (final_margin - initial_margin) * ratio


# ----------------------------------------------------------------------------- parameters
def _bn_keys(prefix, c, rng, randomize_stats):
    d = {prefix + 'weight': 1.0 + 0.1 * rng.standard_normal(c), prefix + 'bias': 0.1 * rng.standard_normal(c)}
    if randomize_stats:
        d[prefix + '_mean'] = 0.1 * rng.standard_normal(c)
        d[prefix + '_variance'] = rng.uniform(0.5, 1.5, c)
    else:
        d[prefix + '_mean'] = np.zeros(c)
        d[prefix + '_variance'] = np.ones(c)
    return d


def _conv_keys(prefix, cout, cin, k, rng):
    bound = 1.0 / math.sqrt(cin * k)
    return {prefix + 'weight': rng.uniform(-bound, bound, (cout, cin, k)) * math.sqrt(3.0),
            prefix + 'bias': rng.uniform(-bound, bound, cout)}


def ecapa_params(input_size=80, seed=1000, randomize_stats=True, dtype=torch.float32, **cfg):
    """Random ECAPA-TDNN parameters keyed with the reference's Paddle names (SURVEY 8(d):
    BN running stats randomised so eval-mode BN is not an identity)."""
    c = dict(ECAPA_DEFAULTS)
    c.update(cfg)
    ch, ks = c['channels'], c['kernel_sizes']
    rng = np.random.RandomState(seed)
    p = {}

    def tdnn(prefix, cin, cout, k):
        p.update(_conv_keys(prefix + 'conv.conv.', cout, cin, k, rng))
        p.update(_bn_keys(prefix + 'norm.norm.', cout, rng, randomize_stats))

    tdnn('blocks.0.', input_size, ch[0], ks[0])
    sc = c['res2net_scale']
    for i in range(1, len(ch) - 1):
        pre = f'blocks.{i}.'
        tdnn(pre + 'tdnn1.', ch[i - 1], ch[i], 1)
        for j in range(sc - 1):
            tdnn(f'{pre}res2net_block.blocks.{j}.', ch[i] // sc, ch[i] // sc, 3)
        tdnn(pre + 'tdnn2.', ch[i], ch[i], 1)
        p.update(_conv_keys(pre + 'se_block.conv1.conv.', c['se_channels'], ch[i], 1, rng))
        p.update(_conv_keys(pre + 'se_block.conv2.conv.', ch[i], c['se_channels'], 1, rng))
        if ch[i - 1] != ch[i]:
            p.update(_conv_keys(pre + 'shortcut.conv.', ch[i], ch[i - 1], 1, rng))
    tdnn('mfa.', ch[-1], ch[-1], ks[-1])
    C = ch[-1]
    tdnn('asp.tdnn.', C * 3 if c['global_context'] else C, c['attention_channels'], 1)
    p.update(_conv_keys('asp.conv.conv.', C, c['attention_channels'], 1, rng))
    p.update(_bn_keys('asp_bn.norm.', 2 * C, rng, randomize_stats))
    p.update(_conv_keys('fc.conv.', c['embd_dim'], 2 * C, 1, rng))
    return {k: torch.tensor(np.asarray(v), dtype=dtype) for k, v in p.items()}


def tdnn_params(input_size=80, channels=512, embd_dim=192, seed=1000, randomize_stats=True,
                dtype=torch.float32):
    rng = np.random.RandomState(seed)
    p = {}
    shapes = [(input_size, 5), (channels, 3), (channels, 3), (channels, 1), (channels, 1)]
    for i, (cin, k) in enumerate(shapes, start=1):
        p.update(_conv_keys(f'td_layer{i}.', channels, cin, k, rng))
        if i < 5:
            p.update(_bn_keys(f'bn{i}.', channels, rng, randomize_stats))
    p.update(_conv_keys('pooling.tdnn.conv.conv.', 128, channels * 3, 1, rng))
    p.update(_bn_keys('pooling.tdnn.norm.norm.', 128, rng, randomize_stats))
    p.update(_conv_keys('pooling.conv.conv.', channels, 128, 1, rng))
    p.update(_bn_keys('bn5.norm.', channels * 2, rng, randomize_stats))
    bound = 1.0 / math.sqrt(channels * 2)
    p['linear.weight'] = rng.uniform(-bound, bound, (channels * 2, embd_dim)) * math.sqrt(3.0)
    p['linear.bias'] = rng.uniform(-bound, bound, embd_dim)
    p.update(_bn_keys('bn6.norm.', embd_dim, rng, randomize_stats))
    return {k: torch.tensor(np.asarray(v), dtype=dtype) for k, v in p.items()}


def head_params(embd_dim=192, num_speakers=2796, seed=1001, dtype=torch.float32):
    """fc.py:31-34: Xavier-uniform [D, C]."""
    rng = np.random.RandomState(seed)
    bound = math.sqrt(6.0 / (embd_dim + num_speakers))
    return torch.tensor(rng.uniform(-bound, bound, (embd_dim, num_speakers)), dtype=dtype)


def count_params(p):
    train = sum(v.numel() for k, v in p.items() if not k.endswith(('_mean', '_variance')))
    buf = sum(v.numel() for k, v in p.items() if k.endswith(('_mean', '_variance')))
    return train, buf

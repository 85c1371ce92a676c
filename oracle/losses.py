"""Oracle (CPU, PyTorch; TEST INFRASTRUCTURE ONLY -- never imported by the product path): the classification losses next to
AAMLoss in the reference's loss package, restated on plain torch so autograd gives the reference gradients.

  loss/amloss.py:14-25          AMLoss.forward        (additive cosine margin on the target, CE sum / B)
  loss/armloss.py:14-31         ARMLoss.forward       (AM logits, then every logit below the target's is zeroed)
  loss/celoss.py:11-19          CELoss.forward        (plain CE on the head's logits)
  loss/subcenterloss.py:32-54   SubCenterLoss.forward (max over the K sub-centres of a class, then the AAM margin)
  loss/sphereface2.py:47-69     SphereFace2.forward   (binary losses per class, margin types 'A' and 'C', learnable bias)

Pinned against those files executed through oracle/paddle_shim (oracle/gen_golden.py, tests/golden/losses_ref.npz).
TripletAngularMarginLoss (loss/tripletangularmarginloss.py) is not restated: its masked_select/reshape mining needs every
speaker to occur equally often in the batch, which the reference's random sampler does not provide.
"""
import torch
import torch.nn.functional as F

from .models import aam_margins


def _smoothed_ce_mean(out, labels, label_smoothing):
    """paddle.nn.CrossEntropyLoss(label_smoothing) mean == CrossEntropyLoss(reduction='sum') / B."""
    C = out.shape[1]
    one_hot = F.one_hot(labels, C).to(out.dtype)
    q = one_hot * (1.0 - label_smoothing) + label_smoothing / C
    return -(q * F.log_softmax(out, dim=1)).sum(1).mean()


def am_loss(logits, labels, margin=0.2, scale=30.0, label_smoothing=0.0):
    """amloss.py:14-25."""
    one_hot = F.one_hot(labels, logits.shape[1]).to(logits.dtype)
    return _smoothed_ce_mean(scale * (logits - margin * one_hot), labels, label_smoothing)


def arm_loss(logits, labels, margin=0.2, scale=30.0, label_smoothing=0.0):
    """armloss.py:14-31: predictions = where(z - z[label] < 0, 0, z), z = scale * (cos - margin * onehot)."""
    one_hot = F.one_hot(labels, logits.shape[1]).to(logits.dtype)
    z = scale * (logits - margin * one_hot)
    zy = z.gather(1, labels.reshape(-1, 1))
    pred = torch.where(z - zy < 0.0, torch.zeros_like(z), z)
    return _smoothed_ce_mean(pred, labels, label_smoothing)


def ce_loss(logits, labels, label_smoothing=0.0):
    """celoss.py:11-19."""
    return _smoothed_ce_mean(logits, labels, label_smoothing)


def subcenter_loss(logits, labels, margin=0.2, scale=32.0, easy_margin=False, K=3, label_smoothing=0.0):
    """subcenterloss.py:32-54.  logits (B, C*K): column c*K + k is sub-centre k of class c (reshape (-1, C, K), max over K)."""
    m = aam_margins(margin)
    cosine = logits.reshape(-1, logits.shape[1] // K, K).max(dim=2)[0]
    sine = torch.sqrt(1.0 - cosine.pow(2))
    phi = cosine * m['cos_m'] - sine * m['sin_m']
    phi = torch.where(cosine > 0, phi, cosine) if easy_margin else torch.where(cosine > m['th'], phi, cosine - m['mmm'])
    one_hot = F.one_hot(labels, cosine.shape[1]).to(logits.dtype)
    return _smoothed_ce_mean((one_hot * phi + (1.0 - one_hot) * cosine) * scale, labels, label_smoothing)


def sphereface2_loss(logits, labels, bias, margin=0.2, scale=32.0, lanbuda=0.7, t=3, margin_type='C'):
    """sphereface2.py:47-69.  bias: 0-d / (1, 1) tensor (the module's learnable parameter, zero-initialised)."""
    m = aam_margins(margin)
    b = bias.reshape(())

    def g(z):
        return 2 * torch.pow((z + 1) / 2, t) - 1

    if margin_type == 'A':
        sin = torch.sqrt(1.0 - logits.pow(2))
        p = scale * g(torch.where(logits > m['th'], logits * m['cos_m'] - sin * m['sin_m'], logits - m['mmm'])) + b
        n = scale * g(logits * m['cos_m'] + sin * m['sin_m']) + b
    else:
        p = scale * (g(logits) - margin) + b
        n = scale * (g(logits) + margin) + b
    pos = lanbuda * torch.log(1 + torch.exp(-1.0 * p))
    neg = (1 - lanbuda) * torch.log(1 + torch.exp(n))
    one_hot = F.one_hot(labels, logits.shape[1]).to(logits.dtype)
    return (one_hot * pos + (1 - one_hot) * neg).sum(1).mean()

"""Oracle (TEST INFRASTRUCTURE ONLY): Nystrom landmark self-attention and the blocks around it.

Plain PyTorch fp32 restatement of
  models/NystromAttention.py:20-35   moore_penrose_iter_pinv   (dup cmta_utils.py:147-162)
  models/NystromAttention.py:39-157  NystromAttention          (dup cmta_utils.py:166-281)
  models/mil.py:171-206              TransLayer, PPEG          (dup cmta_utils.py:858-891)
  models/mil.py:209-259              TransMIL

The pip package ``nystrom_attention`` imported at mil.py:24 is absent from /root/reference and
unpinned; the in-tree copy states the same algorithm and is what this
restatement follows and is pinned against.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


def pinv_newton_schulz(x: torch.Tensor, iters: int = 6, per_bag: bool = False) -> torch.Tensor:
    """Iterative Moore-Penrose pseudo-inverse of x [..., m, m]  (NystromAttention.py:20-35).

    The initial scaling uses the max row/column abs-sum over the WHOLE tensor (:26), which couples
    all bags and heads in a batch - kept as is."""
    ax = x.abs()
    if per_bag:                   # corrected semantics (NOT the reference): every bag scaled by its own maxima
        z = x.transpose(-1, -2) / (ax.sum(dim=-1).amax(dim=(1, 2)) * ax.sum(dim=-2).amax(dim=(1, 2))).view(-1, 1, 1, 1)
    else:
        z = x.transpose(-1, -2) / (ax.sum(dim=-1).max() * ax.sum(dim=-2).max())
    eye = torch.eye(x.shape[-1], dtype=x.dtype, device=x.device)
    for _ in range(iters):
        xz = x @ z
        z = 0.25 * z @ (13 * eye - xz @ (15 * eye - xz @ (7 * eye - xz)))
    return z


def nystrom_attention(
    x: torch.Tensor,
    p: Params,
    *,
    heads: int = 8,
    dim_head: int = 64,
    num_landmarks: int = 256,
    pinv_iterations: int = 6,
    residual: bool = True,
    residual_conv_kernel: int = 33,
    return_aux: bool = False,
    per_bag_pinv_scale: bool = False,
    mask=None,
    eps: float = 1e-8,
):
    """x [B, n, dim] -> [B, n, dim]   (NystromAttention.py:74-157, dropout off; `mask` [B, n] bool as at :84,92-96,106-118,127-133).

    Parameters: 'to_qkv.weight' [3*inner, dim], 'to_out.0.weight' [dim, inner], 'to_out.0.bias',
    'res_conv.weight' [heads, 1, k, 1]."""
    B, n, dim = x.shape
    m = num_landmarks
    inner = heads * dim_head
    scale = dim_head ** -0.5

    pad = (m - n % m) % m
    if pad:
        x = F.pad(x, (0, 0, pad, 0))                       # zero rows in FRONT (:82)
        if mask is not None:
            mask = F.pad(mask, (pad, 0), value=False)
    npad = n + pad

    qkv = x @ p["to_qkv.weight"].t()
    q, k, v = (t.reshape(B, npad, heads, dim_head).permute(0, 2, 1, 3) for t in qkv.chunk(3, dim=-1))
    if mask is not None:                                   # :92-96
        mk = mask.to(torch.bool).reshape(B, 1, npad)
        q, k, v = (t * mk[..., None].to(t.dtype) for t in (q, k, v))
    q = q * scale

    l = math.ceil(n / m)                                   # segment length (:102)
    div = l
    if mask is not None:                                   # masked mean (:106-118)
        msum = mk.reshape(B, 1, npad // l, l).sum(dim=-1)
        div = msum[..., None].to(q.dtype) + eps
        mland = msum > 0
    ql = q.reshape(B, heads, npad // l, l, dim_head).sum(dim=3) / div
    kl = k.reshape(B, heads, npad // l, l, dim_head).sum(dim=3) / div

    s1, s2, s3 = q @ kl.transpose(-1, -2), ql @ kl.transpose(-1, -2), ql @ k.transpose(-1, -2)
    if mask is not None:                                   # :127-133
        neg = -torch.finfo(q.dtype).max
        s1 = s1.masked_fill(~(mk[..., None] & mland[..., None, :]), neg)
        s2 = s2.masked_fill(~(mland[..., None] & mland[..., None, :]), neg)
        s3 = s3.masked_fill(~(mland[..., None] & mk[..., None, :]), neg)
    a1 = torch.softmax(s1, dim=-1)                         # [B, h, n', m]
    a2 = torch.softmax(s2, dim=-1)                         # [B, h, m, m]
    a3 = torch.softmax(s3, dim=-1)                         # [B, h, m, n']
    a2i = pinv_newton_schulz(a2, pinv_iterations, per_bag_pinv_scale)

    out = (a1 @ a2i) @ (a3 @ v)                            # :140
    if residual:
        kk = residual_conv_kernel
        w = p["res_conv.weight"].reshape(heads, 1, kk, 1)
        out = out + F.conv2d(v, w, padding=(kk // 2, 0), groups=heads)   # :144-145
    out = out.permute(0, 2, 1, 3).reshape(B, npad, inner)
    out = out @ p["to_out.0.weight"].t() + p["to_out.0.bias"]
    out = out[:, -n:]                                      # drops the front padding (:149)
    if return_aux:
        return out, dict(a1=a1, a2=a2, a3=a3, a2i=a2i, q=q, k=k, v=v, ql=ql, kl=kl)
    return out


def _sub(p: Params, prefix: str) -> Params:
    return {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)}


def layer_norm(x, p: Params, prefix: str, eps: float = 1e-5):
    return F.layer_norm(x, x.shape[-1:], p[prefix + "weight"], p[prefix + "bias"], eps)


def trans_layer(x: torch.Tensor, p: Params, *, dim: int = 512) -> torch.Tensor:
    """x + NystromAttention(LayerNorm(x))   (mil.py:171-189): heads 8, dim_head dim//8,
    landmarks dim//2, 6 pinv iterations, residual conv 33."""
    y = layer_norm(x, p, "norm.")
    return x + nystrom_attention(y, _sub(p, "attn."), heads=8, dim_head=dim // 8, num_landmarks=dim // 2)


def ppeg(x: torch.Tensor, H: int, W: int, p: Params) -> torch.Tensor:
    """Depthwise 7x7 + 5x5 + 3x3 convolutions + identity on the token grid; cls token bypasses
    (mil.py:192-206).  x [B, 1 + H*W, C]."""
    B, _, C = x.shape
    cls, feat = x[:, :1], x[:, 1:]
    f = feat.transpose(1, 2).reshape(B, C, H, W)
    y = (F.conv2d(f, p["proj.weight"], p["proj.bias"], padding=3, groups=C) + f
         + F.conv2d(f, p["proj1.weight"], p["proj1.bias"], padding=2, groups=C)
         + F.conv2d(f, p["proj2.weight"], p["proj2.bias"], padding=1, groups=C))
    return torch.cat((cls, y.flatten(2).transpose(1, 2)), dim=1)


def trans_mil(x: torch.Tensor, p: Params, *, dim: int = 512):
    """TransMIL forward (mil.py:225-259): fc1+ReLU, wrap-pad to a square, cls token, Nystrom layer,
    PPEG, Nystrom layer, LayerNorm, cls read-out -> (encoded, logits)."""
    h = x if x.dtype == torch.float64 else x.float()       # fp64 only for the tests' noise calibration
    h = torch.relu(h @ p["_fc1.0.weight"].t() + p["_fc1.0.bias"])
    n = h.shape[1]
    side = int(math.ceil(math.sqrt(n)))
    h = torch.cat([h, h[:, : side * side - n]], dim=1)
    h = torch.cat((p["cls_token"].expand(h.shape[0], -1, -1), h), dim=1)
    h = trans_layer(h, _sub(p, "layer1."), dim=dim)
    h = ppeg(h, side, side, _sub(p, "pos_layer."))
    h = trans_layer(h, _sub(p, "layer2."), dim=dim)
    h = layer_norm(h, p, "norm.")[:, 0]
    logits = h @ p["_fc2.weight"].t() + p["_fc2.bias"]
    encoded = h @ p["multimodal_projection.weight"].t() + p["multimodal_projection.bias"]
    return encoded, logits

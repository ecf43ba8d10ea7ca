"""Oracle (TEST INFRASTRUCTURE ONLY): co-attention returning raw scores, and the gated bilinear fusion.

Plain PyTorch fp32 restatement of
  models/MultiheadAttention.py:116-321  multi_head_attention_forward (need_raw path, optional masks, dropout off)
  models/fusion.py:6-63                 BilinearFusion (eval-mode BatchNorm / Dropout, or batch statistics)
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


def coattention(query, key, value, p: Params, num_heads: int = 1, key_padding_mask=None, attn_mask=None):
    """query [L, B, E], key/value [S, B, E] -> (out [L, B, E], raw pre-softmax scores [B, h, L, S]).
    Masks as in MultiheadAttention.py:284-298: bool attn_mask [L, S] / [B h, L, S] and key_padding_mask [B, S] fill with -inf, a float attn_mask is
    added; the raw scores returned are the masked ones."""
    L, B, E = query.shape
    S = key.shape[0]
    hd = E // num_heads
    w, b = p["in_proj_weight"], p["in_proj_bias"]
    q = (query @ w[:E].t() + b[:E]) * hd ** -0.5
    k = key @ w[E:2 * E].t() + b[E:2 * E]
    v = value @ w[2 * E:].t() + b[2 * E:]
    qh = q.reshape(L, B * num_heads, hd).transpose(0, 1)
    kh = k.reshape(S, B * num_heads, hd).transpose(0, 1)
    vh = v.reshape(S, B * num_heads, hd).transpose(0, 1)
    raw = qh @ kh.transpose(1, 2)
    if attn_mask is not None:
        am = attn_mask.unsqueeze(0) if attn_mask.dim() == 2 else attn_mask
        raw = raw.masked_fill(am, float("-inf")) if am.dtype == torch.bool else raw + am.to(raw.dtype)
    if key_padding_mask is not None:
        raw = raw.reshape(B, num_heads, L, S).masked_fill(key_padding_mask.view(B, 1, 1, S), float("-inf")).reshape(B * num_heads, L, S)
    o = torch.softmax(raw, dim=-1) @ vh
    o = o.transpose(0, 1).reshape(L, B, E)
    return o @ p["out_proj.weight"].t() + p["out_proj.bias"], raw.reshape(B, num_heads, L, S)


def bilinear_fusion(vec1, vec2, p: Params, *, skip: int = 0, train_stats: bool = False, eps: float = 1e-5):
    """fusion.py:36-63 with gate1 = gate2 = use_bilinear = 1; BatchNorm uses running statistics (eval) unless
    train_stats; dropout off."""
    lin = lambda x, pre: x @ p[pre + "weight"].t() + p[pre + "bias"]
    bil = lambda a, b, pre: torch.einsum("bi,oij,bj->bo", a, p[pre + "weight"], b) + p[pre + "bias"]

    def bn(x, pre):
        if train_stats:
            mu, var = x.mean(0), x.var(0, unbiased=False)
        else:
            mu, var = p[pre + "running_mean"], p[pre + "running_var"]
        return (x - mu) / torch.sqrt(var + eps) * p[pre + "weight"] + p[pre + "bias"]

    v1, v2 = torch.relu(vec1), torch.relu(vec2)
    o1 = torch.relu(lin(torch.sigmoid(bil(v1, v2, "linear_z1.")) * torch.relu(lin(v1, "linear_h1.0.")), "linear_o1.0."))
    o2 = torch.relu(lin(torch.sigmoid(bil(v1, v2, "linear_z2.")) * torch.relu(lin(v2, "linear_h2.0.")), "linear_o2.0."))
    one = torch.ones(o1.shape[0], 1, dtype=o1.dtype)
    o1, o2 = torch.cat((o1, one), 1), torch.cat((o2, one), 1)
    o12 = (o1.unsqueeze(2) * o2.unsqueeze(1)).flatten(1)
    out = torch.relu(bn(lin(o12, "encoder1.0."), "encoder1.1."))
    if skip:
        out = torch.cat((out, o1, o2), 1)
    return torch.relu(bn(lin(out, "encoder2.0."), "encoder2.1."))

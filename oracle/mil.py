"""Oracle (TEST INFRASTRUCTURE ONLY): the MIL assembly around the deformable cross-attention.

Plain PyTorch fp32 restatement of
  models/DeformCrossTransMIL.py:28-38    FusionNet
  models/DeformCrossTransMIL.py:40-77    DeformCrossTransLayer
  models/DeformCrossTransMIL.py:79-160   DeformCrossTransMIL
  models/DeformCrossTransMIL.py:169-202  Pooler
  models/model.py:142-187                MaxNet   (eval mode: AlphaDropout is the identity)
  models/model.py:440-544                DeformPathomicNet (fusion_type='concat')
parametrised in the token grid (reference: 50x50, DeformCrossTransMIL.py:104).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.nn.functional as F

from .deform import deform_cross_attention_1d, deform_cross_attention_2d
from .nystrom import _sub, layer_norm

Params = Dict[str, torch.Tensor]


def _fl(x):
    """the reference casts inputs with .float(); an fp64 run (noise calibration of the tests) keeps fp64"""
    return x if x.dtype == torch.float64 else x.float()


def linear(x, p: Params, prefix: str):
    return x @ p[prefix + "weight"].t() + p[prefix + "bias"]


def deform_cross_trans_layer(x1, x2, p: Params, attn_dim: int, grid_hw, q_chunk: int = 512):
    """x1 + attn(LN(x1)^T, LN(x2)^T)^T with one shared LayerNorm (DeformCrossTransMIL.py:62-77).
    x1/x2 token-major [B, n, C]; the 2-D module runs dim 128, heads 8, dim_head 64, groups 8,
    offset_scale 4 (:45-54), the 1-D module dim 128, offset_scale 2, groups 4 (:55-60)."""
    a = layer_norm(x1, p, "norm.").transpose(1, 2)
    b = layer_norm(x2, p, "norm.").transpose(1, 2)
    if attn_dim == 2:
        y, vgrid = deform_cross_attention_2d(a, b, _sub(p, "attn2d."), grid_hw=grid_hw, q_chunk=q_chunk)
    else:
        y, vgrid = deform_cross_attention_1d(a, b, _sub(p, "attn1d."), offset_scale=2.0, q_chunk=q_chunk)
    return x1 + y.transpose(1, 2), vgrid


def pooler(h, p: Params):
    """mean over tokens -> Linear -> tanh (DeformCrossTransMIL.py:193-200)."""
    return torch.tanh(linear(h.mean(dim=1), p, "dense."))


# Imposed decisions of the first layer's ReLU (parity tests at the headline size; same idea as oracle.deform.DECISIONS): relu(_fc1(bag)) is
# piecewise linear in 1.3e6 pre-activations per bag, and one of them within fp32 rounding of zero decides differently in two fp32 programs -
# which moves one ROW of d _fc1.weight by that token's whole contribution (5e-3 of the row's scale at 10 000 tokens).  A test may queue one
# bool mask [B, N, C] per deform_cross_trans_mil call (consumed in call order): relu(x) then becomes x * mask.
FC1_DECISIONS = None


def deform_cross_trans_mil(path, omic, p: Params, *, attn_dim: int = 2, grid_hw: Tuple[int, int] = (50, 50),
                           q_chunk: int = 512):
    """path [B, N, F_in], omic [B, C] -> (encoded [B, C], logits [B, n_classes], omic_tiled [B, N, C], vgrid)
    (DeformCrossTransMIL.py:97-160)."""
    pre = linear(_fl(path), p, "_fc1.0.")
    if FC1_DECISIONS:
        path = pre * FC1_DECISIONS.pop(0).to(device=pre.device, dtype=pre.dtype)
    else:
        path = torch.relu(pre)                                         # :100
    N = path.shape[1]
    omic_t = _fl(omic).unsqueeze(1).repeat(1, N, 1)                    # :104 (2500 in the reference)
    h = linear(torch.cat((path, omic_t), dim=-1), p, "fusion_layer.fusion_layer.")   # :35-37,110
    if attn_dim == 1:
        cls = p["cls_token"].expand(h.shape[0], -1, -1)
        h = torch.cat((cls, h), dim=1)
        path = torch.cat((cls, path), dim=1)
        h, vgrid = deform_cross_trans_layer(h, path, _sub(p, "layer3."), 1, None, q_chunk)
        h = layer_norm(h, p, "norm.")[:, 0]                               # :127
    else:
        h, vgrid = deform_cross_trans_layer(h, path, _sub(p, "layer3."), 2, grid_hw, q_chunk)
        h = pooler(layer_norm(h, p, "norm."), _sub(p, "pooler."))         # :144
    logits = linear(h, p, "_fc2.")
    encoded = linear(h, p, "multimodal_projection.")
    return encoded, logits, omic_t, vgrid


def max_net(x, p: Params):
    """4 x (Linear + ELU [+ AlphaDropout, identity in eval]) -> ReLU -> classifier (model.py:148-187)."""
    h = x
    for i in range(4):
        h = F.elu(linear(h, p, f"encoder.{i}.0."))
    feats = torch.relu(h)
    return feats, linear(feats, p, "classifier.0.")


def deform_pathomic_net(x_path, x_omic_tumor, x_omic_immune, p: Params, *, attn_dim: int = 2,
                        grid_hw=(50, 50), task_type: str = "diag2021", q_chunk: int = 512):
    """DeformPathomicNet.forward with fusion_type='concat', return_vgrid=True (model.py:481-544).
    Returns (features, vec_tumor, vec_immune, [haz_t, haz_i, haz, omic_t, vgrid_t, omic_i, vgrid_i])."""
    ot, _ = max_net(x_omic_tumor, _sub(p, "omic_net_tumor."))
    vt, _, omic_t, vg_t = deform_cross_trans_mil(x_path, ot, _sub(p, "pathomic_net_tumor."),
                                                 attn_dim=attn_dim, grid_hw=grid_hw, q_chunk=q_chunk)
    oi, _ = max_net(x_omic_immune, _sub(p, "omic_net_immune."))
    vi, _, omic_i, vg_i = deform_cross_trans_mil(x_path, oi, _sub(p, "pathomic_net_immune."),
                                                 attn_dim=attn_dim, grid_hw=grid_hw, q_chunk=q_chunk)
    feats = torch.cat((vt, vi), dim=1)
    haz = linear(feats, p, "classifier.")
    haz_t = linear(vt, p, "classifier_tumor.0.")
    haz_i = linear(vi, p, "classifier_immune.0.")
    if task_type == "survival":
        haz, haz_t, haz_i = torch.sigmoid(haz), torch.sigmoid(haz_t), torch.sigmoid(haz_i)
    return feats, vt, vi, [haz_t, haz_i, haz, omic_t, vg_t, omic_i, vg_i]

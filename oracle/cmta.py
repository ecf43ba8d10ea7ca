"""Oracle (TEST INFRASTRUCTURE ONLY): the CMTA assembly around the Nystrom blocks and the co-attention pair.

Plain PyTorch fp32 restatement of
  models/cmta_utils.py:894-924   Transformer_P
  models/cmta_utils.py:927-948   Transformer_G
  models/mcat_utils.py:81-95     SNN_Block        (eval mode: AlphaDropout is the identity)
  models/model.py:714-853        CMTA.forward     (fusion 'concat'; eval mode: Dropout(0.25) is the identity)
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

from .coattn import coattention
from .nystrom import _sub, layer_norm, ppeg, trans_layer

Params = Dict[str, torch.Tensor]


def transformer_p(features, p: Params, dim: int):
    """features [B, n, dim] -> (cls [B, dim], patches [B, side^2, dim])   (cmta_utils.py:906-924)."""
    n = features.shape[1]
    side = int(math.ceil(math.sqrt(n)))
    h = torch.cat([features, features[:, : side * side - n]], dim=1)
    h = torch.cat((p["cls_token"].expand(h.shape[0], -1, -1), h), dim=1)
    h = trans_layer(h, _sub(p, "layer1."), dim=dim)
    h = ppeg(h, side, side, _sub(p, "pos_layer."))
    h = trans_layer(h, _sub(p, "layer2."), dim=dim)
    h = layer_norm(h, p, "norm.")
    return h[:, 0], h[:, 1:]


def transformer_g(features, p: Params, dim: int):
    """features [B, n, dim] -> (cls, patches)   (cmta_utils.py:937-948)."""
    h = torch.cat((p["cls_token"].expand(features.shape[0], -1, -1), features), dim=1)
    h = trans_layer(h, _sub(p, "layer1."), dim=dim)
    h = trans_layer(h, _sub(p, "layer2."), dim=dim)
    h = layer_norm(h, p, "norm.")
    return h[:, 0], h[:, 1:]


def cmta(x_path, x_omic, p: Params, omic_sizes=(100, 100, 100, 131)):
    """CMTA.forward, fusion = 'concat', model sizes 'small' (model.py:778-853).
    Returns (logits, hazards, S, cls_p_enc, cls_p_dec, cls_g_enc, cls_g_dec)."""
    lin = lambda x, pre: x @ p[pre + "weight"].t() + p[pre + "bias"]
    h_path = torch.relu(lin(x_path, "wsi_net.0."))
    h_omic = []
    off = 0
    for i, sz in enumerate(omic_sizes):
        h = x_omic[:, off:off + sz]; off += sz
        j = 0
        while f"sig_networks.{i}.{j}.0.weight" in p:
            h = F.elu(lin(h, f"sig_networks.{i}.{j}.0."))
            j += 1
        h_omic.append(h)
    g_feat = torch.stack(h_omic).transpose(0, 1)
    cls_p_enc, patch_p = transformer_p(h_path, _sub(p, "pathomics_encoder."), 256)
    cls_g_enc, patch_g = transformer_g(g_feat, _sub(p, "genomics_encoder."), 256)
    p_in_g, _ = coattention(patch_p.transpose(1, 0), patch_g.transpose(1, 0), patch_g.transpose(1, 0), _sub(p, "P_in_G_Att."))
    g_in_p, _ = coattention(patch_g.transpose(1, 0), patch_p.transpose(1, 0), patch_p.transpose(1, 0), _sub(p, "G_in_P_Att."))
    cls_p_dec, _ = transformer_p(p_in_g.transpose(1, 0), _sub(p, "pathomics_decoder."), 256)
    cls_g_dec, _ = transformer_g(g_in_p.transpose(1, 0), _sub(p, "genomics_decoder."), 256)
    f = torch.cat(((cls_p_enc + cls_p_dec) / 2, (cls_g_enc + cls_g_dec) / 2), dim=1)
    f = torch.relu(lin(torch.relu(lin(f, "mm.0.")), "mm.2."))
    logits = lin(f, "classifier.")
    hazards = torch.sigmoid(logits)
    return logits, hazards, torch.cumprod(1 - hazards, dim=1), cls_p_enc, cls_p_dec, cls_g_enc, cls_g_dec

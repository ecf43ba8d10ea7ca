"""Host-side mirror of the reference's default model (`mode: cmta`, config/config_mine.yaml:57) on the HIP kernels:

  Transformer_P   models/cmta_utils.py:894-924   wrap-pad to a square, cls token, Nystrom layer, PPEG, Nystrom layer, LayerNorm
  Transformer_G   models/cmta_utils.py:927-948   cls token, two Nystrom layers, LayerNorm
  SNN_Block       models/mcat_utils.py:81-95     Linear + ELU + AlphaDropout (omic signature encoders)
  CMTA            models/model.py:714-853        wsi fc -> encoders -> P_in_G / G_in_P co-attention pair -> decoders ->
                                                 concat / bilinear fusion of the averaged cls tokens -> classifier

Same constructors, forward kwargs, return tuples and parameter names (checkpoint format).  TransLayer / PPEG /
NystromAttention / MultiheadAttention are the package's HIP-backed modules; every nn.Linear runs through smml_gemm.
What stays in ATen: ELU / AlphaDropout / Dropout on [B, <= 256] or [B, n, 256] activations, cat / stack, sigmoid, cumprod.
The reference's hard `.cuda()` calls (cmta_utils.py:914,940) become `.to(features.device)`."""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from . import functional as Fh
from .coattention import MultiheadAttention
from .fusion import BilinearFusion
from .nystrom_attention import PPEG, TransLayer


class Transformer_P(nn.Module):
    def __init__(self, feature_dim=512, compute_dtype=None):
        super().__init__()
        self.pos_layer = PPEG(dim=feature_dim)
        self.cls_token = nn.Parameter(torch.randn(1, 1, feature_dim))
        nn.init.normal_(self.cls_token, std=1e-6)
        self.layer1 = TransLayer(dim=feature_dim, compute_dtype=compute_dtype)      # compute_dtype: extension, see nystrom_attention.TransLayer
        self.layer2 = TransLayer(dim=feature_dim, compute_dtype=compute_dtype)
        self.norm = nn.LayerNorm(feature_dim)

    def forward(self, features):
        H = features.shape[1]
        _H = _W = int(np.ceil(np.sqrt(H)))
        add_length = _H * _W - H
        h = torch.cat([features, features[:, :add_length, :]], dim=1)          # wrap-pad to a square (:908-910)
        B = h.shape[0]
        h = torch.cat((self.cls_token.expand(B, -1, -1).to(h.device), h), dim=1)
        h = self.layer1(h)
        h = self.pos_layer(h, _H, _W)
        h = self.layer2(h)
        h = Fh.layer_norm(h, self.norm.weight, self.norm.bias, self.norm.eps)
        return h[:, 0], h[:, 1:]


class Transformer_G(nn.Module):
    def __init__(self, feature_dim=512, compute_dtype=None):
        super().__init__()
        self.cls_token = nn.Parameter(torch.randn(1, 1, feature_dim))
        nn.init.normal_(self.cls_token, std=1e-6)
        self.layer1 = TransLayer(dim=feature_dim, compute_dtype=compute_dtype)      # compute_dtype: extension, see nystrom_attention.TransLayer
        self.layer2 = TransLayer(dim=feature_dim, compute_dtype=compute_dtype)
        self.norm = nn.LayerNorm(feature_dim)

    def forward(self, features):
        h = torch.cat((self.cls_token.expand(features.shape[0], -1, -1).to(features.device), features), dim=1)
        h = self.layer1(h)
        h = self.layer2(h)
        h = Fh.layer_norm(h, self.norm.weight, self.norm.bias, self.norm.eps)
        return h[:, 0], h[:, 1:]


def SNN_Block(dim1, dim2, dropout=0.25):
    return nn.Sequential(nn.Linear(dim1, dim2), nn.ELU(), nn.AlphaDropout(p=dropout, inplace=False))


def _snn(seq: nn.Sequential, x):
    for blk in seq:
        x = blk[2](blk[1](Fh.linear(x, blk[0].weight, blk[0].bias)))
    return x


class CMTA(nn.Module):
    def __init__(self, args, fusion='concat', omic_sizes=[100, 100, 100, 131], n_classes=4, model_size_wsi: str = 'small',
                 model_size_omic: str = 'small', dropout=0.25):
        super().__init__()
        self.args = args
        self.fusion = fusion
        self.omic_sizes = omic_sizes
        self.n_classes = args.label_dim
        self.size_dict_WSI = {"small": [1024, 256, 256], "big": [1024, 512, 384]}
        self.size_dict_omic = {'small': [256, 256], 'big': [1024, 1024, 1024, 256]}
        size = self.size_dict_WSI[model_size_wsi]
        self.wsi_net = nn.Sequential(nn.Linear(size[0], size[1]), nn.ReLU(), nn.Dropout(0.25))
        hidden = self.size_dict_omic[model_size_omic]
        sig_networks = []
        for input_dim in omic_sizes:
            fc_omic = [SNN_Block(dim1=input_dim, dim2=hidden[0])]
            for i, _ in enumerate(hidden[1:]):
                fc_omic.append(SNN_Block(dim1=hidden[i], dim2=hidden[i + 1], dropout=0.25))
            sig_networks.append(nn.Sequential(*fc_omic))
        self.sig_networks = nn.ModuleList(sig_networks)
        cd = getattr(args, "nystrom_compute_dtype", None)          # extension key: None (exact fp32) | 'bf16' | 'fp16' for the Nystrom blocks
        self.pathomics_encoder = Transformer_P(feature_dim=hidden[-1], compute_dtype=cd)
        self.pathomics_decoder = Transformer_P(feature_dim=hidden[-1], compute_dtype=cd)
        self.P_in_G_Att = MultiheadAttention(embed_dim=256, num_heads=1)
        self.G_in_P_Att = MultiheadAttention(embed_dim=256, num_heads=1)
        self.genomics_encoder = Transformer_G(feature_dim=hidden[-1], compute_dtype=cd)
        self.genomics_decoder = Transformer_G(feature_dim=hidden[-1], compute_dtype=cd)
        if self.fusion == 'concat':
            self.mm = nn.Sequential(nn.Linear(256 * 2, size[2]), nn.ReLU(), nn.Linear(size[2], size[2]), nn.ReLU())
        elif self.fusion == 'bilinear':
            self.mm = BilinearFusion(dim1=256, dim2=256, scale_dim1=8, scale_dim2=8, mmhid=256)
        else:
            self.mm = None
        self.classifier = nn.Linear(size[2], self.n_classes)

    def forward(self, **kwargs):
        x_path = kwargs['x_path']
        x_omic_all = kwargs['x_omic']
        sizes = self.omic_sizes
        x_omic = [x_omic_all[:, sum(sizes[:i]):sum(sizes[:i + 1])] for i in range(len(sizes))]
        # wsi fc: Linear + ReLU fused in the GEMM epilogue, Dropout(0.25) in ATen (train mode only)
        h_path_bag = self.wsi_net[2](Fh.linear(x_path.float(), self.wsi_net[0].weight, self.wsi_net[0].bias, act=Fh.ACT_RELU))
        h_omic = [_snn(self.sig_networks[i], x.float()) for i, x in enumerate(x_omic)]
        genomics_features = torch.stack(h_omic).transpose(0, 1)                 # [B, 4, 256]
        pathomics_features = h_path_bag                                         # [B, n, 256]
        cls_p_enc, patch_p = self.pathomics_encoder(pathomics_features)
        cls_g_enc, patch_g = self.genomics_encoder(genomics_features)
        p_in_g, _ = self.P_in_G_Att(patch_p.transpose(1, 0), patch_g.transpose(1, 0), patch_g.transpose(1, 0))
        g_in_p, _ = self.G_in_P_Att(patch_g.transpose(1, 0), patch_p.transpose(1, 0), patch_p.transpose(1, 0))
        cls_p_dec, _ = self.pathomics_decoder(p_in_g.transpose(1, 0))
        cls_g_dec, _ = self.genomics_decoder(g_in_p.transpose(1, 0))
        a, b = (cls_p_enc + cls_p_dec) / 2, (cls_g_enc + cls_g_dec) / 2
        if self.fusion == "concat":
            f = Fh.linear(torch.cat((a, b), dim=1), self.mm[0].weight, self.mm[0].bias, act=Fh.ACT_RELU)
            f = Fh.linear(f, self.mm[2].weight, self.mm[2].bias, act=Fh.ACT_RELU)
        elif self.fusion == "bilinear":
            f = self.mm(a, b)
        else:
            raise NotImplementedError("Fusion [{}] is not implemented".format(self.fusion))
        logits = Fh.linear(f, self.classifier.weight, self.classifier.bias)
        hazards = torch.sigmoid(logits)
        S = torch.cumprod(1 - hazards, dim=1)
        return logits, hazards, S, cls_p_enc, cls_p_dec, cls_g_enc, cls_g_dec

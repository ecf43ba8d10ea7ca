"""Host-side mirror of models/fusion.py:6-63 BilinearFusion (gated multimodal units + Kronecker fusion), built by
define_bifusion when fusion_type == 'pofusion' (models/model.py:131-137,458-465).

Same constructor, forward(vec1, vec2) and parameter names.  The Linear layers and the two nn.Bilinear gates
(x1^T W_o x2 evaluated as a batched GEMM over the 128 output slices) run through the HIP GEMM; the remaining
arithmetic is on [B, <= 16 641]-sized tensors (B = batch per rank <= 8): sigmoid / ReLU, the (d1+1) x (d2+1)
outer product and nn.BatchNorm1d stay torch modules - the latter so that main.py's
SyncBatchNorm.convert_sync_batchnorm (main.py:118) keeps working.  TrilinearFusion_A/B (never constructed by the
reference) are not built."""
from __future__ import annotations

import math

import torch
from torch import nn

from . import functional as Fh


def _init_max_weights(module):
    for m in module.modules():
        if type(m) == nn.Linear:
            stdv = 1. / math.sqrt(m.weight.size(1))
            m.weight.data.normal_(0, stdv)
            m.bias.data.zero_()


def _bilinear(bl: nn.Bilinear, x1, x2):
    """out[b, o] = x1[b]^T W[o] x2[b] + bias[o]: t[o] = x1 @ W[o] on the matrix cores (batch over o), then a
    [B, out, in2]-sized elementwise product and row sum."""
    B = x1.shape[0]
    O, I1, I2 = bl.weight.shape
    t = Fh.matmul4(x1.reshape(1, 1, B, I1), bl.weight.reshape(1, O, I1, I2))         # [1, O, B, I2]
    z = (t[0] * x2.unsqueeze(0)).sum(dim=-1).t()                                    # [B, O]
    return z + bl.bias if bl.bias is not None else z


class BilinearFusion(nn.Module):
    def __init__(self, skip=1, use_bilinear=1, gate1=1, gate2=1, dim1=32, dim2=32, scale_dim1=1, scale_dim2=1, mmhid=64,
                 dropout_rate=0.25):
        super().__init__()
        self.skip, self.use_bilinear, self.gate1, self.gate2 = skip, use_bilinear, gate1, gate2
        self.relu = nn.ReLU(inplace=False)
        dim1_og, dim2_og, dim1, dim2 = dim1, dim2, dim1 // scale_dim1, dim2 // scale_dim2
        skip_dim = dim1 + dim2 + 2 if skip else 0
        self.linear_h1 = nn.Sequential(nn.Linear(dim1_og, dim1), nn.ReLU())
        self.linear_z1 = nn.Bilinear(dim1_og, dim2_og, dim1) if use_bilinear else nn.Sequential(nn.Linear(dim1_og + dim2_og, dim1))
        self.linear_o1 = nn.Sequential(nn.Linear(dim1, dim1), nn.ReLU(), nn.Dropout(p=dropout_rate))
        self.linear_h2 = nn.Sequential(nn.Linear(dim2_og, dim2), nn.ReLU())
        self.linear_z2 = nn.Bilinear(dim1_og, dim2_og, dim2) if use_bilinear else nn.Sequential(nn.Linear(dim1_og + dim2_og, dim2))
        self.linear_o2 = nn.Sequential(nn.Linear(dim2, dim2), nn.ReLU(), nn.Dropout(p=dropout_rate))
        self.post_fusion_dropout = nn.Dropout(p=dropout_rate)
        self.encoder1 = nn.Sequential(nn.Linear((dim1 + 1) * (dim2 + 1), mmhid), nn.BatchNorm1d(mmhid), nn.ReLU(),
                                      nn.Dropout(p=dropout_rate))
        self.encoder2 = nn.Sequential(nn.Linear(mmhid + skip_dim, mmhid), nn.BatchNorm1d(mmhid), nn.ReLU(),
                                      nn.Dropout(p=dropout_rate))
        _init_max_weights(self)

    @staticmethod
    def _lin(seq, x, act=Fh.ACT_NONE):
        return Fh.linear(x, seq[0].weight, seq[0].bias, act=act)

    def _gate(self, z_mod, vec1, vec2):
        if self.use_bilinear:
            return _bilinear(z_mod, vec1, vec2)
        return self._lin(z_mod, torch.cat((vec1, vec2), dim=1))

    def forward(self, vec1, vec2):
        vec1, vec2 = self.relu(vec1), self.relu(vec2)
        if self.gate1:
            h1 = self._lin(self.linear_h1, vec1, Fh.ACT_RELU)
            o1 = self.linear_o1[2](self._lin(self.linear_o1, torch.sigmoid(self._gate(self.linear_z1, vec1, vec2)) * h1, Fh.ACT_RELU))
        else:
            o1 = self.linear_o1[2](self._lin(self.linear_o1, vec1, Fh.ACT_RELU))
        if self.gate2:
            h2 = self._lin(self.linear_h2, vec2, Fh.ACT_RELU)
            o2 = self.linear_o2[2](self._lin(self.linear_o2, torch.sigmoid(self._gate(self.linear_z2, vec1, vec2)) * h2, Fh.ACT_RELU))
        else:
            o2 = self.linear_o2[2](self._lin(self.linear_o2, vec2, Fh.ACT_RELU))
        one = torch.ones(o1.shape[0], 1, device=o1.device, dtype=o1.dtype)      # reference: torch.cuda.FloatTensor(...).fill_(1)
        o1, o2 = torch.cat((o1, one), 1), torch.cat((o2, one), 1)
        # Kronecker product [B, d1 + 1] x [B, d2 + 1] -> [B, (d1 + 1)(d2 + 1)] (fusion.py:58): a batched K = 1 product through the
        # package's own GEMM (forward and both gradients in smml_gemm_f32) - no library GEMM on the path
        Bn = o1.shape[0]
        o12 = Fh.matmul4(o1.reshape(1, Bn, -1, 1), o2.reshape(1, Bn, 1, -1)).reshape(Bn, -1)
        out = self.post_fusion_dropout(o12)
        out = self.encoder1[3](self.encoder1[2](self.encoder1[1](self._lin(self.encoder1, out))))
        if self.skip:
            out = torch.cat((out, o1, o2), 1)
        out = self.encoder2[3](self.encoder2[2](self.encoder2[1](self._lin(self.encoder2, out))))
        return out


def define_bifusion(fusion_type, skip=1, use_bilinear=1, gate1=1, gate2=1, dim1=32, dim2=32, scale_dim1=1, scale_dim2=1,
                    mmhid=32, dropout_rate=0.25):
    """models/model.py:131-137."""
    if fusion_type == 'pofusion':
        return BilinearFusion(skip=skip, use_bilinear=use_bilinear, gate1=gate1, gate2=gate2, dim1=dim1, dim2=dim2,
                              scale_dim1=scale_dim1, scale_dim2=scale_dim2, mmhid=mmhid, dropout_rate=dropout_rate)
    raise NotImplementedError('fusion type [%s] is not found' % fusion_type)

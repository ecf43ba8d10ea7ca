"""Host-side mirror of models/DeformCrossTransMIL.py (FusionNet :28-38, DeformCrossTransLayer :40-77,
DeformCrossTransMIL :79-160, Pooler :169-202) on the HIP kernels; same constructors, forward
signatures, return tuples and parameter names.

Additive knobs (reference defaults kept): `args.input_path_dim` (config/config_mine.yaml:25, default 1024)
sizes `_fc1`; the token count is taken from the bag instead of the hard-wired 2500
(DeformCrossTransMIL.py:104); `grid_hw` is forwarded to the 2-D attention; `args.deform_compute_dtype` (absent / None | 'bf16' |
'fp16') selects the 16-bit compute mode of the fused attention core (BASELINE config 4), `args.deform_cpb_table` its table mode
(the position-bias MLP evaluated on a grid once per call and interpolated per pair: include/smml.h).  `args.wrap_pad_to_square` (off by default;
SURVEY.md 8(f) row 4) lets the 2-D branch take bags whose instance count is not a square: both token streams are
wrap-padded to the next square as TransMIL does (models/mil.py:232-235), attended, and cropped back to N.

Differences in mechanism, not in values: the 2500x tiled omic matrix is never fed through a GEMM - the
fusion layer is evaluated as path @ W[:, :C]^T + (omic @ W[:, C:]^T + b) with the second term broadcast
per bag - but the tiled tensor is still returned (BatchLoss consumes it, utils/loss.py:22)."""
from __future__ import annotations

import math

import torch
from torch import nn

from . import functional as Fh
from .deform_attention import DeformCrossAttention1D, DeformCrossAttention2D


class FusionNet(nn.Module):
    def __init__(self, feature_dim=128):
        super().__init__()
        self.fusion_layer = nn.Linear(feature_dim * 2, feature_dim)

    def forward(self, gene_features, image_features):
        """cat((gene_features, image_features), -1) -> Linear.  `image_features` may be the tiled [B, N, C]
        tensor (reference call) or the un-tiled [B, C] vector (broadcast per bag)."""
        C = self.fusion_layer.out_features
        w = self.fusion_layer.weight
        if image_features.dim() == 3:
            image_features = image_features[:, 0]          # every row of the tile is the same vector
        row_bias = Fh.linear(image_features, w[:, C:], self.fusion_layer.bias)          # [B, C]
        return Fh.linear(gene_features, w[:, :C], row_bias, rows_per_bias=gene_features.shape[1])


class DeformCrossTransLayer(nn.Module):
    def __init__(self, norm_layer=nn.LayerNorm, dim=128, grid_hw=None, compute_dtype=None, cpb_table=False):
        super().__init__()
        self.norm = norm_layer(dim)
        self.attn2d = DeformCrossAttention2D(dim=128, dim_head=64, heads=8, dropout=0.1, downsample_factor=4,
                                             offset_scale=4, offset_groups=8, offset_kernel_size=6, grid_hw=grid_hw,
                                             compute_dtype=compute_dtype, cpb_table=cpb_table)
        self.attn1d = DeformCrossAttention1D(dim=128, downsample_factor=4, offset_scale=2, offset_kernel_size=6,
                                             compute_dtype=compute_dtype, cpb_table=cpb_table)

    def forward(self, x1, x2, attn_dim, return_vgrid):
        n = self.norm
        a = Fh.layer_norm(x1, n.weight, n.bias, n.eps)
        b = Fh.layer_norm(x2, n.weight, n.bias, n.eps)
        if attn_dim == 1:
            return self.attn1d.forward_tokens(a, b, False, residual=x1)
        if attn_dim == 2:
            return self.attn2d.forward_tokens(a, b, bool(return_vgrid), residual=x1)
        raise ValueError(f"attn_dim must be 1 or 2 (got {attn_dim})")


class Pooler(nn.Module):
    def __init__(self, hidden_size):
        super().__init__()
        self.dense = nn.Linear(hidden_size, hidden_size)
        self.activation = nn.Tanh()

    def forward(self, hidden_states):
        avg = Fh.token_mean(hidden_states)
        return Fh.linear(avg, self.dense.weight, self.dense.bias, act=Fh.ACT_TANH)


class DeformCrossTransMIL(nn.Module):
    def __init__(self, args, n_classes=4):
        super().__init__()
        # extension key (absent = None = fp32-grade everywhere).  Only the fused attention core and its output projection change arithmetic:
        # everything UPSTREAM of the sample positions (_fc1, fusion, LayerNorm, to_q, the offsets network) stays fp32-grade, so vgrid and the
        # sampler's integer corner / mask path are bit-identical to the fp32-grade path (north_star: integer paths bit-exact)
        cd = getattr(args, "deform_compute_dtype", None)
        self.fusion_layer = FusionNet(feature_dim=128)
        in_dim = int(getattr(args, "input_path_dim", 1024) or 1024)
        self._fc1 = nn.Sequential(nn.Linear(in_dim, args.path_dim), nn.ReLU())
        self.cls_token = nn.Parameter(torch.randn(1, 1, args.path_dim))
        self.args = args
        self.n_classes = n_classes
        # 'bf16' | 'fp16' runs the fused attention core and its output projection in the 16-bit compute mode
        # `args.deform_cpb_table` (absent / False | 'forward' | True; with a 16-bit compute dtype only): position bias from a table of the MLP
        self.layer3 = DeformCrossTransLayer(dim=args.path_dim, grid_hw=getattr(args, "grid_hw", None), compute_dtype=cd,
                                            cpb_table=getattr(args, "deform_cpb_table", False))
        self.norm = nn.LayerNorm(args.path_dim)
        self._fc2 = nn.Linear(args.path_dim, self.n_classes)
        self.pooler = Pooler(args.path_dim)
        self.multimodal_projection = nn.Linear(args.path_dim, self.args.path_dim)

    def prefetch(self, n_tokens: int) -> None:
        """Work that depends on the parameters only and can run beside the first layers: the position bias's region tables."""
        if self.args.attn_dim == 2 and not bool(getattr(self.args, "wrap_pad_to_square", False)):
            self.layer3.attn2d.prefetch_regions(n_tokens)

    def forward(self, path, omic):
        self.prefetch(path.shape[1])
        return self.forward_features(Fh.linear(path.float(), self._fc1[0].weight, self._fc1[0].bias, act=Fh.ACT_RELU), omic)   # [B, N, C]

    def forward_features(self, path, omic):
        """Everything after relu(_fc1(bag)): DeformPathomicNet evaluates the _fc1 of both branches in one launch over the shared bag
        (functional.dual_linear_relu) and enters here."""
        omic = omic.float()
        N = path.shape[1]
        h = self.fusion_layer(path, omic)
        vgrid = None
        if self.args.attn_dim == 1:
            B = h.shape[0]
            cls = self.cls_token.expand(B, -1, -1).to(h.device)
            h = torch.cat((cls, h), dim=1)
            pth = torch.cat((cls, path), dim=1)
            h = self.layer3(h, pth, 1, self.args.return_vgrid)
            h = Fh.layer_norm(h[:, :1], self.norm.weight, self.norm.bias, self.norm.eps)[:, 0]
            logits = Fh.linear(h, self._fc2.weight, self._fc2.bias)
        elif self.args.attn_dim == 2:
            pth = path
            side = int(math.ceil(math.sqrt(N)))
            wrap = bool(getattr(self.args, "wrap_pad_to_square", False)) and side * side != N and getattr(self.args, "grid_hw", None) is None
            if wrap:                                   # h[:, :add] appended, as mil.py:232-235 does for TransMIL
                add = side * side - N
                h = torch.cat((h, h[:, :add]), dim=1)
                pth = torch.cat((path, path[:, :add]), dim=1)
            if self.args.return_vgrid:
                h, vgrid = self.layer3(h, pth, 2, True)
            else:
                h = self.layer3(h, pth, 2, False)
            if wrap:
                h = h[:, :N]
            avg = Fh.layer_norm_token_mean(h, self.norm.weight, self.norm.bias, self.norm.eps)
            h = Fh.linear(avg, self.pooler.dense.weight, self.pooler.dense.bias, act=Fh.ACT_TANH)
            logits = Fh.linear(h, self._fc2.weight, self._fc2.bias)
        else:
            raise ValueError(f"attn_dim must be 1 or 2 (got {self.args.attn_dim})")
        encoded = Fh.linear(h, self.multimodal_projection.weight, self.multimodal_projection.bias)
        path_grads = None
        if self.args.return_vgrid:
            if vgrid is None:
                raise RuntimeError("return_vgrid requires attn_dim == 2 (the reference raises NameError here, "
                                   "DeformCrossTransMIL.py:158)")
            omic_tiled = Fh.tile_tokens(omic, N)                 # omic.unsqueeze(1).repeat(1, N, 1), :104
            return encoded, logits, path_grads, omic_tiled, vgrid
        return encoded, logits, path_grads

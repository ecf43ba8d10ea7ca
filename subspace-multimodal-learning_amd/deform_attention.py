"""Host-side mirror of the reference's deformable cross-attention modules on the HIP kernels.

Same constructor keywords / defaults, forward signatures, tensor layouts and parameter names
(= checkpoint format) as
  models/DeformableAttention2D.py:161-325  DeformCrossAttention2D (+ CPB :120-157, Scale :110-116)
  models/DeformableAttention1D.py:106-240  DeformCrossAttention1D (+ CPB :60-102)
so `state_dict`s are interchangeable.  The nn.Conv / nn.Linear sub-modules only hold the parameters
(identical default initialisation); the arithmetic runs in the kernels behind `functional`.

Additive knobs (reference defaults kept): `grid_hw` - the reference hard-wires a 50 x 50 token grid
(DeformableAttention2D.py:239-240,318); here the grid defaults to the square root of the token count.
`compute_dtype` (None | 'bf16' | 'fp16', default None): the 16-bit compute mode of the fused attention core
(csrc/deform_attn16.hip; BASELINE config 4 names bf16) - parameters, inputs, outputs and gradients stay fp32.
Corrected-semantics switches, OFF by default (bit-parity with the reference needs them off):
  * DeformCrossAttention2D(consistent_grid_norm=True): the reference normalises sample positions with the
    align_corners=True formula 2 v / (t - 1) - 1 (:100-108,265; x divided by rows - 1, y by cols - 1) but samples with
    F.grid_sample(align_corners=False) (:268) - ~21 % of the corner fetches fall outside the map at initialisation.  With
    the switch both the sample positions and the query grid use the pixel-centre convention (2 p + 1) / size - 1, x scaled
    by the number of columns and y by the number of rows.
  * DeformCrossAttention1D(true_1d_sampling=True): the reference's grid_sample_1d (:36-43) lays the features out
    [H = n, W = 1] while the grid carries (x = vs, y = 0), so every "sampled" key is the centre token scaled; with the switch
    the map is [H = 1, W = n] and vs interpolates along the tokens.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import torch
from torch import nn

from . import functional as Fh


def _default(v, d):
    return d if v is None else v


class Scale(nn.Module):
    def __init__(self, scale):
        super().__init__()
        self.scale = scale

    def forward(self, x):
        return x * self.scale


class CPB(nn.Module):
    """Parameter holder of the continuous position bias MLP (in -> dim -> dim -> heads // offset_groups)."""

    def __init__(self, dim, *, heads, offset_groups, depth, in_dim=2, log_distance=True):
        super().__init__()
        self.heads, self.offset_groups, self.log_distance = heads, offset_groups, log_distance
        self.mlp = nn.ModuleList([nn.Sequential(nn.Linear(in_dim, dim), nn.ReLU())])
        for _ in range(depth - 1):
            self.mlp.append(nn.Sequential(nn.Linear(dim, dim), nn.ReLU()))
        self.mlp.append(nn.Linear(dim, heads // offset_groups))
        if depth != 2 or dim != 32:
            raise NotImplementedError("the position-bias kernels are built for depth 2 and width 32 (dim = 128)")

    def tensors(self):
        m = self.mlp
        return (m[0][0].weight, m[0][0].bias, m[1][0].weight, m[1][0].bias, m[2].weight, m[2].bias)


def _dropout_args(mod, device=None):
    """nn.Dropout semantics on the attention probabilities: active in train() with p > 0; the 64-bit seed of the
    in-kernel counter-based mask is drawn from torch's default (CPU) generator, so torch.manual_seed reproduces it.
    While the step is being captured in a hipGraph the host seed would be the same in every replay: the call then also gets a
    device-resident offset that a captured counter bumps per call and per replay (functional.graph_seed_offset)."""
    p = float(mod.dropout.p)
    if not mod.training or p <= 0.0:
        return {"dropout_p": 0.0, "dropout_seed": 0}
    seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())
    mod.last_dropout_seed = seed
    off = Fh.graph_seed_offset(device) if device is not None and torch.device(device).type == "cuda" else None
    return {"dropout_p": p, "dropout_seed": seed, "dropout_seed_offset": off}


_GRID_CACHE = {}
_GRAD_FORK = __import__("os").environ.get("SMML_GRAD_FORK", "1") != "0"       # measurement switch (functional.GradFork)


def _grid_queries_2d(Hh: int, Ww: int, device) -> torch.Tensor:
    """Cached per (rows, cols, device): a constant of the token grid (eight small kernels per forward otherwise)."""
    key = (Hh, Ww, str(device))
    g = _GRID_CACHE.get(key)
    if g is None:
        g = _GRID_CACHE[key] = _build_grid_queries_2d(Hh, Ww, device)
    return g


def _build_grid_queries_2d(Hh: int, Ww: int, device) -> torch.Tensor:
    """Normalised query grid [N, 2] = (x, y) per token, restating create_grid_like + normalize_grid(dim=0)
    (DeformableAttention2D.py:88-108,296-297): x is divided by (rows - 1), y by (cols - 1)."""
    qx = 2.0 * torch.arange(Ww, dtype=torch.float32, device=device) / max(Hh - 1, 1) - 1.0
    qy = 2.0 * torch.arange(Hh, dtype=torch.float32, device=device) / max(Ww - 1, 1) - 1.0
    return torch.stack((qx.view(1, Ww).expand(Hh, Ww), qy.view(Hh, 1).expand(Hh, Ww)), dim=-1).reshape(Hh * Ww, 2).contiguous()


class DeformCrossAttention2D(nn.Module):
    def __init__(self, *, dim, dim_head=64, heads=8, dropout=0., downsample_factor=4, offset_scale=4,
                 offset_groups=8, offset_kernel_size=6, group_queries=True, group_key_values=True,
                 grid_hw: Optional[Tuple[int, int]] = None, consistent_grid_norm: bool = False, compute_dtype=None,
                 cpb_table: bool = False):
        super().__init__()
        self.consistent_grid_norm = bool(consistent_grid_norm)
        Fh._dtype16(compute_dtype)             # validates: None | 'bf16' | 'fp16'
        self.compute_dtype = compute_dtype     # additive: None = the fp32-grade path, else the 16-bit compute mode of the fused core
        if cpb_table and compute_dtype is None:
            raise ValueError("cpb_table belongs to the 16-bit compute modes: pass compute_dtype='bf16' or 'fp16'")
        self.cpb_table = cpb_table             # additive: False | True ('full') | 'forward' - position bias from a table of the MLP (functional.deform_attention)
        offset_scale = _default(offset_scale, downsample_factor)
        assert offset_kernel_size >= downsample_factor, \
            'offset kernel size must be greater than or equal to the downsample factor'
        assert (offset_kernel_size - downsample_factor) % 2 == 0
        offset_groups = _default(offset_groups, heads)
        assert heads % offset_groups == 0
        inner_dim = dim_head * heads
        self.scale = dim_head ** -0.5
        self.heads, self.offset_groups = heads, offset_groups
        self.dim, self.dim_head = dim, dim_head
        offset_dims = inner_dim // offset_groups
        self.downsample_factor = downsample_factor
        self.offset_kernel_size, self.offset_scale = offset_kernel_size, float(offset_scale)
        self.group_queries, self.group_key_values = group_queries, group_key_values
        self.grid_hw = grid_hw
        self.to_offsets = nn.Sequential(
            nn.Conv2d(offset_dims, offset_dims, offset_kernel_size, groups=offset_dims, stride=downsample_factor,
                      padding=(offset_kernel_size - downsample_factor) // 2),
            nn.GELU(),
            nn.Conv2d(offset_dims, 2, 1, bias=False),
            nn.Tanh(),
            Scale(offset_scale))
        self.rel_pos_bias = CPB(dim // 4, offset_groups=offset_groups, heads=heads, depth=2)
        self.dropout = nn.Dropout(dropout)
        self.to_q = nn.Conv2d(dim, inner_dim, 1, groups=offset_groups if group_queries else 1, bias=False)
        self.to_k = nn.Conv2d(dim, inner_dim, 1, groups=offset_groups if group_key_values else 1, bias=False)
        self.to_v = nn.Conv2d(dim, inner_dim, 1, groups=offset_groups if group_key_values else 1, bias=False)
        self.to_out = nn.Conv2d(inner_dim, dim, 1)

    def _grid(self, n: int) -> Tuple[int, int]:
        if self.grid_hw is not None:
            return self.grid_hw
        s = int(round(math.sqrt(n)))
        if s * s != n:
            raise ValueError(f"token count {n} is not a square; pass grid_hw=(rows, cols)")
        return s, s

    def _region_pmax(self, Hh: int, Ww: int) -> float:
        """Half-width of the square of signed-log offsets the position bias can be asked for: |gq| and |vs| are bounded by the grids'
        shapes and tanh . offset_scale (normalize_grid divides x by rows - 1, y by cols - 1)."""
        L = Fh.capi.lib()
        th = L.smml_offsets_out_len(Hh, self.offset_kernel_size, self.downsample_factor)
        tw = L.smml_offsets_out_len(Ww, self.offset_kernel_size, self.downsample_factor)
        lo_q, lo_k = max(min(Hh, Ww) - 1, 1), max(min(th, tw) - 1, 1)
        gqb = max(1.0, abs(2.0 * (max(Hh, Ww) - 1) / lo_q - 1.0))
        vsb = max(abs(2.0 * (max(th, tw) - 1 + self.offset_scale) / lo_k - 1.0), 1.0 + 2.0 * self.offset_scale / lo_k)
        return Fh.table_pmax(gqb, vsb)

    def _regions_apply(self) -> bool:
        return (Fh.CPB_REGIONS and not self.cpb_table and not self.consistent_grid_norm and self.heads == self.offset_groups
                and self.rel_pos_bias.mlp[0][0].weight.is_cuda)          # fp32-grade core and the 16-bit compute modes alike

    def prefetch_regions(self, n_tokens: int) -> None:
        """Starts the build of the position bias's region tables on a side stream (functional.RegionPrefetch); the next forward_tokens on
        a bag of n_tokens joins it.  Optional: without it the tables are built in front of the attention launch."""
        self._prefetch = None
        if not (self._regions_apply() and Fh.REGION_PREFETCH):
            return
        Hh, Ww = self._grid(n_tokens)
        self._prefetch = Fh.RegionPrefetch(*self.rel_pos_bias.tensors(), self._region_pmax(Hh, Ww))

    def forward_tokens(self, x1t, x2t, return_vgrid=False, residual=None):
        """Token-major entry: x1t (queries) / x2t (keys, values) [B, N, C] -> [B, N, C] (+ residual)."""
        B, N, C = x1t.shape
        Hh, Ww = self._grid(N)
        G, H = self.offset_groups, self.heads
        q = Fh.grouped_pointwise(x1t, self.to_q.weight, G if self.group_queries else 1)            # [B, N, inner]
        fork = Fh.GradFork() if (q.requires_grad and not self.consistent_grid_norm and _GRAD_FORK) else None       # q's two gradients meet in one buffer
        vgrid, vs = Fh.offsets(q.view(B, Hh, Ww, -1), self.to_offsets[0].weight, self.to_offsets[0].bias,
                               self.to_offsets[2].weight.reshape(2, -1), groups=G, ks=self.offset_kernel_size,
                               r=self.downsample_factor, posdim=2, offset_scale=self.offset_scale, fork=fork)
        gq = _grid_queries_2d(Hh, Ww, x1t.device)
        if self.consistent_grid_norm:          # corrected semantics (off by default): pixel-centre convention on both sides
            th, tw = vgrid.shape[-2:]
            vs = torch.stack(((2.0 * vgrid[:, 0] + 1.0) / tw - 1.0, (2.0 * vgrid[:, 1] + 1.0) / th - 1.0), dim=-1)
            vs = vs.reshape(B * G, th * tw, 2).contiguous()
            ar = lambda n: (2.0 * torch.arange(n, dtype=torch.float32, device=x1t.device) + 1.0) / n - 1.0
            gq = torch.stack((ar(Ww).view(1, Ww).expand(Hh, Ww), ar(Hh).view(Hh, 1).expand(Hh, Ww)), dim=-1).reshape(Hh * Ww, 2).contiguous()
        kv = Fh.bilinear_sample(x2t.reshape(B, Hh, Ww, C), vs, groups=G, posdim=2)                  # [B, J, C]
        gk = G if self.group_key_values else 1
        k = Fh.grouped_pointwise(kv, self.to_k.weight, gk)
        v = Fh.grouped_pointwise(kv, self.to_v.weight, gk)
        tab = {}
        # |gq|, |vs| are bounded by the grids' shapes and tanh . offset_scale (normalize_grid divides x by rows - 1, y by cols - 1)
        th, tw = vgrid.shape[-2:]
        lo_q, lo_k = max(min(Hh, Ww) - 1, 1), max(min(th, tw) - 1, 1)
        gqb = max(1.0, abs(2.0 * (max(Hh, Ww) - 1) / lo_q - 1.0))
        vsb = max(abs(2.0 * (max(th, tw) - 1 + self.offset_scale) / lo_k - 1.0), 1.0 + 2.0 * self.offset_scale / lo_k)
        if not self.cpb_table and not self.consistent_grid_norm:
            # the square the region tables of the position bias cover (no host sync), and the build started by prefetch_regions, if any
            tab = {"cpb_region_pmax": self._region_pmax(Hh, Ww), "cpb_region_prefetch": getattr(self, "_prefetch", None)}
            self._prefetch = None
        if self.cpb_table:
            tab = {"cpb_table": self.cpb_table, "cpb_table_pmax": None if self.consistent_grid_norm else Fh.table_pmax(gqb, vsb),   # None: from the data
                   "cpb_table_grid": (Hh, Ww)}                 # gq is a regular grid in both normalisations
        o = Fh.deform_attention(q, k, v, vs, gq, *self.rel_pos_bias.tensors(), heads=H, groups=G, scale=self.scale,
                                compute_dtype=self.compute_dtype, fork=fork, **tab, **_dropout_args(self, q.device))
        # the output projection follows the core's compute mode (single-term 16-bit operands, fp32 accumulation and storage)
        out = Fh.linear(o, self.to_out.weight.reshape(self.dim, -1), self.to_out.bias, residual=residual, prec=Fh.prec16(self.compute_dtype))
        return (out, vgrid) if return_vgrid else out

    def forward(self, x1, x2, return_vgrid=False):
        """x1, x2 [B, C, N] channels-first as in the reference -> [B, C, N] (and vgrid [(B g), 2, th, tw])."""
        r = self.forward_tokens(x1.transpose(1, 2), x2.transpose(1, 2), return_vgrid)
        if return_vgrid:
            return r[0].transpose(1, 2), r[1]
        return r.transpose(1, 2)


class DeformCrossAttention1D(nn.Module):
    def __init__(self, *, dim, dim_head=64, heads=8, dropout=0., downsample_factor=4, offset_scale=None,
                 offset_groups=4, offset_kernel_size=6, cpb_log_distance=True, group_queries=False,
                 group_key_values=False, true_1d_sampling: bool = False, compute_dtype=None, cpb_table: bool = False):
        super().__init__()
        self.true_1d_sampling = bool(true_1d_sampling)
        Fh._dtype16(compute_dtype)
        self.compute_dtype = compute_dtype
        if cpb_table and compute_dtype is None:
            raise ValueError("cpb_table belongs to the 16-bit compute modes: pass compute_dtype='bf16' or 'fp16'")
        self.cpb_table = cpb_table
        offset_scale = _default(offset_scale, downsample_factor)
        assert offset_kernel_size >= downsample_factor, \
            'offset kernel size must be greater than or equal to the downsample factor'
        assert (offset_kernel_size - downsample_factor) % 2 == 0
        offset_groups = _default(offset_groups, heads)
        assert heads % offset_groups == 0
        if not cpb_log_distance and cpb_table:
            raise NotImplementedError("the table modes are built for the log-distance form of the position bias only")
        self.cpb_log_distance = bool(cpb_log_distance)
        inner_dim = dim_head * heads
        self.scale = dim_head ** -0.5
        self.heads, self.offset_groups = heads, offset_groups
        self.dim, self.dim_head = dim, dim_head
        offset_dims = inner_dim // offset_groups
        self.downsample_factor = downsample_factor
        self.offset_kernel_size, self.offset_scale = offset_kernel_size, float(offset_scale)
        self.group_queries, self.group_key_values = group_queries, group_key_values
        self.to_offsets = nn.Sequential(
            nn.Conv1d(offset_dims, offset_dims, offset_kernel_size, groups=offset_dims, stride=downsample_factor,
                      padding=(offset_kernel_size - downsample_factor) // 2),
            nn.GELU(),
            nn.Conv1d(offset_dims, 1, 1, bias=False),
            nn.Identity(),          # placeholder for the reference's Rearrange('b 1 n -> b n') (no parameters)
            nn.Tanh(),
            Scale(offset_scale))
        self.rel_pos_bias = CPB(dim // 4, offset_groups=offset_groups, heads=heads, depth=2, in_dim=1,
                                log_distance=cpb_log_distance)
        self.dropout = nn.Dropout(dropout)
        self.to_q = nn.Conv1d(dim, inner_dim, 1, groups=offset_groups if group_queries else 1, bias=False)
        self.to_k = nn.Conv1d(dim, inner_dim, 1, groups=offset_groups if group_key_values else 1, bias=False)
        self.to_v = nn.Conv1d(dim, inner_dim, 1, groups=offset_groups if group_key_values else 1, bias=False)
        self.to_out = nn.Conv1d(inner_dim, dim, 1)

    def forward_tokens(self, x1t, x2t, return_vgrid=False, residual=None):
        B, n, C = x1t.shape
        G, H = self.offset_groups, self.heads
        q = Fh.grouped_pointwise(x1t, self.to_q.weight, G if self.group_queries else 1)
        vgrid, vs = Fh.offsets(q.view(B, 1, n, -1), self.to_offsets[0].weight, self.to_offsets[0].bias,
                               self.to_offsets[2].weight.reshape(1, -1), groups=G, ks=self.offset_kernel_size,
                               r=self.downsample_factor, posdim=1, offset_scale=self.offset_scale)
        # bug-compatible degenerate sampling (DeformableAttention1D.py:36-43): map laid out [H = n, W = 1]; the corrected
        # switch lays it out [H = 1, W = n] so that vs interpolates along the tokens
        kv = Fh.bilinear_sample(x2t.reshape(B, 1, n, C) if self.true_1d_sampling else x2t.reshape(B, n, 1, C), vs, groups=G, posdim=1)
        gk = G if self.group_key_values else 1
        k = Fh.grouped_pointwise(kv, self.to_k.weight, gk)
        v = Fh.grouped_pointwise(kv, self.to_v.weight, gk)
        seq = (2.0 * torch.arange(n, dtype=torch.float32, device=x1t.device) / max(n - 1, 1) - 1.0).view(n, 1)
        tab = {}
        if self.cpb_table:
            t = vgrid.shape[-1]
            tab = {"cpb_table": self.cpb_table, "cpb_table_pmax": Fh.table_pmax(1.0, 1.0 + 2.0 * self.offset_scale / max(t - 1, 1))}
        o = Fh.deform_attention(q, k, v, vs, seq.contiguous(), *self.rel_pos_bias.tensors(), heads=H, groups=G,
                                scale=self.scale, compute_dtype=self.compute_dtype, log_distance=self.cpb_log_distance, **tab,
                                **_dropout_args(self, q.device))
        out = Fh.linear(o, self.to_out.weight.reshape(self.dim, -1), self.to_out.bias, residual=residual, prec=Fh.prec16(self.compute_dtype))
        return (out, vgrid) if return_vgrid else out

    def forward(self, x1, x2, return_vgrid=False):
        r = self.forward_tokens(x1.transpose(1, 2), x2.transpose(1, 2), return_vgrid)
        if return_vgrid:
            return r[0].transpose(1, 2), r[1]
        return r.transpose(1, 2)

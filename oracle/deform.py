"""Oracle (TEST INFRASTRUCTURE ONLY): deformable cross-attention, 2-D and 1-D.

Plain PyTorch fp32 restatement, on token-major tensors, of
  models/DeformableAttention2D.py:88-325  (create_grid_like, normalize_grid,
                                           CPB, DeformCrossAttention2D)
  models/DeformableAttention1D.py:36-240  (grid_sample_1d, normalize_grid,
                                           CPB, DeformCrossAttention1D)
The restatement is parametrised in the token-grid size (the reference is
hard-wired to 50x50, DeformableAttention2D.py:239-240,318) and evaluates the
continuous position bias in query chunks so that N = 10 000 fits host RAM.

Parameters are passed as a dict keyed by the reference's parameter names
(= checkpoint format).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------
# small pieces
# --------------------------------------------------------------------------
def signed_log(d: torch.Tensor) -> torch.Tensor:
    """sign(d) * log(|d| + 1)   -- DeformableAttention2D.py:148, DeformableAttention1D.py:93."""
    return torch.sign(d) * torch.log(d.abs() + 1)


# Tests set this to a dict to learn the NATURAL SCALE of d mlp.2.bias: that gradient is sum over all (query, key) pairs
# of d bias, which is exactly 0 in exact arithmetic (softmax is shift invariant), so its size can only be judged against
# sum |d bias| - collected here per bias tensor (key: id of the parameter tensor).
GRAD_PROBE = None


# Imposed decisions (parity tests).  The module is piecewise linear in three places - the two ReLU layers of the position-bias
# MLP and the cell a bilinear sample falls into - and wherever a pre-activation / pixel coordinate lies within fp32 rounding of the
# kink, two fp32 evaluations (the reference's and the kernels') may legitimately decide differently; the GRADIENT then jumps by a
# finite amount.  To compare gradients without exempting such inputs, a test may queue one decision object per attention call
# (consumed in call order by deform_cross_attention_2d / _1d): the oracle then evaluates relu(x) as x * mask with the masks the
# kernels used and takes the sampler's cell from the kernels' floor() - values change by at most the rounding-level
# pre-activation, gradients follow the imposed branch.  Interface (tests/helpers.py Decisions): .cells -> (x0, y0) int64
# [(B g), J] or None; .relu_masks(i0, i1) -> (m1, m2) bool [(B g), i1 - i0, J, 32].
DECISIONS = None


def _next_decisions():
    if DECISIONS:
        return DECISIONS.pop(0)
    return None


def cpb_mlp(pos: torch.Tensor, p: Params, prefix: str = "rel_pos_bias.", masks=None) -> torch.Tensor:
    """The position-bias MLP  in -> 32 -> 32 -> heads//groups  (depth = 2).

    DeformableAttention2D.py:129-152 / DeformableAttention1D.py:69-98.
    ``pos`` is [..., in] and already signed-log transformed.  ``masks`` = (m1, m2): imposed ReLU decisions (see DECISIONS)."""
    x1 = pos @ p[prefix + "mlp.0.0.weight"].t() + p[prefix + "mlp.0.0.bias"]
    h = torch.relu(x1) if masks is None else x1 * masks[0].to(device=x1.device, dtype=x1.dtype)
    x2 = h @ p[prefix + "mlp.1.0.weight"].t() + p[prefix + "mlp.1.0.bias"]
    h = torch.relu(x2) if masks is None else x2 * masks[1].to(device=x2.device, dtype=x2.dtype)
    out = h @ p[prefix + "mlp.2.weight"].t() + p[prefix + "mlp.2.bias"]
    if GRAD_PROBE is not None and out.requires_grad:
        key = id(p[prefix + "mlp.2.bias"])
        probe = GRAD_PROBE
        probe["cpb_units"] = probe.get("cpb_units", 0) + 2 * h.numel()       # ReLU units evaluated (both hidden layers)

        def _hook(g, key=key, probe=probe):
            probe[key] = probe.get(key, 0.0) + float(g.detach().abs().sum())
        out.register_hook(_hook)
    return out


def out_len(s: int, ksize: int, r: int) -> int:
    """Output length of the strided offset conv (kernel ksize, stride r, pad (ksize-r)//2)."""
    pad = (ksize - r) // 2
    return (s + 2 * pad - ksize) // r + 1


def grouped_pointwise(x: torch.Tensor, w: torch.Tensor, groups: int) -> torch.Tensor:
    """1x1 (grouped) convolution on token-major data: x [B, n, Cin], w [Cout, Cin/groups(,1(,1))]
    -> [B, n, Cout].  Restates nn.Conv2d(dim, inner, 1, groups=g) (DeformableAttention2D.py:218-220)."""
    w = w.reshape(w.shape[0], -1)
    B, n, cin = x.shape
    cout = w.shape[0]
    xg = x.reshape(B, n, groups, cin // groups)
    wg = w.reshape(groups, cout // groups, cin // groups)
    return torch.einsum("bngc,goc->bngo", xg, wg).reshape(B, n, cout)


def sample_positions(vx: torch.Tensor, vy: torch.Tensor, W: int, H: int, cells=None):
    """Integer path of F.grid_sample(mode='bilinear', padding_mode='zeros', align_corners=False)
    as called at DeformableAttention2D.py:268-271: pixel coordinates, the four corner indices,
    their in-bounds masks and bilinear weights.

    Coordinates use ((v + 1) * size - 1) / 2 with one rounding per operation (no FMA contraction);
    the HIP kernel evaluates the same sequence so indices/masks compare bit-exactly."""
    ix = ((vx + 1.0) * float(W) - 1.0) / 2.0
    iy = ((vy + 1.0) * float(H) - 1.0) / 2.0
    if GRAD_PROBE is not None:
        # bilinear interpolation has a kink at integer pixel coordinates: where a sample position lies within fp32 rounding of
        # one, no fp32 evaluation determines which cell's slope the position gradient takes.  Tests read this margin.
        with torch.no_grad():      # only coordinates that carry a gradient matter (the 1-D module's y coordinate is the constant 0)
            d = 1.0
            if vx.requires_grad:
                d = min(d, float((ix - torch.round(ix)).abs().min()))
            if vy.requires_grad:
                d = min(d, float((iy - torch.round(iy)).abs().min()))
            GRAD_PROBE["boundary"] = min(GRAD_PROBE.get("boundary", 1.0), d)
    x0f, y0f = torch.floor(ix), torch.floor(iy)
    if cells is not None:            # imposed cells (see DECISIONS): the bilinear formula of that cell, extended past its edge by <= rounding
        x0f, y0f = cells[0].to(device=ix.device, dtype=ix.dtype), cells[1].to(device=iy.device, dtype=iy.dtype)
        with torch.no_grad():
            for c, f in ((ix, x0f), (iy, y0f)):
                assert float((c - f).min()) > -1e-3 and float((c - f).max()) < 1 + 1e-3, "imposed cell is not the sample's (or its neighbour within rounding)"
    x0, y0 = x0f.to(torch.int64), y0f.to(torch.int64)
    x1, y1 = x0 + 1, y0 + 1
    wx1, wy1 = ix - x0f, iy - y0f
    wx0, wy0 = 1.0 - wx1, 1.0 - wy1
    mx0, mx1 = (x0 >= 0) & (x0 < W), (x1 >= 0) & (x1 < W)
    my0, my1 = (y0 >= 0) & (y0 < H), (y1 >= 0) & (y1 < H)
    corners = (
        (x0, y0, wx0 * wy0, mx0 & my0),
        (x1, y0, wx1 * wy0, mx1 & my0),
        (x0, y1, wx0 * wy1, mx0 & my1),
        (x1, y1, wx1 * wy1, mx1 & my1),
    )
    return ix, iy, corners


def bilinear_gather(feats: torch.Tensor, vx: torch.Tensor, vy: torch.Tensor, cells=None) -> torch.Tensor:
    """feats [Bg, H, W, c] token-major, vx/vy [Bg, J] normalised sample positions -> [Bg, J, c]."""
    Bg, H, W, c = feats.shape
    _, _, corners = sample_positions(vx, vy, W, H, cells)
    flat = feats.reshape(Bg, H * W, c)
    out = torch.zeros(Bg, vx.shape[1], c, dtype=feats.dtype, device=feats.device)
    for (cx, cy, wgt, m) in corners:
        idx = (cy.clamp(0, H - 1) * W + cx.clamp(0, W - 1))
        g = torch.gather(flat, 1, idx.unsqueeze(-1).expand(-1, -1, c))
        out = out + g * (wgt * m.to(wgt.dtype)).unsqueeze(-1)
    return out


def _attend(q, k, v, bias_fn, heads, scale, q_chunk, attn_keep=None, keep_scale=1.0):
    """dropout(softmax(scale*q k^T + bias)) v, evaluated in query chunks; `attn_keep` [B, h, n, J] is an explicit
    0/1 keep mask (nn.Dropout semantics: kept probabilities are multiplied by keep_scale = 1 / (1 - p)).
    q [B, n, inner], k/v [B, J, inner]; bias_fn(i0, i1) -> [B, heads, i1-i0, J].
    DeformableAttention2D.py:284-312 / DeformableAttention1D.py:205-232 (dropout off)."""
    B, n, inner = q.shape
    J = k.shape[1]
    d = inner // heads
    qh = q.reshape(B, n, heads, d).permute(0, 2, 1, 3) * scale
    kh = k.reshape(B, J, heads, d).permute(0, 2, 1, 3)
    vh = v.reshape(B, J, heads, d).permute(0, 2, 1, 3)
    outs = []
    for i0 in range(0, n, q_chunk):
        i1 = min(n, i0 + q_chunk)
        sim = torch.einsum("bhid,bhjd->bhij", qh[:, :, i0:i1], kh) + bias_fn(i0, i1)
        sim = sim - sim.amax(dim=-1, keepdim=True).detach()
        attn = sim.softmax(dim=-1)
        if attn_keep is not None:
            attn = attn * (attn_keep[:, :, i0:i1].to(attn.dtype) * keep_scale)
        outs.append(torch.einsum("bhij,bhjd->bhid", attn, vh))
    out = torch.cat(outs, dim=2)                       # [B, h, n, d]
    return out.permute(0, 2, 1, 3).reshape(B, n, inner)  # b n (h d)


# --------------------------------------------------------------------------
# 2-D  (DeformableAttention2D.py:161-325)
# --------------------------------------------------------------------------
def deform_cross_attention_2d(
    x1: torch.Tensor,
    x2: torch.Tensor,
    p: Params,
    *,
    grid_hw: Tuple[int, int] = (50, 50),
    heads: int = 8,
    dim_head: int = 64,
    offset_groups: int = 8,
    downsample_factor: int = 4,
    offset_scale: float = 4.0,
    offset_kernel_size: int = 6,
    group_queries: bool = True,
    group_key_values: bool = True,
    q_chunk: int = 512,
    return_aux: bool = False,
    attn_keep: Optional[torch.Tensor] = None,
    dropout_p: float = 0.0,
    consistent_grid_norm: bool = False,
):
    """x1 (queries, fused stream) and x2 (keys/values, path stream): [B, C, N] channels-first as in
    the reference; returns (out [B, C, N], vgrid [(B g), 2, th, tw]) and optionally a dict of
    intermediates (vs, kv_feats, q, k, v, corner indices/masks)."""
    B, C, N = x1.shape
    Hh, Ww = grid_hw
    assert Hh * Ww == N, "token count must equal the grid size"
    G, r = offset_groups, downsample_factor
    inner = heads * dim_head
    dg = inner // G            # offset_dims: channels of q per offset group
    cg = C // G                # channels of x2 per offset group
    scale = dim_head ** -0.5

    x1t, x2t = x1.transpose(1, 2), x2.transpose(1, 2)                       # [B, N, C]
    q = grouped_pointwise(x1t, p["to_q.weight"], G if group_queries else 1)  # :246

    # offsets: depthwise strided conv -> GELU -> 1x1 (dg -> 2) -> tanh -> * offset_scale  (:207-213,255)
    qg = q.reshape(B, Hh, Ww, G, dg).permute(0, 3, 4, 1, 2).reshape(B * G, dg, Hh, Ww)
    pad = (offset_kernel_size - r) // 2
    y = F.conv2d(qg, p["to_offsets.0.weight"], p["to_offsets.0.bias"], stride=r, padding=pad, groups=dg)
    y = F.gelu(y)
    w2 = p["to_offsets.2.weight"].reshape(2, dg)
    offsets = torch.tanh(torch.einsum("bchw,oc->bohw", y, w2)) * offset_scale   # [(B g), 2, th, tw]
    th, tw = offsets.shape[-2:]

    # grid (xy indexing: channel 0 = column index, channel 1 = row index) + offsets   (:88-98,260-262)
    gx = torch.arange(tw, dtype=x1.dtype, device=x1.device).view(1, tw).expand(th, tw)
    gy = torch.arange(th, dtype=x1.dtype, device=x1.device).view(th, 1).expand(th, tw)
    vgrid = torch.stack((gx, gy), dim=0) + offsets
    # normalize_grid (:100-108): channel 0 is divided by (rows-1), channel 1 by (cols-1)
    vsx = 2.0 * vgrid[:, 0] / max(th - 1, 1) - 1.0
    vsy = 2.0 * vgrid[:, 1] / max(tw - 1, 1) - 1.0
    J = th * tw
    if consistent_grid_norm:      # corrected semantics (NOT the reference): pixel-centre convention, x by columns, y by rows
        vsx = (2.0 * vgrid[:, 0] + 1.0) / tw - 1.0
        vsy = (2.0 * vgrid[:, 1] + 1.0) / th - 1.0
    vsx, vsy = vsx.reshape(B * G, J), vsy.reshape(B * G, J)

    # bilinear sampling of the grouped path stream at the *full* map size (:268-274)
    feats = x2t.reshape(B, Hh, Ww, G, cg).permute(0, 3, 1, 2, 4).reshape(B * G, Hh, Ww, cg)
    dec = _next_decisions()
    kv = bilinear_gather(feats, vsx, vsy, dec.cells if dec is not None else None)   # [(B g), J, cg]
    kv = kv.reshape(B, G, J, cg).permute(0, 2, 1, 3).reshape(B, J, C)       # b j (g c)

    k = grouped_pointwise(kv, p["to_k.weight"], G if group_key_values else 1)   # :279
    v = grouped_pointwise(kv, p["to_v.weight"], G if group_key_values else 1)

    # continuous position bias (:120-157,296-299); query grid normalised by the full map size
    qx = 2.0 * torch.arange(Ww, dtype=x1.dtype, device=x1.device) / max(Hh - 1, 1) - 1.0
    qy = 2.0 * torch.arange(Hh, dtype=x1.dtype, device=x1.device) / max(Ww - 1, 1) - 1.0
    if consistent_grid_norm:
        qx = (2.0 * torch.arange(Ww, dtype=x1.dtype, device=x1.device) + 1.0) / Ww - 1.0
        qy = (2.0 * torch.arange(Hh, dtype=x1.dtype, device=x1.device) + 1.0) / Hh - 1.0
    gq = torch.stack((qx.view(1, Ww).expand(Hh, Ww), qy.view(Hh, 1).expand(Hh, Ww)), dim=-1).reshape(N, 2)
    vs = torch.stack((vsx, vsy), dim=-1)                                    # [(B g), J, 2]
    o = heads // G

    def bias_fn(i0, i1):
        pos = gq[i0:i1].view(1, i1 - i0, 1, 2) - vs.view(B * G, 1, J, 2)
        b = cpb_mlp(signed_log(pos), p, masks=dec.relu_masks(i0, i1) if dec is not None else None)   # [(B g), i, J, o]
        return b.reshape(B, G, i1 - i0, J, o).permute(0, 1, 4, 2, 3).reshape(B, heads, i1 - i0, J)

    attn_out = _attend(q, k, v, bias_fn, heads, scale, q_chunk, attn_keep, 1.0 / (1.0 - dropout_p))   # [B, N, inner]
    wo = p["to_out.weight"].reshape(C, inner)
    out = attn_out @ wo.t() + p["to_out.bias"]                              # :313
    out = out.transpose(1, 2)                                               # [B, C, N]
    if not return_aux:
        return out, vgrid
    _, _, corners = sample_positions(vsx, vsy, Ww, Hh)
    aux = dict(q=q, k=k, v=v, kv=kv, vsx=vsx, vsy=vsy, attn_out=attn_out,
               corner_x=torch.stack([c[0] for c in corners], -1),
               corner_y=torch.stack([c[1] for c in corners], -1),
               corner_mask=torch.stack([c[3] for c in corners], -1))
    return out, vgrid, aux


# --------------------------------------------------------------------------
# 1-D  (DeformableAttention1D.py:106-240)
# --------------------------------------------------------------------------
def deform_cross_attention_1d(
    x1: torch.Tensor,
    x2: torch.Tensor,
    p: Params,
    *,
    heads: int = 8,
    dim_head: int = 64,
    offset_groups: int = 4,
    downsample_factor: int = 4,
    offset_scale: Optional[float] = None,
    offset_kernel_size: int = 6,
    cpb_log_distance: bool = True,
    group_queries: bool = False,
    group_key_values: bool = False,
    q_chunk: int = 512,
    return_aux: bool = False,
    true_1d_sampling: bool = False,
):
    """x1, x2 [B, C, n] -> (out [B, C, n], vgrid [(B g), t]).

    Bug-compatible with the reference's ``grid_sample_1d`` (:36-43): the sampling grid is padded as
    (x = vs, y = 0) while the features are laid out [H = n, W = 1], so ``vs`` indexes the size-1
    axis and the row coordinate is the constant centre (n - 1) / 2."""
    B, C, n = x1.shape
    G, r = offset_groups, downsample_factor
    offset_scale = float(r if offset_scale is None else offset_scale)
    inner = heads * dim_head
    dg, cg = inner // G, C // G
    scale = dim_head ** -0.5

    x1t, x2t = x1.transpose(1, 2), x2.transpose(1, 2)
    q = grouped_pointwise(x1t, p["to_q.weight"], G if group_queries else 1)     # :169

    qg = q.reshape(B, n, G, dg).permute(0, 2, 3, 1).reshape(B * G, dg, n)
    pad = (offset_kernel_size - r) // 2
    y = F.gelu(F.conv1d(qg, p["to_offsets.0.weight"], p["to_offsets.0.bias"], stride=r, padding=pad, groups=dg))
    w2 = p["to_offsets.2.weight"].reshape(1, dg)
    offsets = torch.tanh(torch.einsum("bcn,oc->bon", y, w2)[:, 0]) * offset_scale   # [(B g), t]
    t = offsets.shape[-1]
    vgrid = torch.arange(t, dtype=x1.dtype, device=x1.device) + offsets             # :180-182
    vs = 2.0 * vgrid / max(t - 1, 1) - 1.0                                          # :45-48,183

    feats = x2t.reshape(B, n, G, cg).permute(0, 2, 1, 3).reshape(B * G, n, 1, cg)   # H = n, W = 1
    if true_1d_sampling:          # corrected semantics (NOT the reference): H = 1, W = n - vs runs along the tokens
        feats = feats.reshape(B * G, 1, n, cg)
    dec = _next_decisions()
    kv = bilinear_gather(feats, vs, torch.zeros_like(vs), dec.cells if dec is not None else None)   # [(B g), t, cg]
    kv = kv.reshape(B, G, t, cg).permute(0, 2, 1, 3).reshape(B, t, C)

    k = grouped_pointwise(kv, p["to_k.weight"], G if group_key_values else 1)
    v = grouped_pointwise(kv, p["to_v.weight"], G if group_key_values else 1)

    seq = 2.0 * torch.arange(n, dtype=x1.dtype, device=x1.device) / max(n - 1, 1) - 1.0   # :213-214
    o = heads // G

    def bias_fn(i0, i1):
        pos = seq[i0:i1].view(1, i1 - i0, 1, 1) - vs.view(B * G, 1, t, 1)
        if cpb_log_distance:
            pos = signed_log(pos)
        b = cpb_mlp(pos, p, masks=dec.relu_masks(i0, i1) if dec is not None else None)   # [(B g), i, t, o]
        return b.reshape(B, G, i1 - i0, t, o).permute(0, 1, 4, 2, 3).reshape(B, heads, i1 - i0, t)

    attn_out = _attend(q, k, v, bias_fn, heads, scale, q_chunk)
    wo = p["to_out.weight"].reshape(C, inner)
    out = (attn_out @ wo.t() + p["to_out.bias"]).transpose(1, 2)
    if not return_aux:
        return out, vgrid
    return out, vgrid, dict(q=q, k=k, v=v, kv=kv, vs=vs, attn_out=attn_out)

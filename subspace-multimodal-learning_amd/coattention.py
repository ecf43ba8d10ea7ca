"""Host-side mirror of the reference's co-attention: models/MultiheadAttention.py:7-321 (functional form) and
:333-489 (module) - a fork of torch.nn.MultiheadAttention whose forward returns the RAW pre-softmax scores
(`need_raw`, :299,308-312).  Used by MCAT_Surv.coattn (models/model.py:587,626-628) and CMTA's P_in_G_Att /
G_in_P_Att (models/model.py:748-750,809-818), always with embed_dim 256, one head, no masks.

Same constructor, forward signature, return tuple and parameter names (in_proj_weight, in_proj_bias,
out_proj.weight/bias).  In-projections, Q K^T, softmax, P V and the out-projection run on the HIP kernels.
Every constructor option of the reference's class runs - no caller in
the reference uses them.  The reference's `torch.equal(query, key)` host sync (:126,130) only selects between
algebraically identical in-projection paths and is not reproduced."""
from __future__ import annotations

import torch
import torch.nn.functional as F
from torch import nn
from torch.nn.init import constant_, xavier_uniform_

from . import functional as Fh


class MultiheadAttention(nn.Module):
    def __init__(self, embed_dim, num_heads, dropout=0., bias=True, add_bias_kv=False, add_zero_attn=False, kdim=None,
                 vdim=None):
        super().__init__()
        self.embed_dim = embed_dim
        self.kdim = kdim if kdim is not None else embed_dim
        self.vdim = vdim if vdim is not None else embed_dim
        self._qkv_same_embed_dim = self.kdim == embed_dim and self.vdim == embed_dim
        self.num_heads = num_heads
        self.dropout = dropout
        self.head_dim = embed_dim // num_heads
        assert self.head_dim * num_heads == self.embed_dim, "embed_dim must be divisible by num_heads"
        if self._qkv_same_embed_dim:
            self.in_proj_weight = nn.Parameter(torch.empty(3 * embed_dim, embed_dim))
            self.register_parameter('q_proj_weight', None)
            self.register_parameter('k_proj_weight', None)
            self.register_parameter('v_proj_weight', None)
        else:                                 # MultiheadAttention.py:372-379 (no caller in the reference asks for it): separate projections
            self.q_proj_weight = nn.Parameter(torch.empty(embed_dim, embed_dim))
            self.k_proj_weight = nn.Parameter(torch.empty(embed_dim, self.kdim))
            self.v_proj_weight = nn.Parameter(torch.empty(embed_dim, self.vdim))
            self.register_parameter('in_proj_weight', None)
        if bias:
            self.in_proj_bias = nn.Parameter(torch.empty(3 * embed_dim))
        else:
            self.register_parameter('in_proj_bias', None)
        self.out_proj = nn.modules.linear.NonDynamicallyQuantizableLinear(embed_dim, embed_dim)
        if add_bias_kv:                       # MultiheadAttention.py:393-397 (no caller in the reference asks for it)
            self.bias_k = nn.Parameter(torch.empty(1, 1, embed_dim))
            self.bias_v = nn.Parameter(torch.empty(1, 1, embed_dim))
        else:
            self.bias_k = self.bias_v = None
        self.add_zero_attn = add_zero_attn
        self._reset_parameters()

    def _reset_parameters(self):
        if self._qkv_same_embed_dim:
            xavier_uniform_(self.in_proj_weight)
        else:
            xavier_uniform_(self.q_proj_weight); xavier_uniform_(self.k_proj_weight); xavier_uniform_(self.v_proj_weight)
        if self.in_proj_bias is not None:
            constant_(self.in_proj_bias, 0.)
            constant_(self.out_proj.bias, 0.)
        if self.bias_k is not None:
            nn.init.xavier_normal_(self.bias_k)
            nn.init.xavier_normal_(self.bias_v)

    def forward(self, query, key, value, key_padding_mask=None, need_weights=True, need_raw=True, attn_mask=None):
        """query [L, B, E], key / value [S, B, E] (sequence first) -> (out [L, B, E], raw scores [B, h, L, S])."""
        L, B, E = query.shape
        S = key.shape[0]
        h, hd = self.num_heads, self.head_dim
        scaling = float(hd) ** -0.5
        if self._qkv_same_embed_dim:
            wq, wk, wv = self.in_proj_weight.chunk(3, dim=0)
        else:
            wq, wk, wv = self.q_proj_weight, self.k_proj_weight, self.v_proj_weight
        bq = bk = bv = None
        if self.in_proj_bias is not None:
            bq, bk, bv = self.in_proj_bias.chunk(3, dim=0)

        def heads_first(t, n):          # [B, n, E] -> [B, h, n, hd]
            return t.reshape(B, 1, n, E) if h == 1 else t.reshape(B, n, h, hd).permute(0, 2, 1, 3)

        q = heads_first(Fh.linear(query.transpose(0, 1), wq, bq), L)
        kp, vp = Fh.linear(key.transpose(0, 1), wk, bk), Fh.linear(value.transpose(0, 1), wv, bv)      # [B, S, E]
        pad_cols = 0
        if self.bias_k is not None:           # one learned key / value row appended (:236-243); masks grow by an unmasked column
            kp = torch.cat((kp, self.bias_k.expand(B, 1, E)), dim=1)
            vp = torch.cat((vp, self.bias_v.expand(B, 1, E)), dim=1)
            S += 1; pad_cols += 1
        k, v = heads_first(kp, S), heads_first(vp, S)
        if self.add_zero_attn:                # a zero key / value row per head (:271-279)
            zrow = torch.zeros(B, h, 1, hd, dtype=k.dtype, device=k.device)
            k, v = torch.cat((k, zrow), dim=2), torch.cat((v, zrow), dim=2)
            S += 1; pad_cols += 1
        if pad_cols:
            if attn_mask is not None:
                attn_mask = F.pad(attn_mask, (0, pad_cols))
            if key_padding_mask is not None:
                key_padding_mask = F.pad(key_padding_mask, (0, pad_cols))
        raw = Fh.matmul4(q, k, tb=True, alpha=scaling)                 # (q * scaling) k^T, :284
        # masks (MultiheadAttention.py:206-227,284-296; no caller in the reference passes one): a bool attn_mask / the key_padding_mask fill with -inf, a
        # float attn_mask is added; the returned raw scores are the MASKED ones (:298).  [B, h, L, S]-sized elementwise work, only on this branch
        if attn_mask is not None:
            am = attn_mask.to(torch.bool) if attn_mask.dtype == torch.uint8 else attn_mask
            if am.dim() == 2:
                if tuple(am.shape) != (L, S):
                    raise RuntimeError("The size of the 2D attn_mask is not correct.")
                am = am.view(1, 1, L, S)
            elif am.dim() == 3:
                if tuple(am.shape) != (B * h, L, S):
                    raise RuntimeError("The size of the 3D attn_mask is not correct.")
                am = am.view(B, h, L, S)
            else:
                raise RuntimeError(f"attn_mask's dimension {am.dim()} is not supported")
            raw = raw.masked_fill(am, float("-inf")) if am.dtype == torch.bool else raw + am.to(raw.dtype)
        if key_padding_mask is not None:
            kpm = key_padding_mask.to(torch.bool)
            if tuple(kpm.shape) != (B, S):
                raise RuntimeError("key_padding_mask must be [batch, source length]")
            raw = raw.masked_fill(kpm.view(B, 1, 1, S), float("-inf"))
        attn = Fh.softmax_rows(raw)
        if self.training and self.dropout > 0:
            attn = F.dropout(attn, p=self.dropout, training=True)
        o = Fh.matmul4(attn, v, merged=True)                           # [B, L, h*hd]
        out = Fh.linear(o, self.out_proj.weight, self.out_proj.bias).transpose(0, 1)
        if need_weights:
            if need_raw:
                return out, raw
            return out, attn.sum(dim=1) / h
        return out, None

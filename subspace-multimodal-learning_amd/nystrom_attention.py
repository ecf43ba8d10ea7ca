"""Host-side mirror of the Nystrom landmark self-attention block on the HIP kernels.

  NystromAttention  models/NystromAttention.py:39-157 (dup models/cmta_utils.py:166-281; the pip package
                    `nystrom_attention` imported at models/mil.py:24 states the same algorithm)
  TransLayer        models/mil.py:171-189   (dup cmta_utils.py:858-874)
  PPEG              models/mil.py:192-206   (dup cmta_utils.py:877-891)
  TransMIL          models/mil.py:209-259

Same constructors, forward signatures and parameter names.  Every contraction (qkv projection, the three
similarity products, the six Newton-Schulz iterations of the pseudo-inverse written as alpha/beta GEMM
epilogues, (attn1 z)(attn3 v), output projection) runs on the matrix cores through smml_gemm_f32; softmax,
landmark means, the 33-tap residual convolution and PPEG's merged 7x7 depthwise pass are HBM-bound kernels.
Kept as in the reference: zero padding in FRONT of the sequence (:82), the batch-global max in the
pseudo-inverse initialisation (:26).  The `mask` argument (no caller in the reference passes it) runs on the exact fp32 path only.

16-bit compute mode (BASELINE configs 2 / 4 / 5 quote bf16 / fp16 bags): `compute_dtype` 'bf16' / 'fp16', or a bag that
arrives in that dtype, routes the n'-sized contractions to the 16-bit matrix pipe with fp32 storage and accumulation:
  * softmax(q kl^T) and softmax(ql k^T) v become two calls of the fused attention kernel (csrc/attn16.hip): the [n', m] and
    [m, n'] probability matrices are never written, out = attn1 (z (attn3 v)) replaces (attn1 z)(attn3 v) (same value, the
    [n', m] x [m, m] product becomes [m, m] x [m, d]);
  * the qkv / output projections run with single-term 16-bit operands (bf16 mode: bf16 everywhere; fp16 mode: fp16 in the forward
    product, bf16 in the two gradient products - gradients have no place in fp16's range without a loss scale);
  * the m x m part - sim2 and its softmax - stays exact fp32; the Newton-Schulz iteration runs on two bf16 planes per matrix (16 operand
    mantissa bits, z to 2e-5 of an fp64 evaluation: ~25 x finer than the fp16 attention products around it).
Accuracy of that mode: bf16 ~ 1e-2, fp16 ~ 2e-3 of a tensor's scale (8 / 11 mantissa bits on the operands); the default
fp32 path is unchanged and is what the 1e-4 parity tests cover."""
from __future__ import annotations

import math
import os

import numpy as np
import torch
import torch.nn.functional as F
from torch import nn

from . import _capi as capi
from . import functional as Fh


class _NewtonSchulz(torch.autograd.Function):
    """z_{k+1} = 1/4 z_k (13 I - x z_k (15 I - x z_k (7 I - x z_k))), `iters` times, as four batched GEMMs per iteration with the
    affine parts in the GEMM epilogues (xz; a = 7 xz - xz xz; b = 15 xz - xz a; z' = 3.25 z - 0.25 z b), and a hand-written
    backward of eight GEMMs + one elementwise update per iteration:
        dz = 3.25 dz' - 0.25 dz' b^T        db = -0.25 z^T dz'
        dxz = 15 db - db a^T                da = -xz^T db
        dxz += 7 da - da xz^T - xz^T da
        dx += dxz z^T                       dz += x^T dxz
    Each direction is ONE call into the C-ABI (csrc/pinv_chain.hip), which issues the 24 / 54 launches from a C loop: issued one by
    one from here they cost more host time (~25 us each) than GPU time (~17 us)."""

    @staticmethod
    def forward(ctx, x, z0, iters, reduced=False):
        x, z0 = Fh._c(x), Fh._c(z0)
        red = int(bool(reduced))
        m = x.shape[-1]
        nb = x.numel() // (m * m)
        L = capi.lib()
        saved = torch.empty(L.smml_newton_schulz_saved_floats(nb, m, iters, red), device=x.device, dtype=torch.float32)
        z = torch.empty_like(x)
        capi.check(L.smml_newton_schulz_fwd(capi.fptr(x), capi.fptr(z0), capi.fptr(saved), capi.fptr(z), nb, m, iters, red,
                                      capi.stream()), "newton_schulz_fwd")
        ctx.iters, ctx.red = iters, red
        ctx.save_for_backward(x, z0, saved)
        return z

    @staticmethod
    def backward(ctx, dz):
        x, z0, saved = ctx.saved_tensors
        dz = Fh._c(dz)
        m = x.shape[-1]
        nb = x.numel() // (m * m)
        dx, dz0 = torch.empty_like(x), torch.empty_like(x)
        L = capi.lib()
        scratch = torch.empty(L.smml_newton_schulz_scratch_floats(nb, m, ctx.iters, ctx.red), device=x.device, dtype=torch.float32)
        capi.check(L.smml_newton_schulz_bwd(capi.fptr(x), capi.fptr(z0), capi.fptr(saved), capi.fptr(dz), capi.fptr(dx),
                                                     capi.fptr(dz0), capi.fptr(scratch), nb, m, ctx.iters, ctx.red, capi.stream()),
                   "newton_schulz_bwd")
        return dx, dz0, None, None


def moore_penrose_iter_pinv(x, iters=6, per_bag=False, reduced=False):
    """x [B, h, m, m] -> Newton-Schulz pseudo-inverse (NystromAttention.py:20-35).
    reduced=True (the block's 16-bit compute mode): products with 16-bit operand mantissas are acceptable (csrc/pinv_chain.hip).
    z <- 1/4 z (13 I - xz (15 I - xz (7 I - xz))) evaluated as four GEMMs per iteration (see _NewtonSchulz).
    per_bag=True (corrected semantics, off by default): the initial scale uses each bag's own max row / column sums instead
    of the max over the whole batch (:26), so that a bag's result does not depend on which other bags share its batch."""
    ax = x.abs()
    if per_bag:
        scale = ax.sum(dim=-1).amax(dim=(1, 2)) * ax.sum(dim=-2).amax(dim=(1, 2))
        z = (x.transpose(-1, -2) / scale.view(-1, 1, 1, 1)).contiguous()
    else:
        z = (x.transpose(-1, -2) / (ax.sum(dim=-1).max() * ax.sum(dim=-2).max())).contiguous()
    if iters <= 0:
        return z
    return _NewtonSchulz.apply(x.contiguous(), z, iters, reduced)


_SIDE_STREAMS = {}
PINV_OVERLAP = os.environ.get("SMML_NYSTROM_OVERLAP", "1") != "0"     # measurement switch


class _PinvFork:
    """The pseudo-inverse is a chain of 24 (backward: 54) dependent 256^3 batched products, each too small to fill the chip (512
    workgroups, ~17 us), and nothing else in the block needs its result before the last product: it runs on a second HIP stream
    beside the n'-sized kernels (softmax(ql k^T) v, the residual convolution, their backward passes - autograd replays every
    node on the stream of its forward).  fork() makes the side stream wait for what the main stream has produced so far; join()
    makes the main stream wait for the side stream and tells the caching allocator that the tensors are now in use there."""

    def __init__(self, like: torch.Tensor):
        self.on = bool(PINV_OVERLAP and like.is_cuda)
        if not self.on:
            return
        dev = like.device
        self.main = torch.cuda.current_stream(dev)
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        self.side = _SIDE_STREAMS.get(key)
        if self.side is None:
            self.side = _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
        self.side.wait_stream(self.main)

    def run(self, fn, *inputs):
        if not self.on:
            return fn(*inputs)
        for t in inputs:
            if torch.is_tensor(t):
                t.record_stream(self.side)
        with torch.cuda.stream(self.side):
            return fn(*inputs)

    def join(self, *outputs):
        if self.on:
            self.main.wait_stream(self.side)
            for t in outputs:
                t.record_stream(self.main)


# measurement switches of the bf16 compute mode: SMML_NYSTROM_B16 = 2 (default) bf16 storage end to end, 1 = bf16-storage projections around
# the fp32-storage attention kernels (round-3 intermediate), 0 = fp32 storage everywhere (round 2)
_B16_LEVEL = int(os.environ.get("SMML_NYSTROM_B16", "2"))
B16_PROJECTIONS = _B16_LEVEL >= 1
B16_STORAGE = _B16_LEVEL >= 2


class NystromAttention(nn.Module):
    def __init__(self, dim, dim_head=64, heads=8, num_landmarks=256, pinv_iterations=6, residual=True,
                 residual_conv_kernel=33, eps=1e-8, dropout=0., per_bag_pinv_scale: bool = False, compute_dtype=None):
        super().__init__()
        self.eps = eps
        if compute_dtype not in (None, "fp32", "bf16", "fp16"):
            raise ValueError("compute_dtype must be None (follow the bag's dtype), 'fp32', 'bf16' or 'fp16'")
        self.compute_dtype = compute_dtype
        self.per_bag_pinv_scale = bool(per_bag_pinv_scale)     # corrected semantics (off by default), see moore_penrose_iter_pinv
        inner_dim = heads * dim_head
        self.num_landmarks = num_landmarks
        self.pinv_iterations = pinv_iterations
        self.heads = heads
        self.dim_head = dim_head
        self.scale = dim_head ** -0.5
        self.to_qkv = nn.Linear(dim, inner_dim * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, dim), nn.Dropout(dropout))
        self.residual = residual
        if residual:
            kernel_size = residual_conv_kernel
            padding = residual_conv_kernel // 2
            self.res_conv = nn.Conv2d(heads, heads, (kernel_size, 1), padding=(padding, 0), groups=heads, bias=False)

    def matrix_pipe(self, dtype, return_attn=False) -> str:
        """Which matrix pipe forward() issues its contractions on for a bag of `dtype`: "f32" (exact path), "bf16" or "f16" (the
        16-bit compute mode: head dim 64, no attention matrix requested).  A pure function of the module's configuration - callers
        that price a measurement (bench.py) ask here rather than reading state a forward left behind."""
        mode = self.compute_dtype or {torch.bfloat16: "bf16", torch.float16: "fp16"}.get(dtype, "fp32")
        if mode == "fp32" or self.dim_head != 64 or return_attn:
            return "f32"
        return "f16" if mode == "fp16" else "bf16"

    def forward(self, x, mask=None, return_attn=False):
        b, n, dim = x.shape
        h, m, d = self.heads, self.num_landmarks, self.dim_head
        pipe = self.matrix_pipe(x.dtype, return_attn)
        if mask is not None:
            pipe = "f32"                                       # the masked form (no caller in the reference passes one) exists on the exact path only
            x = x.float()
        if pipe != "f32":
            return self._forward16(x, pipe == "f16")
        pad = (m - n % m) % m
        if pad:
            x = F.pad(x, (0, 0, pad, 0), value=0)              # zero rows in FRONT (:82)
            if mask is not None:
                mask = F.pad(mask, (pad, 0), value=False)      # (:84)
        npad = n + pad
        l = math.ceil(n / m)
        # one projection GEMM for q, k and v (M = b n', N = 3 inner), then one strided copy into the head-major layout the
        # batched products read; the softmax scale (:98) rides on the three similarity products as alpha
        qkv = Fh.linear(x, self.to_qkv.weight)                                   # [b, n', 3 h d]
        q, k, v = Fh.head_major_qkv(qkv, h)                                      # each [b, h, n', d]: one strided copy, one gradient buffer
        fill = lambda t, keep: t
        if mask is not None:
            # NystromAttention.py:92-96,106-118,127-133: masked tokens are zeroed in q / k / v, landmarks are means over the UNMASKED tokens of a
            # segment, and every similarity between a masked token / an all-masked landmark and anything else is set to -finfo.max before the
            # softmax.  Elementwise work on n'-sized tensors, on this branch only
            mk = mask.to(torch.bool).view(b, 1, npad)
            q, k, v = (t * mk[..., None].to(t.dtype) for t in (q, k, v))
            msum = mk.view(b, 1, npad // l, l).sum(dim=-1)                      # [b, 1, m] unmasked tokens per segment
            mland = msum > 0
            lscale = (float(l) / (msum.to(q.dtype) + self.eps))[..., None]     # segment_mean divides by l; the masked mean by (count + eps)
            neg = -torch.finfo(q.dtype).max
            fill = lambda t, keep: t.masked_fill(~keep, neg)
        ql, kl = Fh.segment_mean(q, l), Fh.segment_mean(k, l)  # landmarks (:102-118); ql unscaled
        if mask is not None:
            ql, kl = ql * lscale, kl * lscale
        sc = self.scale
        a2 = Fh.softmax_rows(fill(Fh.matmul4(ql, kl, tb=True, alpha=sc), (mland[..., None] & mland[..., None, :]) if mask is not None else None))   # [b, h, m, m]
        fork = _PinvFork(a2)                                   # the pseudo-inverse runs beside the n'-sized products below
        z = fork.run(lambda t: moore_penrose_iter_pinv(t, self.pinv_iterations, self.per_bag_pinv_scale), a2)
        a1 = Fh.softmax_rows(fill(Fh.matmul4(q, kl, tb=True, alpha=sc), (mk[..., None] & mland[..., None, :]) if mask is not None else None))       # [b, h, n', m]
        a3 = Fh.softmax_rows(fill(Fh.matmul4(ql, k, tb=True, alpha=sc), (mland[..., None] & mk[..., None, :]) if mask is not None else None))       # [b, h, m, n']
        right = Fh.matmul4(a3, v)                              # [b, h, m, d]
        res = Fh.resconv(v, self.res_conv.weight) if self.residual else None
        fork.join(z)
        left = Fh.matmul4(a1, z)                               # [b, h, n', m]
        out = Fh.matmul4(left, right, res, merged=True)        # [b, n', h*d]  (:140,144-146)
        out = Fh.linear(out, self.to_out[0].weight, self.to_out[0].bias)
        out = self.to_out[1](out)
        out = out[:, -n:]
        if return_attn:
            attn = Fh.matmul4(left, a3)
            return out, attn
        return out


    def _forward16(self, x, fp16: bool):
        """16-bit compute mode (module docstring): same op order as forward() up to the re-association of the output product."""
        b, n, dim = x.shape
        h, m, d = self.heads, self.num_landmarks, self.dim_head
        pad = (m - n % m) % m
        npad = n + pad
        l = math.ceil(n / m)
        sc = self.scale
        wo, bo = self.to_out[0].weight, self.to_out[0].bias
        if not fp16 and B16_STORAGE and (not self.residual or self.res_conv.weight.shape[2] == 33):
            # bf16 storage end to end (functional.py, "bf16-storage pipeline"): q / k / v live in the projection's token-major bf16 buffer
            qkv, ql, kl = Fh.qkv_project16(x.to(torch.bfloat16), self.to_qkv.weight, h, l, pad)   # the front padding (:82) lives in the row map
            a2 = Fh.softmax_rows(Fh.matmul4(ql, kl, tb=True, alpha=sc))                         # [b, h, m, m], exact fp32
            fork = _PinvFork(a2)                                                                # beside attn3 v and the residual convolution
            z = fork.run(lambda t: moore_penrose_iter_pinv(t, self.pinv_iterations, self.per_bag_pinv_scale, reduced=True), a2)
            right, qkv = Fh.attention16_keys_long(ql, qkv, heads=h, scale=sc, dv_accumulate=self.residual)   # softmax(ql k^T) v
            res = None
            if self.residual:
                res, qkv = Fh.resconv16(qkv, self.res_conv.weight, heads=h)                     # bf16 [b, n', h d]
            fork.join(z)
            w = Fh.matmul4(z, right)                                                            # z (attn3 v)          [b, h, m, d]
            out = Fh.attention16_queries_long(qkv, kl, w, res, heads=h, scale=sc)               # softmax(q kl^T) w + res, bf16
            return self.to_out[1](Fh.linear_b16(out, wo, bo, out_bf16=False, skip=pad))         # [b, n, dim]: the padded rows are not projected
        b16 = not fp16 and B16_PROJECTIONS                      # bf16 mode: the projections read and write bf16 (csrc/gemm_b16.hip)
        if b16:
            x = x.to(torch.bfloat16)
            if pad:
                x = F.pad(x, (0, 0, pad, 0), value=0)
            q, k, v = Fh.head_major_qkv(Fh.linear_b16(x, self.to_qkv.weight), h)               # each [b, h, n', d] fp32
        else:
            x = x.float()
            if pad:
                x = F.pad(x, (0, 0, pad, 0), value=0)
            # projections on the 16-bit pipe with fp32 storage: single-term bf16 operands (bf16 mode, csrc/gemm.hip mode 3) or single-term
            # fp16 operands in the forward product and bf16 ones in the two gradient products (fp16 mode, mode 4: fp16 has no range for
            # gradients without a loss scale).  Round 3 kept the fp16 mode's projections exact: 3 ms of its 5.2 ms step
            gm = 4 if fp16 else 3
            qkv = Fh.linear(x, self.to_qkv.weight, prec=gm)
            q, k, v = Fh.head_major_qkv(qkv, h)                                                    # each [b, h, n', d]
        ql, kl = Fh.segment_mean(q, l), Fh.segment_mean(k, l)
        a2 = Fh.softmax_rows(Fh.matmul4(ql, kl, tb=True, alpha=sc))                            # [b, h, m, m], exact fp32
        fork = _PinvFork(a2)                                                                   # beside attn3 v and the residual convolution
        # the two-plane 16-bit form of the Newton-Schulz chain (16 operand mantissa bits: z to 2e-5) in both 16-bit modes
        z = fork.run(lambda t: moore_penrose_iter_pinv(t, self.pinv_iterations, self.per_bag_pinv_scale, reduced=True), a2)
        right = Fh.attention16(ql, k, v, scale=sc, fp16=fp16)                                  # softmax(ql k^T) v   [b, h, m, d]
        res = Fh.resconv(v, self.res_conv.weight) if self.residual else None                   # [b, n', h d]
        fork.join(z)
        w = Fh.matmul4(z, right)                                                               # z (attn3 v)          [b, h, m, d]
        out = Fh.attention16(q, kl, w, scale=sc, fp16=fp16, merged=True, residual=res)         # softmax(q kl^T) w + res
        out = Fh.linear_b16(out.to(torch.bfloat16), wo, bo, out_bf16=False) if b16 else Fh.linear(out, wo, bo, prec=gm)
        out = self.to_out[1](out)
        return out[:, -n:]


class TransLayer(nn.Module):
    """models/mil.py:168-178 (dup cmta_utils.py).  compute_dtype (extension, default None = follow the input's dtype, i.e. exact fp32 behind
    the fp32 LayerNorm): 'bf16' / 'fp16' runs the block's contractions in its 16-bit compute mode (NystromAttention docstring)."""

    def __init__(self, norm_layer=nn.LayerNorm, dim=512, compute_dtype=None):
        super().__init__()
        self.norm = norm_layer(dim)
        self.attn = NystromAttention(dim=dim, dim_head=dim // 8, heads=8, num_landmarks=dim // 2, pinv_iterations=6,
                                     residual=True, dropout=0.1, compute_dtype=compute_dtype)

    def forward(self, x):
        return x + self.attn(Fh.layer_norm(x, self.norm.weight, self.norm.bias, self.norm.eps))


class PPEG(nn.Module):
    def __init__(self, dim=512):
        super().__init__()
        self.proj = nn.Conv2d(dim, dim, 7, 1, 7 // 2, groups=dim)
        self.proj1 = nn.Conv2d(dim, dim, 5, 1, 5 // 2, groups=dim)
        self.proj2 = nn.Conv2d(dim, dim, 3, 1, 3 // 2, groups=dim)

    def merged_kernel(self):
        """7x7 + zero-padded 5x5 + zero-padded 3x3 + identity -> one [C, 49] depthwise kernel and one bias."""
        w = self.proj.weight + F.pad(self.proj1.weight, (1, 1, 1, 1)) + F.pad(self.proj2.weight, (2, 2, 2, 2))
        C = w.shape[0]
        ident = torch.zeros(1, 1, 7, 7, device=w.device, dtype=w.dtype)
        ident[0, 0, 3, 3] = 1.0
        return (w + ident).reshape(C, 49), self.proj.bias + self.proj1.bias + self.proj2.bias

    def forward(self, x, H, W):
        B, _, C = x.shape
        cls_token, feat_token = x[:, :1], x[:, 1:]
        wm, bias = self.merged_kernel()
        y = Fh.dwconv7(feat_token.reshape(B, H, W, C), wm, bias)     # token-major = channel-last map
        return torch.cat((cls_token, y.reshape(B, H * W, C)), dim=1)


class TransMIL(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.pos_layer = PPEG(dim=512)
        self._fc1 = nn.Sequential(nn.Linear(int(getattr(args, "input_path_dim", 1024) or 1024), 512), nn.ReLU())
        self.cls_token = nn.Parameter(torch.randn(1, 1, 512))
        self.n_classes = self.args.label_dim
        cd = getattr(args, "nystrom_compute_dtype", None)            # extension key: None (exact fp32, the reference's arithmetic) | 'bf16' | 'fp16'
        self.layer1 = TransLayer(dim=512, compute_dtype=cd)
        self.layer2 = TransLayer(dim=512, compute_dtype=cd)
        self.norm = nn.LayerNorm(512)
        self._fc2 = nn.Linear(512, self.n_classes)
        self.multimodal_projection = nn.Linear(512, self.args.path_dim)

    def forward(self, x):
        h = Fh.linear(x.float(), self._fc1[0].weight, self._fc1[0].bias, act=Fh.ACT_RELU)
        Hn = h.shape[1]
        _H = _W = int(np.ceil(np.sqrt(Hn)))
        add_length = _H * _W - Hn
        h = torch.cat([h, h[:, :add_length, :]], dim=1)                 # wrap-pad to a square (:232-235)
        B = h.shape[0]
        h = torch.cat((self.cls_token.expand(B, -1, -1).to(h.device), h), dim=1)
        h = self.layer1(h)
        h = self.pos_layer(h, _H, _W)
        h = self.layer2(h)
        h = Fh.layer_norm(h[:, :1], self.norm.weight, self.norm.bias, self.norm.eps)[:, 0]
        logits = Fh.linear(h, self._fc2.weight, self._fc2.bias)
        encoded = Fh.linear(h, self.multimodal_projection.weight, self.multimodal_projection.bias)
        return encoded, logits, None

"""Autograd bindings of the HIP kernels (C-ABI in include/smml.h, loaded by _capi.py).

Each op is a torch.autograd.Function whose forward and backward launch hand-written gfx950 kernels on
the current HIP stream; PyTorch only provides device memory, the stream and the autograd graph.  All
tensors are fp32 and token-major (channel-last).  There is no CPU / eager fallback."""
from __future__ import annotations

import math
import os
from typing import Optional

import ctypes as C_

import torch

from . import _capi as capi

ACT_NONE, ACT_RELU, ACT_TANH = 0, 1, 2


class KernelTimer:
    """Optional HIP-event timing of the two dominant kernels (fused attention forward, position-bias
    backward) on the stream they are launched on; used by bench.py's roofline leg.  Off by default."""

    def __init__(self):
        self.enabled = False
        self.pairs = {"deform_attn_fwd": [], "cpb_bwd": []}
        self.work = {"deform_attn_fwd": [], "cpb_bwd": []}     # (query, key) pairs per launch

    def events(self, name, pairs_count):
        if not self.enabled:
            return None, None
        L = capi.lib()
        a, b = L.smml_event_create(), L.smml_event_create()
        self.pairs.setdefault(name, []).append((a, b))
        self.work.setdefault(name, []).append(pairs_count)
        return a, b

    def collect(self):
        """-> {name: (launches, mean ms, mean pairs per launch)}; destroys the events."""
        import ctypes
        L = capi.lib()
        out = {}
        for name, evs in self.pairs.items():
            tot = 0.0
            for a, b in evs:
                ms = ctypes.c_float(0)
                capi.check(L.smml_event_elapsed_ms(a, b, ctypes.byref(ms)), "event_elapsed")
                tot += ms.value
                L.smml_event_destroy(a); L.smml_event_destroy(b)
            if evs:
                out[name] = (len(evs), tot / len(evs), sum(self.work[name]) / len(evs))
            self.pairs[name] = []
            self.work[name] = []
        return out


TIMER = KernelTimer()

# Decision tap (parity tests only; None in production).  When a test sets this to a list, every bilinear-sampling launch and every
# fused-attention forward appends what is needed to recover the piecewise-linear decisions the kernels took - the sampler's
# cells, the layer-1 and layer-2 ReLU masks of the position-bias MLP - so that tests can impose the SAME decisions on the fp64
# oracle (tests/helpers.py Decisions).  Nothing here changes what the kernels compute.
DECISION_TAP = None


def relu_masks_rows(masks):
    """The saved layer-2 ReLU masks [B, H, nst / 32, J, 2, 32] (per-tile storage of the kernels) as [B, H, J, 2, nst] rows - the layout
    smml_deform_attn_relu1_masks writes and the tests decode.  Tests only."""
    B, H, NT, J, _, _ = masks.shape
    return masks.permute(0, 1, 3, 4, 2, 5).reshape(B, H, J, 2, NT * 32)


def relu1_masks(vs, gq, w1, b1, *, B: int, N: int, J: int, groups: int, log_distance: bool = True):
    """Layer-1 ReLU decisions of the position-bias MLP as the kernels evaluate them: int16 [(B G), J, 2, nst] in the bit order of the
    saved layer-2 masks (include/smml.h smml_deform_attn_relu1_masks).  Tests only."""
    L = capi.lib()
    nst = L.smml_deform_attn_nst(N)
    out = torch.empty(B * groups, J, 2, nst, device=vs.device, dtype=torch.int16)
    capi.check(L.smml_deform_attn_relu1_masks(capi.fptr(_c(vs)), capi.fptr(_c(gq)), capi.fptr(_c(w1)), capi.fptr(_c(b1)), capi.ptr(out),
                                              B, N, J, groups, vs.shape[-1], capi.stream(), capi.deform_opts(log_distance=bool(log_distance))),
               "relu1_masks")
    return out


class _ZeroPool:
    """Small zero-initialised fp32 tensors (weight / bias gradients that kernels accumulate into with atomics) carved out
    of one pre-zeroed slab per device: one fill kernel per slab instead of one per tensor (a training step asked for ~40
    of them, each a launch-bound 2-5 us).  Slices are never reused - a slab is dropped when it runs out and lives on
    only as long as its slices do - so a tensor handed out here is as private as torch.zeros(...) would be."""
    SLAB = 1 << 20            # floats (4 MiB)
    LIMIT = 1 << 16           # larger requests go to torch.zeros directly

    def __init__(self):
        self.slabs = {}
        self.enabled = os.environ.get("SMML_ZERO_POOL", "1") != "0"   # measurement switch

    def zeros(self, shape, device) -> torch.Tensor:
        shape = tuple(int(d) for d in shape)
        n = 1
        for d in shape:
            n *= d
        # Under hipGraph capture a pooled slice would be zero only in the capture run: the slab's fill is not part of the graph, so a
        # replay would accumulate onto the previous replay's sums.  torch.zeros inside a capture allocates from the graph's private pool
        # and its fill IS a node of the graph - re-zeroed by every replay.
        if (n == 0 or n > self.LIMIT or torch.device(device).type != "cuda" or not self.enabled
                or torch.cuda.is_current_stream_capturing()):
            return torch.zeros(shape, device=device, dtype=torch.float32)
        key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
        buf, used = self.slabs.get(key, (None, self.SLAB))
        n4 = (n + 3) & ~3                                    # keep every slice 16-byte aligned
        if used + n4 > self.SLAB:
            buf, used = torch.zeros(self.SLAB, device=device, dtype=torch.float32), 0
        self.slabs[key] = (buf, used + n4)
        return buf[used:used + n].view(shape)


_ZEROS = _ZeroPool()


def _zeros_like(t: torch.Tensor) -> torch.Tensor:
    return _ZEROS.zeros(t.shape, t.device)


def _c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _gemm(A, B, C, *, M, N, K, sam, sak, sbk, sbn, ldc, bias=None, bias_mode=0, rows_per_bias=1, bias_ld=0,
          residual=None, ldr=0, act=ACT_NONE, splitk=1, alpha=1.0, nb0=1, nb1=1, sa0=0, sa1=0, sb0=0, sb1=0,
          sc0=0, sc1=0, sbias0=0, sbias1=0, beta=1.0, accumulate=0):
    capi.check(capi.lib().smml_gemm_f32(
        capi.fptr(A), capi.fptr(B), capi.fptr(C), capi.fptr(bias), capi.fptr(residual), M, N, K,
        sam, sak, sbk, sbn, ldc, ldr, nb0, nb1, sa0, sa1, sb0, sb1, sc0, sc1, sbias0, sbias1,
        bias_mode, rows_per_bias, bias_ld, act, splitk, accumulate, float(alpha), float(beta), capi.stream()), "gemm")


class _prec:
    """GEMM precision mode (include/smml.h smml_gemm_set_mode) for the launches issued inside the block; mode 0 / None = leave."""

    def __init__(self, mode):
        self.mode = int(mode or 0)

    def __enter__(self):
        if self.mode:
            L = capi.lib()
            self.prev = L.smml_gemm_get_mode()
            L.smml_gemm_set_mode(self.mode)

    def __exit__(self, *a):
        if self.mode:
            capi.lib().smml_gemm_set_mode(self.prev)


def _splitk_for(out_rows: int, out_cols: int, k: int, batches: int = 1) -> int:
    tiles = ((out_rows + 127) // 128) * ((out_cols + 63) // 64) * batches
    want = max(1, 1024 // max(tiles, 1))
    return int(max(1, min(want, (k + 255) // 256, 65535 // max(batches, 1))))


def colsum(x2d_or_3d: torch.Tensor, scale: float = 1.0) -> torch.Tensor:
    """x [nb, R, C] -> [nb, C] column sums * scale."""
    x = x2d_or_3d
    nb, R, Cc = x.shape
    out = _ZEROS.zeros((nb, Cc), x.device)
    capi.check(capi.lib().smml_colsum_f32(capi.fptr(x), capi.fptr(out), nb, R, Cc, float(scale), capi.stream()), "colsum")
    return out


# ------------------------------------------------------------------------------------------------
# y = act(x W^T + bias) [+ residual]; bias is [N] or one row per block of `rows_per_bias` rows
# ------------------------------------------------------------------------------------------------
class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act, rows_per_bias, residual, prec=0):
        x = _c(x); weight = _c(weight)
        ctx.prec = 3 if prec == 4 else prec          # fp16 operands are for forward-range values: the gradient products of mode 4 run in bf16
        K = x.shape[-1]
        M = x.numel() // K
        N = weight.shape[0]
        y = torch.empty(*x.shape[:-1], N, device=x.device, dtype=torch.float32)
        bias_mode = 0
        if bias is not None:
            bias = _c(bias)
            bias_mode = 2 if bias.dim() == 2 else 1
        res = _c(residual) if residual is not None else None
        with _prec(prec):
            _gemm(x, weight, y, M=M, N=N, K=K, sam=K, sak=1, sbk=1, sbn=K, ldc=N, bias=bias, bias_mode=bias_mode,
                  rows_per_bias=rows_per_bias, bias_ld=N, residual=res, ldr=N, act=act)
        ctx.act, ctx.rows_per_bias, ctx.bias_mode = act, rows_per_bias, bias_mode
        ctx.has_res = residual is not None
        ctx.save_for_backward(x, weight, y if act != ACT_NONE else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, y = ctx.saved_tensors
        dy = _c(dy)
        K = x.shape[-1]
        M = x.numel() // K
        N = weight.shape[0]
        dres = dy if ctx.has_res else None
        if ctx.act == ACT_RELU:
            dpre = torch.empty_like(dy)
            capi.check(capi.lib().smml_relu_bwd_f32(capi.fptr(dy), capi.fptr(y), capi.fptr(dpre), dy.numel(),
                                                    capi.stream()), "relu_bwd")
        elif ctx.act == ACT_TANH:
            dpre = dy * (1.0 - y * y)          # only ever [B, C]-sized (Pooler)
        else:
            dpre = dy
        dx = dw = db = None
        with _prec(ctx.prec):
            if ctx.needs_input_grad[0]:
                dx = torch.empty_like(x)
                _gemm(dpre, weight, dx, M=M, N=K, K=N, sam=N, sak=1, sbk=K, sbn=1, ldc=K)
            if ctx.needs_input_grad[1]:
                dw = _zeros_like(weight)
                _gemm(dpre, x, dw, M=N, N=K, K=M, sam=1, sak=N, sbk=K, sbn=1, ldc=K, splitk=_splitk_for(N, K, M))
        if ctx.bias_mode and ctx.needs_input_grad[2]:
            if ctx.bias_mode == 1:
                db = colsum(dpre.reshape(1, M, N))[0]
            else:
                db = colsum(dpre.reshape(M // ctx.rows_per_bias, ctx.rows_per_bias, N))
        return dx, dw, db, None, None, dres, None


def linear(x, weight, bias=None, act: int = ACT_NONE, rows_per_bias: int = 1, residual=None, prec: int = 0):
    """act(x W^T + bias) (+ residual).  prec: GEMM precision mode of the forward and backward products (0 automatic / exact
    fp32, 2 three-term split bf16, 3 single-term bf16 operands - the 16-bit compute mode, 4 single-term fp16 operands in the forward
    product and bf16 ones in the two gradient products - the fp16 compute modes)."""
    return _Linear.apply(x, weight, bias, act, rows_per_bias, residual, prec)


# ------------------------------------------------------------------------------------------------
# two ReLU(Linear) heads over ONE input: the two branches of DeformPathomicNet read the same bag (models/model.py:488-497,
# SURVEY.md K1) - one batched launch per direction, the bag's rows are addressed once per launch instead of once per branch
# ------------------------------------------------------------------------------------------------
class _DualLinearRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w0, b0, w1, b1):
        x = _c(x)
        K = x.shape[-1]
        M = x.numel() // K
        N = w0.shape[0]
        w = torch.stack((_c(w0), _c(w1)))                       # [2, N, K]
        b = torch.stack((_c(b0), _c(b1)))                       # [2, N]
        y = torch.empty(2, *x.shape[:-1], N, device=x.device, dtype=torch.float32)
        _gemm(x, w, y, M=M, N=N, K=K, sam=K, sak=1, sbk=1, sbn=K, ldc=N, bias=b, bias_mode=1, act=ACT_RELU, nb1=2, sa1=0,
              sb1=N * K, sc1=M * N, sbias1=N)
        ctx.save_for_backward(x, y)
        return y[0], y[1]

    @staticmethod
    def backward(ctx, dy0, dy1):
        x, y = ctx.saved_tensors
        K = x.shape[-1]
        M = x.numel() // K
        N = y.shape[-1]
        L = capi.lib()
        dpre = torch.empty_like(y)
        for i, dy in enumerate((dy0, dy1)):
            capi.check(L.smml_relu_bwd_f32(capi.fptr(_c(dy)), capi.fptr(y[i]), capi.fptr(dpre[i]), dy.numel(), capi.stream()), "relu_bwd")
        dx = None
        if ctx.needs_input_grad[0]:
            raise RuntimeError("dual_linear_relu: the shared input is a bag of features (no gradient path is built)")
        dw = _ZEROS.zeros((2, N, K), x.device)
        _gemm(dpre, x, dw, M=N, N=K, K=M, sam=1, sak=N, sbk=K, sbn=1, ldc=K, nb1=2, sa1=M * N, sb1=0, sc1=N * K,
              splitk=_splitk_for(N, K, M, 2))
        db = colsum(dpre.reshape(2, M, N))
        return dx, dw[0], db[0], dw[1], db[1]


def dual_linear_relu(x, w0, b0, w1, b1):
    """(relu(x w0^T + b0), relu(x w1^T + b1)) for two heads of equal shape over one input."""
    return _DualLinearRelu.apply(x, w0, b0, w1, b1)


# ------------------------------------------------------------------------------------------------
# bf16-storage linear (csrc/gemm_b16.hip): x bf16 [.., K], W an fp32 parameter [N, K] -> y bf16 or fp32 [.., N]
# ------------------------------------------------------------------------------------------------
def _bptr(t: torch.Tensor):
    if t.dtype != torch.bfloat16:
        raise RuntimeError(f"smml bf16 kernel got dtype {t.dtype}")
    return capi.ptr(t)


def gemm_b16(A, B, C, *, M, N, K, lda, ldb, ldc, trans=False, bias=None, splitk=1, nb=1, sa=0, sb=0, sc=0, a_off=0, b_off=0, c_off=0):
    """C = A B^T (trans False: A [M, K], B [N, K]) or A^T B (trans True: A [K, M], B [K, N]); A, B bf16, C bf16 or fp32.  nb problems at
    element strides sa / sb / sc (sc = 0: they add up in one zeroed fp32 output); *_off: element offsets of the first problem."""
    if C.dtype not in (torch.bfloat16, torch.float32):
        raise RuntimeError(f"gemm_b16: output dtype {C.dtype}")
    _bptr(A), _bptr(B), capi.ptr(C)                            # device / contiguity checks
    pa = C_.c_void_p(A.data_ptr() + 2 * a_off)
    pb = C_.c_void_p(B.data_ptr() + 2 * b_off)
    pc = C_.c_void_p(C.data_ptr() + C.element_size() * c_off)
    capi.check(capi.lib().smml_gemm_b16_batched(pa, pb, pc, capi.fptr(bias), M, N, K, lda, ldb, ldc, int(trans),
                                                int(C.dtype == torch.bfloat16), splitk, nb, sa, sb, sc, capi.stream()), "gemm_b16")


class _LinearB16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, out_bf16, skip):
        if x.dtype != torch.bfloat16:
            raise RuntimeError("linear_b16: x must be bf16 (the caller casts once)")
        x = x if x.is_contiguous() else x.contiguous()
        wb = weight.detach().to(torch.bfloat16)                 # the parameter stays fp32; [N, K] bf16 is 1.5 MB at most here
        K = x.shape[-1]
        N = weight.shape[0]
        if skip:
            if x.dim() != 3 or not 0 < skip < x.shape[1]:
                raise RuntimeError("linear_b16: skip needs x [b, rows, K] with 0 < skip < rows")
            b, rows = x.shape[0], x.shape[1]
            y = torch.empty(b, rows - skip, N, device=x.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
            gemm_b16(x, wb, y, M=rows - skip, N=N, K=K, lda=K, ldb=K, ldc=N, bias=_c(bias) if bias is not None else None, nb=b,
                     sa=rows * K, sc=(rows - skip) * N, a_off=skip * K)
        else:
            M = x.numel() // K
            y = torch.empty(*x.shape[:-1], N, device=x.device, dtype=torch.bfloat16 if out_bf16 else torch.float32)
            gemm_b16(x, wb, y, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=_c(bias) if bias is not None else None)
        ctx.has_bias = bias is not None
        ctx.skip = skip
        ctx.save_for_backward(x, wb)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wb = ctx.saved_tensors
        K = x.shape[-1]
        N = wb.shape[0]
        skip = ctx.skip
        rows_y = dy.numel() // N
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = colsum(_c(dy).reshape(1, rows_y, N))[0]
        dyb = dy if dy.dtype == torch.bfloat16 else dy.to(torch.bfloat16)
        dyb = dyb if dyb.is_contiguous() else dyb.contiguous()
        dx = dw = None
        if skip:
            b, rows = x.shape[0], x.shape[1]
            n = rows - skip
            if ctx.needs_input_grad[0]:
                wt = wb.t().contiguous()
                dx = torch.empty_like(x)
                dx[:, :skip].zero_()                                # the skipped rows took no part
                gemm_b16(dyb, wt, dx, M=n, N=K, K=N, lda=N, ldb=N, ldc=K, nb=b, sa=n * N, sc=rows * K, c_off=skip * K)
            if ctx.needs_input_grad[1]:
                dw = _ZEROS.zeros((N, K), x.device)
                gemm_b16(dyb, x, dw, M=N, N=K, K=n, lda=N, ldb=K, ldc=K, trans=True, splitk=0, nb=b,
                         sa=n * N, sb=rows * K, sc=0, b_off=skip * K)
            return dx, dw, db, None, None
        M = x.numel() // K
        if ctx.needs_input_grad[0]:
            wt = wb.t().contiguous()                            # [K, N]: dx = dy (W^T)^T is the k-contiguous form again
            dx = torch.empty_like(x)
            gemm_b16(dyb, wt, dx, M=M, N=K, K=N, lda=N, ldb=N, ldc=K)
        if ctx.needs_input_grad[1]:
            dw = _ZEROS.zeros((N, K), x.device)
            gemm_b16(dyb, x, dw, M=N, N=K, K=M, lda=N, ldb=K, ldc=K, trans=True, splitk=0)
        return dx, dw, db, None, None


def linear_b16(x, weight, bias=None, out_bf16: bool = True, skip: int = 0):
    """x W^T (+ bias) with bf16 operands in memory and fp32 accumulation; x bf16, weight / bias fp32 parameters (gradients fp32), the
    result bf16 (out_bf16) or fp32; dx is bf16.  skip > 0 (x [b, rows, K]): the first `skip` rows of every bag are left out - the result
    is [b, rows - skip, N] (the zero rows the Nystrom block pads in front of a bag never reach the output projection)."""
    return _LinearB16.apply(x, weight, bias, out_bf16, skip)


# ------------------------------------------------------------------------------------------------
# The bf16-storage pipeline of the Nystrom block (bf16 compute mode): q / k / v stay in the token-major bf16 buffer qkv [b, n, 3, h, 64]
# the projection writes; the four Functions below read them there at their strides and assemble ONE gradient buffer dqkv of the same
# layout, which the projection's backward consumes.  The buffer is threaded through the chain as an extra output of each stage
#   qkv_project16 -> attention16_keys_long -> resconv16 -> attention16_queries_long
# (each stage returns qkv itself): in backward the chain runs in reverse, the LAST stage allocates dqkv and writes dq, resconv16 writes the v
# part, attention16_keys_long writes dk and adds its dv, qkv_project16 adds the landmark means' gradients and runs the two GEMMs - no stage
# returns a full-size gradient of its own for autograd to add up (six n'-sized adds / copies per step in the fp32-storage composition).
# A stage modifies the incoming dqkv in place: it was produced by the stage after it for this purpose and nobody else holds it.
# ------------------------------------------------------------------------------------------------
def _qkv_dims(qkv, heads):
    b, n, c = qkv.shape
    if qkv.dtype != torch.bfloat16 or not qkv.is_contiguous() or c % (3 * heads):
        raise RuntimeError("qkv must be a contiguous bf16 [b, n, 3 h d] buffer")
    d = c // (3 * heads)
    if d != 64:
        raise RuntimeError("the bf16-storage attention is built for head dim 64")
    return b, n, c, d


def _part(t, i, hd):
    """device pointer of part i (0 q, 1 k, 2 v) of a qkv-shaped bf16 buffer (same strides, shifted base)"""
    _bptr(t)
    return C_.c_void_p(t.data_ptr() + 2 * i * hd)


class _QKVProject16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, heads, l, pad):
        if x.dtype != torch.bfloat16:
            raise RuntimeError("qkv_project16: x must be bf16")
        x = x if x.is_contiguous() else x.contiguous()
        wb = weight.detach().to(torch.bfloat16)
        b, n0, K = x.shape
        n = n0 + pad                                               # rows of a bag in the buffers: `pad` zero rows in front
        N = weight.shape[0]
        d = N // (3 * heads)
        qkv = torch.empty(b, n, N, device=x.device, dtype=torch.bfloat16)
        if pad:
            qkv[:, :pad].zero_()                                   # x W^T of a zero row (the projection has no bias)
        gemm_b16(x, wb, qkv, M=n0, N=N, K=K, lda=K, ldb=K, ldc=N, nb=b, sa=n0 * K, sc=n * N, c_off=pad * N)
        m = n // l
        ql = torch.empty(b, heads, m, d, device=x.device, dtype=torch.float32)
        kl = torch.empty_like(ql)
        capi.check(capi.lib().smml_segment_mean_b16(_bptr(qkv), capi.fptr(ql), capi.fptr(kl), b, n, l, heads, d, capi.stream()), "segment_mean_b16")
        ctx.cfg = (heads, l, d, pad)
        ctx.save_for_backward(x, wb)
        return qkv, ql, kl

    @staticmethod
    def backward(ctx, dqkv, dql, dkl):
        x, wb = ctx.saved_tensors
        heads, l, d, pad = ctx.cfg
        b, n0, K = x.shape
        n = n0 + pad
        N = wb.shape[0]
        if dqkv is None:
            dqkv = torch.zeros(b, n, N, device=x.device, dtype=torch.bfloat16)
        dqkv = dqkv if dqkv.is_contiguous() else dqkv.contiguous()
        if dql is not None or dkl is not None:
            zero = None
            if dql is None or dkl is None:
                zero = torch.zeros(b, heads, n // l, d, device=x.device, dtype=torch.float32)
            capi.check(capi.lib().smml_segment_mean_bwd_add_b16(_bptr(dqkv), capi.fptr(_c(dql) if dql is not None else zero),
                                                                capi.fptr(_c(dkl) if dkl is not None else zero), b, n, l, heads, d,
                                                                capi.stream()), "segment_mean_bwd_add_b16")
        dx = dw = None
        if ctx.needs_input_grad[0]:
            wt = wb.t().contiguous()
            dx = torch.empty_like(x)
            gemm_b16(dqkv, wt, dx, M=n0, N=K, K=N, lda=N, ldb=N, ldc=K, nb=b, sa=n * N, sc=n0 * K, a_off=pad * N)
        if ctx.needs_input_grad[1]:
            dw = _ZEROS.zeros((N, K), x.device)
            gemm_b16(dqkv, x, dw, M=N, N=K, K=n0, lda=N, ldb=K, ldc=K, trans=True, splitk=0, nb=b,
                     sa=n * N, sb=n0 * K, sc=0, a_off=pad * N)
        return dx, dw, None, None, None


def qkv_project16(x, weight, heads: int, l: int, pad: int = 0):
    """x bf16 [b, n0, K] -> (qkv bf16 [b, n, 3 h d], n = pad + n0: `pad` zero rows in front of every bag (NystromAttention.py:82), then x W^T;
    ql, kl fp32 [b, h, n / l, d] = means of q / k over l consecutive tokens).  The padded bag is never materialised: the GEMMs address
    a bag's real rows in place."""
    return _QKVProject16.apply(x, weight, heads, l, pad)


class _Attention16KeysLong(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ql, qkv, heads, scale, dv_accumulate):
        b, n, c, d = _qkv_dims(qkv, heads)
        ql = _c(ql)
        m = ql.shape[2]
        out = torch.empty(b, heads, m, d, device=qkv.device, dtype=torch.float32)
        lse2 = torch.empty(b * heads, m, device=qkv.device, dtype=torch.float32)
        L = capi.lib()
        wsb = L.smml_attn16_fwd_workspace_bytes(b * heads, m, n)
        ws = torch.empty((wsb + 3) // 4, device=qkv.device, dtype=torch.float32) if wsb else None
        hd = heads * d
        capi.check(L.smml_attn16_fwd_b16(capi.fptr(ql), _part(qkv, 1, hd), _part(qkv, 2, hd), capi.fptr(out), None, capi.fptr(lse2),
                                         capi.fptr(ws), wsb, b, heads, m, n, float(scale), 0, n * c, d, c, 0, 0, 0, capi.stream()),
                   "attn16_fwd_b16 (keys long)")
        ctx.cfg = (heads, float(scale), bool(dv_accumulate))
        ctx.save_for_backward(ql, qkv, out, lse2)
        return out, qkv

    @staticmethod
    def backward(ctx, dout, dqkv):
        ql, qkv, out, lse2 = ctx.saved_tensors
        heads, scale, dv_acc = ctx.cfg
        b, n, c, d = _qkv_dims(qkv, heads)
        m = ql.shape[2]
        hd = heads * d
        if dqkv is None:                                   # nothing after this stage contributed: start the buffer here
            dqkv = torch.zeros_like(qkv)
            dv_acc = True
        if dout is None:                                   # this product has no gradient: its parts of the buffer must still be defined
            parts = dqkv.view(b, n, 3, hd)
            parts[:, :, 1].zero_()
            if not dv_acc:
                parts[:, :, 2].zero_()
            return None, dqkv, None, None, None
        dout = _c(dout)
        dql = torch.empty_like(ql)
        L = capi.lib()
        wsb = L.smml_attn16_bwd_workspace_bytes(b * heads, m, n)
        ws = torch.empty((wsb + 3) // 4, device=qkv.device, dtype=torch.float32)
        capi.check(L.smml_attn16_bwd_b16(capi.fptr(ql), _part(qkv, 1, hd), _part(qkv, 2, hd), capi.fptr(out), None, capi.fptr(dout),
                                         capi.fptr(lse2), capi.fptr(dql), _part(dqkv, 1, hd), _part(dqkv, 2, hd), capi.fptr(ws), wsb, b,
                                         heads, m, n, scale, 0, n * c, d, c, 0, 0, 0, n * c, d, c, int(dv_acc), capi.stream()),
                   "attn16_bwd_b16 (keys long)")
        return dql, dqkv, None, None, None


def attention16_keys_long(ql, qkv, *, heads: int, scale: float, dv_accumulate: bool):
    """softmax(scale ql k^T) v with k / v read in the qkv buffer -> (fp32 [b, h, m, 64], qkv).  Backward writes dk into the k part of the
    threaded gradient buffer and writes (dv_accumulate False) or adds (True: a stage after this one already wrote dv) its dv."""
    return _Attention16KeysLong.apply(ql, qkv, heads, scale, dv_accumulate)


class _ResConv16(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, w, heads):
        b, n, c, d = _qkv_dims(qkv, heads)
        hd = heads * d
        kw = w.shape[2]
        w2 = _c(w.reshape(heads, kw))
        res = torch.empty(b, n, hd, device=qkv.device, dtype=torch.bfloat16)
        capi.check(capi.lib().smml_resconv_b16(_part(qkv, 2, hd), capi.fptr(w2), _bptr(res), b, heads, n, d, kw, n * c, c, n * hd, hd, 0,
                                               capi.stream()), "resconv_b16")
        ctx.cfg = (heads, kw, tuple(w.shape))
        ctx.save_for_backward(qkv, w2)
        return res, qkv

    @staticmethod
    def backward(ctx, dres, dqkv):
        qkv, w2 = ctx.saved_tensors
        heads, kw, wshape = ctx.cfg
        b, n, c, d = _qkv_dims(qkv, heads)
        hd = heads * d
        if dqkv is None:
            dqkv = torch.zeros_like(qkv)
        dw = None
        if dres is None:                                   # no gradient through the convolution: the v part must still be defined
            dqkv.view(b, n, 3, hd)[:, :, 2].zero_()
        else:
            dres = dres if dres.is_contiguous() else dres.contiguous()
            L = capi.lib()
            capi.check(L.smml_resconv_b16(_bptr(dres), capi.fptr(w2), _part(dqkv, 2, hd), b, heads, n, d, kw, n * hd, hd, n * c, c, 1,
                                          capi.stream()), "resconv_b16 (data gradient)")
            if ctx.needs_input_grad[1]:
                dw2 = _ZEROS.zeros((heads, kw), qkv.device)
                capi.check(L.smml_resconv_wgrad_b16(_bptr(dres), _part(qkv, 2, hd), capi.fptr(dw2), b, heads, n, d, kw, n * hd, hd, n * c, c,
                                                    capi.stream()), "resconv_wgrad_b16")
                dw = dw2.view(wshape)
        return dqkv, dw, None


def resconv16(qkv, w, *, heads: int):
    """depthwise 33-tap convolution of v over tokens, v read in the qkv buffer -> (bf16 [b, n, h 64], qkv).  Backward WRITES the v part of the
    threaded gradient buffer (it is the first stage of the backward chain that touches it)."""
    return _ResConv16.apply(qkv, w, heads)


class _Attention16QueriesLong(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, kl, w, residual, heads, scale):
        b, n, c, d = _qkv_dims(qkv, heads)
        kl, w = _c(kl), _c(w)
        m = kl.shape[2]
        hd = heads * d
        if residual is not None and (residual.dtype != torch.bfloat16 or tuple(residual.shape) != (b, n, hd) or not residual.is_contiguous()):
            raise RuntimeError("attention16_queries_long: the residual is a contiguous bf16 [b, n, h 64] tensor")
        out = torch.empty(b, n, hd, device=qkv.device, dtype=torch.bfloat16)
        lse2 = torch.empty(b * heads, n, device=qkv.device, dtype=torch.float32)
        capi.check(capi.lib().smml_attn16_fwd_b16(_part(qkv, 0, hd), capi.fptr(kl), capi.fptr(w), _bptr(out),
                                                  _bptr(residual) if residual is not None else None, capi.fptr(lse2), None, 0, b, heads, n, m,
                                                  float(scale), 1, n * c, d, c, n * hd, d, hd, capi.stream()), "attn16_fwd_b16 (queries long)")
        ctx.cfg = (heads, float(scale), residual is not None)
        ctx.save_for_backward(qkv, kl, w, out, residual, lse2)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, kl, w, out, residual, lse2 = ctx.saved_tensors
        heads, scale, has_res = ctx.cfg
        b, n, c, d = _qkv_dims(qkv, heads)
        m = kl.shape[2]
        hd = heads * d
        if dout.dtype != torch.bfloat16:
            dout = dout.to(torch.bfloat16)
        dout = dout if dout.is_contiguous() else dout.contiguous()
        dqkv = torch.empty_like(qkv)                       # the head of the backward chain: q part written here, k / v parts by the earlier stages
        dkl, dw = torch.empty_like(kl), torch.empty_like(w)
        L = capi.lib()
        wsb = L.smml_attn16_bwd_workspace_bytes(b * heads, n, m)
        ws = torch.empty((wsb + 3) // 4, device=qkv.device, dtype=torch.float32)
        capi.check(L.smml_attn16_bwd_b16(_part(qkv, 0, hd), capi.fptr(kl), capi.fptr(w), _bptr(out), _bptr(residual) if has_res else None,
                                         _bptr(dout), capi.fptr(lse2), _part(dqkv, 0, hd), capi.fptr(dkl), capi.fptr(dw), capi.fptr(ws), wsb,
                                         b, heads, n, m, scale, 1, n * c, d, c, n * hd, d, hd, n * c, d, c, 0, capi.stream()),
                   "attn16_bwd_b16 (queries long)")
        return dqkv, dkl, dw, (dout if has_res else None), None, None


def attention16_queries_long(qkv, kl, w, residual=None, *, heads: int, scale: float):
    """softmax(scale q kl^T) w + residual with q read in the qkv buffer -> bf16 [b, n, h 64] (the output projection's operand).  Backward
    allocates the threaded gradient buffer dqkv and writes its q part; the k and v parts are left to the stages before this one."""
    return _Attention16QueriesLong.apply(qkv, kl, w, residual, heads, scale)


class _HeadMajorQKV(torch.autograd.Function):
    """qkv [b, n, 3 h d] (bf16 or fp32, token-major: what the projection writes) -> q, k, v fp32 [b, h, n, d], three slices of one buffer filled by
    ONE strided copy; backward writes the three gradients into one token-major buffer of the input's dtype (no stack / cat of head-major pieces)."""

    @staticmethod
    def forward(ctx, qkv, heads):
        b, n, c = qkv.shape
        d = c // (3 * heads)
        ctx.shape = (b, n, heads, d)
        ctx.in_dtype = qkv.dtype                  # bf16 (the bf16 projections) or fp32 (exact path, fp16 mode): the gradient buffer follows it
        buf = torch.empty(3, b, heads, n, d, device=qkv.device, dtype=torch.float32)
        buf.copy_(qkv.view(b, n, 3, heads, d).permute(2, 0, 3, 1, 4))
        return buf[0], buf[1], buf[2]

    @staticmethod
    def backward(ctx, dq, dk, dv):
        b, n, h, d = ctx.shape
        dqkv = torch.empty(b, n, 3, h, d, device=dq.device, dtype=ctx.in_dtype)
        view = dqkv.permute(2, 0, 3, 1, 4)
        for i, g in enumerate((dq, dk, dv)):
            if g is None:
                view[i].zero_()
            else:
                view[i].copy_(g)
        return dqkv.view(b, n, 3 * h * d), None


def head_major_qkv(qkv, heads: int):
    return _HeadMajorQKV.apply(qkv, heads)


# ------------------------------------------------------------------------------------------------
# grouped 1x1 convolution on token-major data: x [B, n, Cin], w [Cout, Cin / groups] -> [B, n, Cout]
# ------------------------------------------------------------------------------------------------
class _GroupedPointwise(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, groups):
        x = _c(x)
        w = _c(weight).reshape(weight.shape[0], -1)
        Cin, Cout = x.shape[-1], w.shape[0]
        cin_g, cout_g = Cin // groups, Cout // groups
        assert w.shape[1] == cin_g, "weight does not match the group structure"
        M = x.numel() // Cin
        y = torch.empty(*x.shape[:-1], Cout, device=x.device, dtype=torch.float32)
        _gemm(x, w, y, M=M, N=cout_g, K=cin_g, sam=Cin, sak=1, sbk=1, sbn=cin_g, ldc=Cout, nb1=groups,
              sa1=cin_g, sb1=cout_g * cin_g, sc1=cout_g)
        ctx.groups = groups
        ctx.wshape = weight.shape
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        G = ctx.groups
        Cin, Cout = x.shape[-1], w.shape[0]
        cin_g, cout_g = Cin // G, Cout // G
        M = x.numel() // Cin
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            _gemm(dy, w, dx, M=M, N=cin_g, K=cout_g, sam=Cout, sak=1, sbk=cin_g, sbn=1, ldc=Cin, nb1=G,
                  sa1=cout_g, sb1=cout_g * cin_g, sc1=cin_g)
        if ctx.needs_input_grad[1]:
            dw = _zeros_like(w)
            _gemm(dy, x, dw, M=cout_g, N=cin_g, K=M, sam=1, sak=Cout, sbk=Cin, sbn=1, ldc=cin_g, nb1=G,
                  sa1=cout_g, sb1=cin_g, sc1=cout_g * cin_g, splitk=_splitk_for(cout_g, cin_g, M, G))
            dw = dw.reshape(ctx.wshape)
        return dx, dw, None


def grouped_pointwise(x, weight, groups: int = 1):
    return _GroupedPointwise.apply(x, weight, groups)


# ------------------------------------------------------------------------------------------------
# LayerNorm (optionally followed by the mean over tokens, Pooler's first step)
# ------------------------------------------------------------------------------------------------
class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, token_mean):
        x = _c(x); gamma = _c(gamma); beta = _c(beta)
        Cc = x.shape[-1]
        R = x.numel() // Cc
        y = torch.empty_like(x)
        mean = torch.empty(R, device=x.device, dtype=torch.float32)
        rstd = torch.empty(R, device=x.device, dtype=torch.float32)
        capi.check(capi.lib().smml_layernorm_fwd_f32(capi.fptr(x), capi.fptr(gamma), capi.fptr(beta), capi.fptr(y),
                                                     capi.fptr(mean), capi.fptr(rstd), R, Cc, float(eps),
                                                     capi.stream()), "layernorm_fwd")
        ctx.token_mean = token_mean
        ctx.save_for_backward(x, gamma, mean, rstd)
        if token_mean:
            assert x.dim() == 3
            return colsum(y, 1.0 / x.shape[1])
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        dy = _c(dy)
        Cc = x.shape[-1]
        R = x.numel() // Cc
        dx = torch.empty_like(x)
        dg = _zeros_like(gamma)
        db = _zeros_like(gamma)
        rows_per_dy, scale = (x.shape[1], 1.0 / x.shape[1]) if ctx.token_mean else (1, 1.0)
        capi.check(capi.lib().smml_layernorm_bwd_f32(capi.fptr(x), capi.fptr(dy), capi.fptr(gamma), capi.fptr(mean),
                                                     capi.fptr(rstd), capi.fptr(dx), capi.fptr(dg), capi.fptr(db), R, Cc,
                                                     rows_per_dy, float(scale), 0, capi.stream()), "layernorm_bwd")
        return dx, dg, db, None, None


def layer_norm(x, gamma, beta, eps: float = 1e-5):
    return _LayerNorm.apply(x, gamma, beta, eps, False)


def layer_norm_token_mean(x, gamma, beta, eps: float = 1e-5):
    """mean over tokens of LayerNorm(x): x [B, n, C] -> [B, C]."""
    return _LayerNorm.apply(x, gamma, beta, eps, True)


# ------------------------------------------------------------------------------------------------
# offset network: q [B, Hh, Ww, G*dg] -> vgrid [(B G), posdim, th, tw] (or [(B G), t]), vs [(B G), J, posdim]
# ------------------------------------------------------------------------------------------------
class GradFork:
    """q feeds two consumers - the offsets network and the fused attention core - and autograd would add their two [B, N, 512] gradients with
    an elementwise kernel (492 MB of traffic per 8-bag step).  The attention's backward always runs first (the offsets network's backward
    needs the d vs it produces): it parks its dq here, the offsets backward ADDS its own contribution into that buffer in place
    (smml_offsets_bwd_f32, accumulate_dq = 1) and reports no gradient of its own for q.  One object per forward call; None = plain autograd."""
    __slots__ = ("dq",)

    def __init__(self):
        self.dq = None


class _Offsets(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, w0, b0, w2, groups, ks, r, posdim, offset_scale, fork=None):
        ctx.fork = fork
        q = _c(q); w0c = _c(w0); b0c = _c(b0); w2c = _c(w2)
        B, Hh, Ww, inner = q.shape
        dg = inner // groups
        L = capi.lib()
        tw = L.smml_offsets_out_len(Ww, ks, r)
        th = L.smml_offsets_out_len(Hh, ks, r) if posdim == 2 else 1
        if th <= 0 or tw <= 0:
            raise RuntimeError("token grid too small for the offset convolution")
        J = th * tw
        vgrid = torch.empty((B * groups, 2, th, tw) if posdim == 2 else (B * groups, tw), device=q.device,
                            dtype=torch.float32)
        vs = torch.empty(B * groups, J, posdim, device=q.device, dtype=torch.float32)
        capi.check(L.smml_offsets_fwd_f32(capi.fptr(q), capi.fptr(w0c), capi.fptr(b0c), capi.fptr(w2c),
                                          capi.fptr(vgrid), capi.fptr(vs), B, Hh, Ww, groups, dg, ks, r, posdim,
                                          float(offset_scale), capi.stream()), "offsets_fwd")
        ctx.cfg = (groups, ks, r, posdim, float(offset_scale))
        ctx.save_for_backward(q, w0c, b0c, w2c)
        return vgrid, vs

    @staticmethod
    def backward(ctx, dvgrid, dvs):
        q, w0, b0, w2 = ctx.saved_tensors
        groups, ks, r, posdim, offset_scale = ctx.cfg
        B, Hh, Ww, inner = q.shape
        dg = inner // groups
        fork = ctx.fork
        parked = fork.dq if fork is not None else None
        if fork is not None:
            fork.dq = None
        acc = parked is not None and parked.numel() == q.numel() and parked.is_contiguous() and parked.dtype == torch.float32
        dq = parked if acc else torch.empty_like(q)
        dw0, db0, dw2 = torch.empty_like(w0), torch.empty_like(b0), torch.empty_like(w2)
        dvgrid = _c(dvgrid) if dvgrid is not None else None
        dvs = _c(dvs) if dvs is not None else None
        L = capi.lib()
        wsb = L.smml_offsets_bwd_workspace_bytes(B, Hh, Ww, groups, dg, ks, r, posdim)
        ws = torch.empty((wsb + 3) // 4, device=q.device, dtype=torch.float32)
        capi.check(L.smml_offsets_bwd_f32(capi.fptr(q), capi.fptr(w0), capi.fptr(b0), capi.fptr(w2),
                                          capi.fptr(dvgrid), capi.fptr(dvs), capi.fptr(dq), capi.fptr(dw0),
                                          capi.fptr(db0), capi.fptr(dw2), capi.fptr(ws), wsb, B, Hh, Ww, groups, dg, ks, r,
                                          posdim, offset_scale, 1 if acc else 0, capi.stream()), "offsets_bwd")
        # acc: the contribution went into the attention's dq in place - autograd already holds that tensor as q's gradient
        return (None if acc else dq), dw0, db0, dw2, None, None, None, None, None, None


def offsets(q, w0, b0, w2, *, groups, ks, r, posdim, offset_scale, fork=None):
    """fork: a GradFork shared with the deform_attention call that consumes the same q (see GradFork)."""
    return _Offsets.apply(q, w0, b0, w2, groups, ks, r, posdim, offset_scale, fork)


# ------------------------------------------------------------------------------------------------
# bilinear sampling: x [B, Hh, Ww, C], vs [(B G), J, posdim] -> kv [B, J, C]
# ------------------------------------------------------------------------------------------------
class _Sample(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, vs, groups, posdim):
        x = _c(x); vs = _c(vs)
        B, Hh, Ww, Cc = x.shape
        J = vs.shape[1]
        kv = torch.empty(B, J, Cc, device=x.device, dtype=torch.float32)
        capi.check(capi.lib().smml_bilinear_sample_fwd_f32(capi.fptr(x), capi.fptr(vs), capi.fptr(kv), B, Hh, Ww, groups,
                                                           Cc // groups, J, posdim, capi.stream()), "sample_fwd")
        ctx.cfg = (groups, posdim)
        ctx.save_for_backward(x, vs)
        if DECISION_TAP is not None:
            DECISION_TAP.append({"kind": "sample", "vs": vs.detach(), "Hh": Hh, "Ww": Ww, "posdim": posdim})
        return kv

    @staticmethod
    def backward(ctx, dkv):
        x, vs = ctx.saved_tensors
        groups, posdim = ctx.cfg
        B, Hh, Ww, Cc = x.shape
        J = vs.shape[1]
        dx = torch.zeros_like(x)
        dvs = _zeros_like(vs)
        capi.check(capi.lib().smml_bilinear_sample_bwd_f32(capi.fptr(x), capi.fptr(vs), capi.fptr(_c(dkv)), capi.fptr(dx),
                                                           capi.fptr(dvs), B, Hh, Ww, groups, Cc // groups, J, posdim,
                                                           capi.stream()), "sample_bwd")
        return dx, dvs, None, None


def bilinear_sample(x, vs, *, groups, posdim):
    return _Sample.apply(x, vs, groups, posdim)


def bilinear_corners(vs, Hh: int, Ww: int, posdim: int):
    """Integer path of the sampler for vs [n, posdim]: (cx [n,4] int32, cy [n,4] int32, mask [n,4] uint8)."""
    vs = _c(vs).reshape(-1, posdim)
    n = vs.shape[0]
    cx = torch.empty(n, 4, device=vs.device, dtype=torch.int32)
    cy = torch.empty(n, 4, device=vs.device, dtype=torch.int32)
    cm = torch.empty(n, 4, device=vs.device, dtype=torch.uint8)
    capi.check(capi.lib().smml_bilinear_corners_f32(capi.fptr(vs), capi.ptr(cx), capi.ptr(cy), capi.ptr(cm), n, Hh, Ww,
                                                    posdim, capi.stream()), "corners")
    return cx, cy, cm


# ------------------------------------------------------------------------------------------------
# fused attention core with continuous position bias
# ------------------------------------------------------------------------------------------------
_DTYPE16 = {"bf16": (0, torch.bfloat16), "fp16": (1, torch.float16)}


def prec16(compute_dtype) -> int:
    """GEMM precision mode (`linear(..., prec=)`) that goes with a 16-bit compute mode: None -> 0 (exact), 'bf16' -> 3, 'fp16' -> 4."""
    m = _dtype16(compute_dtype)
    return 0 if m is None else (4 if m[0] == 1 else 3)


def _dtype16(compute_dtype):
    """None (the fp32-grade path) or the (C-ABI code, torch dtype) of the 16-bit compute mode."""
    if compute_dtype is None:
        return None
    key = {"bfloat16": "bf16", "float16": "fp16", "half": "fp16"}.get(str(compute_dtype).replace("torch.", ""),
                                                                      str(compute_dtype).replace("torch.", ""))
    if key not in _DTYPE16:
        raise ValueError(f"compute_dtype must be None, 'bf16' or 'fp16' (got {compute_dtype!r})")
    return _DTYPE16[key]


# layer-2 decisions of a cpb_table='forward' backward: 'table' (mask table, include/smml.h) or 'recompute' (layer 2 per pair); measurement / test switch
TABLE_FORWARD_MASKS = __import__("os").environ.get("SMML_TABFWD_MASKS", "table")


# fp32-grade path, 2-D positions: the position bias per LINEAR REGION of its MLP (csrc/cpb_regions.h, include/smml.h) - exact, and the
# default wherever it applies (signed-log offsets, one head per offset group, J <= 1024).  SMML_CPB_REGIONS=0 keeps the per-pair MLP
# kernels (the cross-check of tests/test_gpu_regions.py, and the path of every other configuration).
CPB_REGIONS = __import__("os").environ.get("SMML_CPB_REGIONS", "1") != "0"
REGION_MAX_KEYS = 16384      # RG_MAX_KEYS of csrc/cpb_regions.h
REGION_LDS_CAP = 0           # tests: regions with an id >= this take the global-memory path of the region kernels (0: the default, 2048)


REGION_GRID, REGION_SUB, REGION_SUBCAP, REGION_EDGES, REGION_RCAP = 1024, 8, 16384, 1 << 15, 4096     # csrc/cpb_regions.h
REGION_CODE_SUB0, REGION_CODE_EDGE0 = 4096, 4096 + 16384


def region_tables_view(tables: torch.Tensor):
    """Views into a region-table buffer (tests / diagnostics; layout: region_layout() of csrc/cpb_regions.h): header counters,
    (a0, a1, c) per region, ReLU patterns (D1 | D2 << 32), 16-bit codes of the cells and sub-cells, single-kink records by slot."""
    o = [0]

    def take(nbytes):
        at = o[0]
        o[0] += (nbytes + 255) & ~255
        return at
    G, SUB = REGION_GRID, REGION_SUB
    hdr_o, reg_o, pat_o = take(256), take(REGION_RCAP * 16), take(REGION_RCAP * 8)
    t0_o, t1_o, edge_o = take(G * G * 2), take(REGION_SUBCAP * SUB * SUB * 2), take(REGION_EDGES * 16)
    hdr = tables[hdr_o:hdr_o + 256].view(torch.int32)
    n_sub, n_edge, n_regions = int(hdr[0]), int(hdr[1]), int(hdr[3])
    return {"n_sub": n_sub, "n_edge": n_edge, "n_cand": int(hdr[2]), "n_regions": n_regions, "n_keys": int(hdr[4]), "overflow": int(hdr[5]),
            "n_cand1": int(hdr[6]), "pmax": float(hdr[8:9].view(torch.float32)), "cs": float(hdr[9:10].view(torch.float32)),
            "co": float(hdr[10:11].view(torch.float32)),
            "reg": tables[reg_o:reg_o + REGION_RCAP * 16].view(torch.float32).view(REGION_RCAP, 4)[:n_regions],
            "pat": tables[pat_o:pat_o + REGION_RCAP * 8].view(torch.int64)[:n_regions],
            "t0": tables[t0_o:t0_o + G * G * 2].view(torch.int16).view(G, G),
            "t1": tables[t1_o:t1_o + REGION_SUBCAP * SUB * SUB * 2].view(torch.int16).view(REGION_SUBCAP, SUB * SUB)[:max(min(n_sub, REGION_SUBCAP), 1)],
            "edge": tables[edge_o:edge_o + REGION_EDGES * 16].view(torch.float32).view(REGION_EDGES, 4)}


def region_patterns(region_ids: torch.Tensor, tables: torch.Tensor):
    """(D1, D2) int64 words of the linear piece of every pair of `region_ids` (tests: the ReLU decisions the region kernels stand for);
    pairs without a region (0xFFFF: they evaluated the MLP itself) get -1."""
    pat = region_tables_view(tables)["pat"]
    ids = region_ids.to(torch.int64) & 0xFFFF
    none = ids == 0xFFFF
    p = pat[ids.clamp_max(max(pat.numel() - 1, 0))] if pat.numel() else torch.zeros_like(ids)
    d1, d2 = p & 0xFFFFFFFF, (p >> 32) & 0xFFFFFFFF
    return torch.where(none, torch.full_like(d1, -1), d1), torch.where(none, torch.full_like(d2, -1), d2)


def cpb_regions_build(w1, b1, w2, b2, w3, b3, pmax: float, tables: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The region tables of the position-bias MLP with these parameters over [-pmax, pmax]^2 (uint8 scratch owned by the caller), built
    on the current stream."""
    L = capi.lib()
    nbytes = L.smml_cpb_regions_bytes()
    if tables is None:
        tables = torch.empty(nbytes, device=w1.device, dtype=torch.uint8)
    capi.check(L.smml_cpb_regions_build(capi.fptr(_c(w1)), capi.fptr(_c(b1)), capi.fptr(_c(w2)), capi.fptr(_c(b2)), capi.fptr(_c(w3)),
                                        capi.fptr(_c(b3)), float(pmax), capi.ptr(tables), nbytes, capi.stream()), "cpb_regions_build")
    return tables


_REGION_STREAMS = {}
REGION_PREFETCH = __import__("os").environ.get("SMML_REGION_PREFETCH", "1") != "0"


class RegionPrefetch:
    """Region tables being built on a side stream (they depend on the bias MLP's parameters only, not on the bag): started at the top of
    the module's forward, joined by the attention launch - the ~0.3 ms of the build run beside the projections, the offsets network and
    the sampler instead of in front of the attention kernel."""

    def __init__(self, w1, b1, w2, b2, w3, b3, pmax: float):
        dev = w1.device
        main = torch.cuda.current_stream(dev)
        key = dev.index if dev.index is not None else torch.cuda.current_device()
        side = _REGION_STREAMS.get(key)
        if side is None:
            side = _REGION_STREAMS[key] = torch.cuda.Stream(device=dev)
        self.pmax = float(pmax)
        self.params = tuple(t.data_ptr() for t in (w1, b1, w2, b2, w3, b3)) + tuple(t._version for t in (w1, b1, w2, b2, w3, b3))
        # allocated from the MAIN stream's pool (that is where it is used last and freed); the side stream starts behind everything the
        # main stream has queued so far, i.e. behind the previous owner of the block
        self.tables = torch.empty(capi.lib().smml_cpb_regions_bytes(), device=dev, dtype=torch.uint8)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            cpb_regions_build(w1.detach(), b1.detach(), w2.detach(), b2.detach(), w3.detach(), b3.detach(), pmax, self.tables)
            self.event = torch.cuda.Event()
            self.event.record(side)
        # a prefetch that is dropped without join() (the forward found other parameters or another square) frees the block on the main
        # stream while the side stream may still be writing it: the allocator must not hand it out before the side stream is done
        self.tables.record_stream(side)

    def matches(self, w1, b1, w2, b2, w3, b3, pmax: float) -> bool:
        return (self.pmax == float(pmax) and
                self.params == tuple(t.data_ptr() for t in (w1, b1, w2, b2, w3, b3)) + tuple(t._version for t in (w1, b1, w2, b2, w3, b3)))

    def join(self) -> torch.Tensor:
        torch.cuda.current_stream(self.tables.device).wait_event(self.event)
        return self.tables


class _DeformAttn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, vs, gq, w1, b1, w2, b2, w3, b3, heads, groups, scale, dropout_p, dropout_seed, seed_offset=None,
                compute_dtype=None, fork=None, table_pmax_fwd=None, log_distance=True, region_pmax=None, region_prefetch=None):
        ctx.fork = fork
        ctx.log_distance = bool(log_distance)
        if not log_distance and (vs.shape[-1] != 1 or table_pmax_fwd is not None):
            raise NotImplementedError("the raw-offset position transform (cpb_log_distance=False) exists for 1-D positions without the table modes")
        q, k, v, vs, gq = _c(q), _c(k), _c(v), _c(vs), _c(gq)
        w1, b1, w2, b2, w3, b3 = (_c(t) for t in (w1, b1, w2, b2, w3, b3))
        B, N, HD = q.shape
        J = k.shape[1]
        posdim = vs.shape[-1]
        if HD != heads * 64:
            raise RuntimeError("the attention kernels are built for dim_head = 64")
        if tuple(w2.shape) != (32, 32):
            raise RuntimeError("the position-bias kernels are built for a hidden width of 32 (dim = 128)")
        L = capi.lib()
        m16 = _dtype16(compute_dtype)
        out = torch.empty_like(q)
        lse = torch.empty(B, heads, N, device=q.device, dtype=torch.float32)
        need_grad = any(ctx.needs_input_grad)
        logits = masks = None
        tabfwd = table_pmax_fwd is not None           # 16-bit mode, forward bias from the table; the backward takes layer 2's decisions from a mask table
        ctx.table_pmax = table_pmax_fwd               # (or recomputes layer 2 per pair: TABLE_FORWARD_MASKS)
        ctx.export_masks = None
        regions = (region_pmax is not None and not tabfwd and posdim == 2 and log_distance and heads == groups
                   and J <= REGION_MAX_KEYS and tuple(w3.shape) == (1, 32))        # fp32-grade core or (m16) the 16-bit core, same lookup
        ctx.regions = regions
        if regions:
            if region_prefetch is not None and region_prefetch.matches(w1, b1, w2, b2, w3, b3, region_pmax):
                tables = region_prefetch.join()                 # built beside the layers in front of the attention (RegionPrefetch)
            else:
                tables = cpb_regions_build(w1, b1, w2, b2, w3, b3, region_pmax)
            rid = None
            if need_grad:
                nst = L.smml_deform_attn_nst(N)
                logits = torch.empty(B, heads, nst // 32, J, 32, device=q.device, dtype=torch.float32 if m16 is None else torch.float16)
                rid = torch.empty(B, heads, nst // 32, J, 32, device=q.device, dtype=torch.int16)      # the linear piece of every pair
            ropts = capi.deform_opts(seed_offset, region_lds_cap=REGION_LDS_CAP)
            if m16 is None:
                capi.check(L.smml_deform_attn_region_fwd_f32(capi.fptr(q), capi.fptr(k), capi.fptr(v), capi.fptr(vs), capi.fptr(gq),
                                                             capi.fptr(w1), capi.fptr(b1), capi.fptr(w2), capi.fptr(b2), capi.fptr(w3),
                                                             capi.fptr(b3), capi.ptr(tables), capi.fptr(out), capi.fptr(lse), capi.fptr(logits),
                                                             capi.ptr(rid), B, N, J, heads, float(scale), float(dropout_p), int(dropout_seed),
                                                             *TIMER.events("deform_region_fwd", B * heads * N * J), capi.stream(), ropts),
                           "deform_attn_region_fwd")
            else:
                capi.check(L.smml_deform_attn16_region_fwd(capi.fptr(q), capi.fptr(k), capi.fptr(v), capi.fptr(vs), capi.fptr(gq),
                                                           capi.fptr(w1), capi.fptr(b1), capi.fptr(w2), capi.fptr(b2), capi.fptr(w3),
                                                           capi.fptr(b3), capi.ptr(tables), capi.fptr(out), capi.fptr(lse), capi.ptr(logits),
                                                           capi.ptr(rid), B, N, J, heads, float(scale), float(dropout_p), int(dropout_seed),
                                                           m16[0], *TIMER.events("deform16_region_fwd", B * heads * N * J), capi.stream(), ropts),
                           "deform_attn16_region_fwd")
            ctx.seed_offset = seed_offset
            ctx.cfg = (heads, groups, float(scale), float(dropout_p), int(dropout_seed), m16)
            ctx.save_for_backward(q, k, v, vs, gq, w1, b1, w2, b2, w3, b3, out, lse, logits, rid, tables)
            if DECISION_TAP is not None:
                DECISION_TAP.append({"kind": "attn", "vs": vs.detach(), "gq": gq.detach(), "w1": w1.detach(), "b1": b1.detach(),
                                     "w2": w2.detach(), "b2": b2.detach(), "masks2": None, "region_ids": rid, "tables": tables,
                                     "B": B, "N": N, "J": J, "heads": heads, "groups": groups, "table_pmax": None, "log_distance": True})
            return out
        if need_grad:
            nst = L.smml_deform_attn_nst(N)
            # score-shaped tensors are stored per 32-query tile (include/smml.h): [B, H, nst / 32, J, 32] (+ the lane-half axis of the masks)
            logits = torch.empty(B, heads, nst // 32, J, 32, device=q.device, dtype=torch.float32 if m16 is None else torch.float16)   # 16-bit mode: fp16 scores in both sub-modes
            if not tabfwd:
                masks = torch.empty(B, heads, nst // 32, J, 2, 32, device=q.device, dtype=torch.int16)   # layer-2 ReLU decisions of the bias MLP
            elif DECISION_TAP is not None:            # tests: the backward writes the decisions it recomputed here
                ctx.export_masks = torch.zeros(B, heads, nst // 32, J, 2, 32, device=q.device, dtype=torch.int16)
        opts = capi.deform_opts(seed_offset, ctx.log_distance)
        if tabfwd:
            if m16 is None:
                raise ValueError("cpb_table belongs to the 16-bit compute modes: pass compute_dtype='bf16' or 'fp16'")
            points = L.smml_deform_attn_table_points(posdim)
            with torch.no_grad():
                table = cpb_table_fn(w1, b1, w2, b2, w3, b3, posdim=posdim, pmax=table_pmax_fwd, device=q.device)
            capi.check(L.smml_deform_attn_table_fwd(capi.fptr(q), capi.fptr(k), capi.fptr(v), capi.fptr(vs), capi.fptr(gq), capi.fptr(table),
                                                    capi.fptr(out), capi.fptr(lse), capi.ptr(logits), B, N, J, heads, groups, posdim, points,
                                                    float(table_pmax_fwd), float(scale), float(dropout_p), int(dropout_seed), m16[0],
                                                    *TIMER.events("deform_table_fwd", B * heads * N * J), capi.stream(), opts), "deform_attn_table_fwd")
        elif m16 is None:
            capi.check(L.smml_deform_attn_fwd_f32(capi.fptr(q), capi.fptr(k), capi.fptr(v), capi.fptr(vs), capi.fptr(gq),
                                                  capi.fptr(w1), capi.fptr(b1), capi.fptr(w2), capi.fptr(b2), capi.fptr(w3),
                                                  capi.fptr(b3), capi.fptr(out), capi.fptr(lse), capi.fptr(logits), capi.ptr(masks), B, N, J,
                                                  heads, groups, posdim, float(scale), float(dropout_p), int(dropout_seed),
                                                  *TIMER.events("deform_attn_fwd", B * heads * N * J), capi.stream(), opts),
                       "deform_attn_fwd")
        else:
            capi.check(L.smml_deform_attn16_fwd(capi.fptr(q), capi.fptr(k), capi.fptr(v), capi.fptr(vs), capi.fptr(gq),
                                                capi.fptr(w1), capi.fptr(b1), capi.fptr(w2), capi.fptr(b2), capi.fptr(w3),
                                                capi.fptr(b3), capi.fptr(out), capi.fptr(lse), capi.ptr(logits), capi.ptr(masks), B, N, J,
                                                heads, groups, posdim, float(scale), float(dropout_p), int(dropout_seed), m16[0],
                                                *TIMER.events("deform16_fwd", B * heads * N * J), capi.stream(), opts),
                       "deform_attn16_fwd")
        ctx.seed_offset = seed_offset           # a device int64 [1] owned by this call (hipGraph replays: deform_attention)
        ctx.cfg = (heads, groups, float(scale), float(dropout_p), int(dropout_seed), m16)
        ctx.save_for_backward(q, k, v, vs, gq, w1, b1, w2, b2, w3, b3, out, lse, logits, masks)
        if DECISION_TAP is not None:
            DECISION_TAP.append({"kind": "attn", "vs": vs.detach(), "gq": gq.detach(), "w1": w1.detach(), "b1": b1.detach(),
                                 "masks2": masks if not tabfwd else ctx.export_masks, "B": B, "N": N, "J": J, "heads": heads, "groups": groups,
                                 "table_pmax": table_pmax_fwd, "log_distance": ctx.log_distance})
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.regions:
            return _DeformAttn._backward_regions(ctx, dout)
        q, k, v, vs, gq, w1, b1, w2, b2, w3, b3, out, lse, logits, masks = ctx.saved_tensors
        heads, groups, scale, dropout_p, dropout_seed, m16 = ctx.cfg
        B, N, _ = q.shape
        J = k.shape[1]
        posdim = vs.shape[-1]
        L = capi.lib()
        dout = _c(dout)
        dlogits = torch.empty_like(logits) if m16 is None else torch.empty(logits.shape, device=q.device, dtype=torch.bfloat16)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        dvs = torch.empty_like(vs)
        dw1, db1, dw2, db2, dw3, db3 = (torch.empty_like(t) for t in (w1, b1, w2, b2, w3, b3))
        wsb = L.smml_deform_attn_bwd_workspace_bytes(B, N, J, heads)
        ws = torch.empty((wsb + 3) // 4, device=q.device, dtype=torch.float32)
        if m16 is None:
            capi.check(L.smml_deform_attn_bwd_f32(
                capi.fptr(q), capi.fptr(k), capi.fptr(v), capi.fptr(vs), capi.fptr(gq), capi.fptr(w1), capi.fptr(b1),
                capi.fptr(w2), capi.fptr(b2), capi.fptr(w3), capi.fptr(b3), capi.fptr(out), capi.fptr(dout), capi.fptr(lse),
                capi.fptr(logits), capi.ptr(masks), capi.fptr(dlogits), capi.fptr(dq), capi.fptr(dk), capi.fptr(dv), capi.fptr(dvs),
                capi.fptr(dw1), capi.fptr(db1), capi.fptr(dw2), capi.fptr(db2), capi.fptr(dw3), capi.fptr(db3),
                capi.fptr(ws), wsb, B, N, J, heads, groups, posdim, scale, dropout_p, dropout_seed,
                *TIMER.events("cpb_bwd", B * heads * N * J),
                capi.stream(), capi.deform_opts(ctx.seed_offset, ctx.log_distance)), "deform_attn_bwd")
        else:
            mtab = None
            if masks is None and TABLE_FORWARD_MASKS == "table":
                # table-forward call: layer-2 decisions from a mask table (include/smml.h) built from the current weights - one small launch
                cells = L.smml_cpb_mask_table_cells(posdim)
                mtab = torch.empty(cells ** posdim, device=q.device, dtype=torch.int32)
                capi.check(L.smml_cpb_mask_table(capi.fptr(w1), capi.fptr(b1), capi.fptr(w2), capi.fptr(b2), capi.ptr(mtab), posdim,
                                                 float(ctx.table_pmax), capi.stream()), "cpb_mask_table")
            capi.check(L.smml_deform_attn16_bwd(
                capi.fptr(q), capi.fptr(k), capi.fptr(v), capi.fptr(vs), capi.fptr(gq), capi.fptr(w1), capi.fptr(b1),
                capi.fptr(w2), capi.fptr(b2), capi.fptr(w3), capi.fptr(b3), capi.fptr(out), capi.fptr(dout), capi.fptr(lse),
                capi.ptr(logits), capi.ptr(masks), capi.ptr(dlogits), capi.fptr(dq), capi.fptr(dk), capi.fptr(dv), capi.fptr(dvs),
                capi.fptr(dw1), capi.fptr(db1), capi.fptr(dw2), capi.fptr(db2), capi.fptr(dw3), capi.fptr(db3),
                capi.fptr(ws), wsb, B, N, J, heads, groups, posdim, scale, dropout_p, dropout_seed, m16[0],
                *TIMER.events("cpb16_bwd", B * heads * N * J),
                capi.stream(), capi.deform_opts(ctx.seed_offset, ctx.log_distance, mtab, float(ctx.table_pmax or 0.0), ctx.export_masks)),
                "deform_attn16_bwd")
        if ctx.fork is not None and ctx.needs_input_grad[0]:
            ctx.fork.dq = dq.view(B, N, -1)          # parked for the offsets network's backward (GradFork); still returned to autograd
        return dq, dk, dv, dvs, None, dw1, db1, dw2, db2, dw3, db3, None, None, None, None, None, None, None, None, None, None, None, None

    @staticmethod
    def _backward_regions(ctx, dout):
        q, k, v, vs, gq, w1, b1, w2, b2, w3, b3, out, lse, logits, rid, tables = ctx.saved_tensors
        heads, groups, scale, dropout_p, dropout_seed, _ = ctx.cfg
        B, N, _ = q.shape
        J = k.shape[1]
        L = capi.lib()
        dout = _c(dout)
        m16 = ctx.cfg[5]
        dlogits = torch.empty_like(logits) if m16 is None else torch.empty(logits.shape, device=q.device, dtype=torch.bfloat16)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        dvs = torch.empty_like(vs)
        dw1, db1, dw2, db2, dw3, db3 = (torch.empty_like(t) for t in (w1, b1, w2, b2, w3, b3))
        wsb = L.smml_deform_attn_region_bwd_workspace_bytes(B, N, J, heads)
        ws = torch.empty(wsb, device=q.device, dtype=torch.uint8)
        if m16 is not None:
            capi.check(L.smml_deform_attn16_region_bwd(
                capi.fptr(q), capi.fptr(k), capi.fptr(v), capi.fptr(vs), capi.fptr(gq), capi.fptr(w1), capi.fptr(b1), capi.fptr(w2),
                capi.fptr(b2), capi.fptr(w3), capi.fptr(b3), capi.ptr(tables), capi.fptr(out), capi.fptr(dout), capi.fptr(lse),
                capi.ptr(logits), capi.ptr(rid), capi.ptr(dlogits), capi.fptr(dq), capi.fptr(dk), capi.fptr(dv), capi.fptr(dvs),
                capi.fptr(dw1), capi.fptr(db1), capi.fptr(dw2), capi.fptr(db2), capi.fptr(dw3), capi.fptr(db3), capi.ptr(ws), wsb,
                B, N, J, heads, scale, dropout_p, dropout_seed, m16[0], *TIMER.events("cpb16_region_bwd", B * heads * N * J), capi.stream(),
                capi.deform_opts(ctx.seed_offset, region_lds_cap=REGION_LDS_CAP)), "deform_attn16_region_bwd")
            if ctx.fork is not None and ctx.needs_input_grad[0]:
                ctx.fork.dq = dq.view(B, N, -1)
            return dq, dk, dv, dvs, None, dw1, db1, dw2, db2, dw3, db3, None, None, None, None, None, None, None, None, None, None, None, None
        capi.check(L.smml_deform_attn_region_bwd_f32(
            capi.fptr(q), capi.fptr(k), capi.fptr(v), capi.fptr(vs), capi.fptr(gq), capi.fptr(w1), capi.fptr(b1), capi.fptr(w2),
            capi.fptr(b2), capi.fptr(w3), capi.fptr(b3), capi.ptr(tables), capi.fptr(out), capi.fptr(dout), capi.fptr(lse),
            capi.fptr(logits), capi.ptr(rid), capi.fptr(dlogits), capi.fptr(dq), capi.fptr(dk), capi.fptr(dv), capi.fptr(dvs),
            capi.fptr(dw1), capi.fptr(db1), capi.fptr(dw2), capi.fptr(db2), capi.fptr(dw3), capi.fptr(db3), capi.ptr(ws), wsb,
            B, N, J, heads, scale, dropout_p, dropout_seed, *TIMER.events("cpb_region_bwd", B * heads * N * J), capi.stream(),
            capi.deform_opts(ctx.seed_offset, region_lds_cap=REGION_LDS_CAP)), "deform_attn_region_bwd")
        if ctx.fork is not None and ctx.needs_input_grad[0]:
            ctx.fork.dq = dq.view(B, N, -1)
        return dq, dk, dv, dvs, None, dw1, db1, dw2, db2, dw3, db3, None, None, None, None, None, None, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------
# table mode of the 16-bit core: the position-bias MLP evaluated once per call on a grid, interpolated per pair (include/smml.h)
# ------------------------------------------------------------------------------------------------
_TABLE_POINTS = {}


def _table_points(points: int, posdim: int, pmax: float, device) -> torch.Tensor:
    """[points^posdim, posdim] grid of signed-log offsets, point (i0, i1) in row i1 * points + i0 (a constant per (pmax, device))."""
    key = (points, posdim, float(pmax), str(device))
    g = _TABLE_POINTS.get(key)
    if g is None:
        ax = (-pmax + torch.arange(points, dtype=torch.float64) * (2.0 * pmax / (points - 1))).float()
        if posdim == 2:
            g = torch.stack((ax.view(1, points).expand(points, points), ax.view(points, 1).expand(points, points)), dim=-1).reshape(-1, 2)
        else:
            g = ax.view(points, 1)
        if len(_TABLE_POINTS) > 16:
            _TABLE_POINTS.clear()
        g = _TABLE_POINTS[key] = g.contiguous().to(device)
    return g


def cpb_table_fn(w1, b1, w2, b2, w3, b3, *, posdim: int, pmax: float, device):
    """The position-bias MLP (DeformableAttention2D.py:129-152 / DeformableAttention1D.py:69-98, already log-transformed input) on the
    table grid: [heads // groups, points^posdim], fp32, three exact-fp32 GEMM launches; autograd carries d table back to the six tensors."""
    points = capi.lib().smml_deform_attn_table_points(posdim)
    pts = _table_points(points, posdim, pmax, device)
    h = linear(pts, w1, b1, act=ACT_RELU)
    h = linear(h, w2, b2, act=ACT_RELU)
    t = linear(h, w3, b3)                                   # [cells, o]
    return t.reshape(1, -1) if t.shape[1] == 1 else t.t().contiguous()


class _DeformAttnTable(torch.autograd.Function):
    @staticmethod
    def forward(ctx, q, k, v, vs, gq, table, heads, groups, scale, dropout_p, dropout_seed, seed_offset, compute_dtype, fork, pmax, grid):
        ctx.fork = fork
        ctx.grid = tuple(int(x) for x in grid) if grid else (0, 0)
        q, k, v, vs, gq, table = _c(q), _c(k), _c(v), _c(vs), _c(gq), _c(table)
        B, N, HD = q.shape
        J = k.shape[1]
        posdim = vs.shape[-1]
        if HD != heads * 64:
            raise RuntimeError("the attention kernels are built for dim_head = 64")
        L = capi.lib()
        m16 = _dtype16(compute_dtype)
        if m16 is None:
            raise ValueError("the table mode belongs to the 16-bit compute modes: pass compute_dtype='bf16' or 'fp16'")
        points = L.smml_deform_attn_table_points(posdim)
        if tuple(table.shape) != (heads // groups, points ** posdim):
            raise RuntimeError(f"table must be [{heads // groups}, {points ** posdim}] (got {tuple(table.shape)})")
        out = torch.empty_like(q)
        lse = torch.empty(B, heads, N, device=q.device, dtype=torch.float32)
        logits = None
        if any(ctx.needs_input_grad):
            nst = L.smml_deform_attn_nst(N)
            logits = torch.empty(B, heads, nst // 32, J, 32, device=q.device, dtype=torch.float16)
        capi.check(L.smml_deform_attn_table_fwd(capi.fptr(q), capi.fptr(k), capi.fptr(v), capi.fptr(vs), capi.fptr(gq), capi.fptr(table),
                                                capi.fptr(out), capi.fptr(lse), capi.ptr(logits), B, N, J, heads, groups, posdim, points,
                                                float(pmax), float(scale), float(dropout_p), int(dropout_seed), m16[0],
                                                *TIMER.events("deform_table_fwd", B * heads * N * J), capi.stream(), capi.deform_opts(seed_offset)),
                   "deform_attn_table_fwd")
        ctx.seed_offset = seed_offset
        ctx.cfg = (heads, groups, float(scale), float(dropout_p), int(dropout_seed), m16, points, float(pmax))
        ctx.save_for_backward(q, k, v, vs, gq, table, out, lse, logits)
        if DECISION_TAP is not None:       # no per-pair ReLU decisions in this mode: the MLP runs on grid points only
            DECISION_TAP.append({"kind": "attn", "vs": vs.detach(), "gq": gq.detach(), "masks2": None, "B": B, "N": N, "J": J,
                                 "heads": heads, "groups": groups, "table_pmax": float(pmax)})
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, vs, gq, table, out, lse, logits = ctx.saved_tensors
        heads, groups, scale, dropout_p, dropout_seed, m16, points, pmax = ctx.cfg
        B, N, _ = q.shape
        J = k.shape[1]
        posdim = vs.shape[-1]
        L = capi.lib()
        dout = _c(dout)
        dlogits = torch.empty(logits.shape, device=q.device, dtype=torch.bfloat16)
        dq, dk, dv, dvs, dtable = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v), torch.empty_like(vs), torch.empty_like(table)
        wsb = L.smml_deform_attn_table_bwd_workspace_bytes(B, N, J, heads, posdim)
        ws = torch.empty((wsb + 3) // 4, device=q.device, dtype=torch.float32)
        capi.check(L.smml_deform_attn_table_bwd(
            capi.fptr(q), capi.fptr(k), capi.fptr(v), capi.fptr(vs), capi.fptr(gq), capi.fptr(table), capi.fptr(out), capi.fptr(dout),
            capi.fptr(lse), capi.ptr(logits), capi.ptr(dlogits), capi.fptr(dq), capi.fptr(dk), capi.fptr(dv), capi.fptr(dvs),
            capi.fptr(dtable), capi.fptr(ws), wsb, B, N, J, heads, groups, posdim, points, pmax, ctx.grid[0], ctx.grid[1], scale, dropout_p,
            dropout_seed, m16[0],
            *TIMER.events("cpb_table_bwd", B * heads * N * J), capi.stream(), capi.deform_opts(ctx.seed_offset)), "deform_attn_table_bwd")
        if ctx.fork is not None and ctx.needs_input_grad[0]:
            ctx.fork.dq = dq.view(B, N, -1)
        return dq, dk, dv, dvs, None, dtable, None, None, None, None, None, None, None, None, None, None


def table_pmax(gq_bound: float, vs_bound: float) -> float:
    """Half-width of the table in signed-log units for |gq| <= gq_bound, |vs| <= vs_bound (positions beyond it take the edge value)."""
    return float(__import__("math").log1p(gq_bound + vs_bound) * 1.0001)


_REPLAY_COUNTER = {}         # device index -> int64 [1]: bumped once per dropout call inside a captured graph


def graph_seed_offset(device, allocate_only: bool = False):
    """The per-call seed offset of a dropout launch that is being captured in a hipGraph: a device-resident counter is bumped and
    copied (both operations are part of the graph), so every replay - and every call within a replay - draws a new mask while the
    forward and the backward of one call read the same value.  The counter itself must exist before the capture starts (it would
    otherwise live in the graph's private pool and be re-zeroed by every replay): any eager dropout call on the device creates it."""
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    ctr = _REPLAY_COUNTER.get(key)
    capturing = torch.cuda.is_current_stream_capturing()
    if ctr is None:
        if capturing:
            raise RuntimeError("run one eager training step before capturing it in a graph (the dropout replay counter is created "
                               "by the first eager call)")
        ctr = _REPLAY_COUNTER[key] = torch.zeros(1, device=dev, dtype=torch.int64)
    if allocate_only or not capturing:
        return None
    ctr.add_(1)
    return ctr.clone()


def deform_attention(q, k, v, vs, gq, w1, b1, w2, b2, w3, b3, *, heads: int, groups: int, scale: float,
                     dropout_p: float = 0.0, dropout_seed: int = 0, dropout_seed_offset=None, compute_dtype=None, fork=None,
                     cpb_table: bool = False, cpb_table_pmax=None, cpb_table_grid=None, log_distance: bool = True,
                     cpb_regions=None, cpb_region_pmax=None, cpb_region_prefetch=None):
    """dropout(softmax(scale q k^T + CPB(gq - vs))) v.  q [B, N, H*64], k/v [B, J, H*64], vs [(B G), J, P], gq [N, P].
    dropout_p > 0 applies nn.Dropout semantics to the probabilities with a counter-based mask from dropout_seed
    (+ the value of the device tensor dropout_seed_offset at run time, see graph_seed_offset).
    compute_dtype None: fp32-grade split products (csrc/deform_attn.hip); 'bf16' / 'fp16': the 16-bit compute mode
    (csrc/deform_attn16.hip: single-term 16-bit MFMA operands, 16-bit score storage; inputs, outputs and gradients stay fp32).
    log_distance False (1-D positions only): the bias MLP reads the raw offset gq - vs (DeformableAttention1D.py:92, cpb_log_distance=False).
    cpb_table (with a 16-bit compute_dtype): True / 'full' - the position-bias MLP is evaluated once on a grid and interpolated per pair,
    forward and backward (include/smml.h, "table mode": approximate parameter gradients); 'forward' - only the forward takes its bias from the
    table (|error| <= ~1e-3 of the bias range, below the 16-bit operand rounding), the backward differentiates the per-pair MLP itself, recomputing
    layer 2 (parity-grade gradients at the 16-bit mode's tolerances); cpb_table_pmax = half-width of the grid in signed-log units (table_pmax(); None: taken from the data, one host sync);
    cpb_table_grid = (rows, cols) asserts that the queries sit on a regular grid, gq[y * cols + x] = (X[x], Y[y]) - the backward then
    builds d table on the matrix pipe instead of with LDS atomics."""
    if cpb_table not in (False, True, None, "forward", "full"):
        raise ValueError("cpb_table must be False, True / 'full' or 'forward'")
    if cpb_table and cpb_table_pmax is None:
        cpb_table_pmax = table_pmax(float(gq.detach().abs().max()), float(vs.detach().abs().max()))
    if cpb_table and not log_distance:
        raise NotImplementedError("the table modes are built for the signed-log position transform only")
    if cpb_table == "forward":
        return _DeformAttn.apply(q, k, v, vs, gq, w1, b1, w2, b2, w3, b3, heads, groups, scale, dropout_p, dropout_seed,
                                 dropout_seed_offset, compute_dtype, fork, cpb_table_pmax)
    if cpb_table:
        posdim = vs.shape[-1]
        table = cpb_table_fn(w1, b1, w2, b2, w3, b3, posdim=posdim, pmax=cpb_table_pmax, device=q.device)
        return _DeformAttnTable.apply(q, k, v, vs, gq, table, heads, groups, scale, dropout_p, dropout_seed, dropout_seed_offset,
                                      compute_dtype, fork, cpb_table_pmax, cpb_table_grid)
    # fp32-grade path: the position bias per linear region of its MLP wherever that applies (see CPB_REGIONS); cpb_regions False / True
    # overrides the default, cpb_region_pmax = half-width of the tabulated square in signed-log units (None: from the data, one host sync)
    use_regions = CPB_REGIONS if cpb_regions is None else bool(cpb_regions)
    region_pmax = None
    if (use_regions and log_distance and vs.shape[-1] == 2 and heads == groups and k.shape[1] <= REGION_MAX_KEYS
            and tuple(w3.shape) == (1, 32)):
        if cpb_region_pmax is not None:
            region_pmax = cpb_region_pmax
        elif not torch.cuda.is_current_stream_capturing():         # from the data: a host sync, impossible inside a hipGraph capture - such a
            region_pmax = table_pmax(float(gq.detach().abs().max()), float(vs.detach().abs().max()))   # call keeps the per-pair kernels
    return _DeformAttn.apply(q, k, v, vs, gq, w1, b1, w2, b2, w3, b3, heads, groups, scale, dropout_p, dropout_seed,
                             dropout_seed_offset, compute_dtype, fork, None, log_distance, region_pmax, cpb_region_prefetch)


def deform_attention_dropout_mask(B: int, N: int, J: int, H: int, dropout_p: float, dropout_seed: int, device, seed_offset=None):
    """The keep mask (0 / 1) [B, H, N, J] a launch with this (p, seed[, device-resident replay offset]) applies; tests only."""
    mask = torch.empty(B, H, N, J, device=device, dtype=torch.float32)
    L = capi.lib()
    capi.check(L.smml_deform_attn_dropout_mask_f32(capi.fptr(mask), B, N, J, H, float(dropout_p), int(dropout_seed),
                                                   capi.stream(), capi.deform_opts(seed_offset)), "dropout_mask")
    return mask


# ------------------------------------------------------------------------------------------------
# token mean / token tile / Gram matrices (Pooler, omic tiling, BatchLoss)
# ------------------------------------------------------------------------------------------------
class _TokenMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        ctx.n = x.shape[1]
        return colsum(x, 1.0 / x.shape[1])

    @staticmethod
    def backward(ctx, dy):
        return (dy / ctx.n).unsqueeze(1).expand(-1, ctx.n, -1).contiguous()


def token_mean(x):
    """x [B, n, C] -> mean over tokens [B, C]."""
    return _TokenMean.apply(x)


class _TileTokens(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, n):
        return v.unsqueeze(1).repeat(1, n, 1)

    @staticmethod
    def backward(ctx, dy):
        return colsum(_c(dy)), None


def tile_tokens(v, n: int):
    """v [B, C] -> [B, n, C] (values of v.unsqueeze(1).repeat(1, n, 1)); backward is a column sum.
    The result carries the un-tiled vector as `_smml_compact`: BatchLoss uses it to gather [B, C] instead of the
    n-times larger tile across ranks (the Gram of a tile is n x the Gram of the vector, and the row normalisation
    that follows cancels the factor)."""
    out = _TileTokens.apply(v, n)
    out._smml_compact = v
    return out


class _Gram(torch.autograd.Function):
    """x [nb, R, K] -> x x^T [nb, R, R]: skinny, HBM-bound (K up to N*C = 1.28 M): split-K over the long axis."""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        nb, R, K = x.shape
        g = torch.zeros(nb, R, R, device=x.device, dtype=torch.float32)
        splitk = int(max(1, min((K + 255) // 256, 1024 // max(nb, 1), 65535 // nb)))   # R x R outputs: the K axis is all there is to spread
        _gemm(x, x, g, M=R, N=R, K=K, sam=K, sak=1, sbk=1, sbn=K, ldc=R, nb0=nb, sa0=R * K, sb0=R * K, sc0=R * R,
              splitk=splitk)
        ctx.save_for_backward(x)
        return g

    @staticmethod
    def backward(ctx, dg):
        (x,) = ctx.saved_tensors
        nb, R, K = x.shape
        s = _c(dg + dg.transpose(1, 2))
        dx = torch.empty_like(x)
        _gemm(s, x, dx, M=R, N=K, K=R, sam=R, sak=1, sbk=K, sbn=1, ldc=K, nb0=nb, sa0=R * R, sb0=R * K, sc0=R * K)
        return dx


def gram(x):
    return _Gram.apply(x)


class _BatchLossTail(torch.autograd.Function):
    """(go / ||go||_row - mean_g gv[g] / ||gv[g]||_row)^2 / n on [R, R] matrices (utils/loss.py:26-40): one launch per direction."""

    @staticmethod
    def forward(ctx, go, gv, n_total):
        go, gv = _c(go), _c(gv)
        R, nv = go.shape[-1], gv.shape[0]
        out = torch.empty(R, R, device=go.device, dtype=torch.float32)
        capi.check(capi.lib().smml_batchloss_tail_f32(capi.fptr(go), capi.fptr(gv), None, capi.fptr(out), None, None, R, nv, float(n_total),
                                                      capi.stream()), "batchloss_tail")
        ctx.n_total = float(n_total)
        ctx.save_for_backward(go, gv)
        return out

    @staticmethod
    def backward(ctx, dl):
        go, gv = ctx.saved_tensors
        R, nv = go.shape[-1], gv.shape[0]
        dgo, dgv = torch.empty_like(go), torch.empty_like(gv)
        capi.check(capi.lib().smml_batchloss_tail_f32(capi.fptr(go), capi.fptr(gv), capi.fptr(_c(dl)), None, capi.fptr(dgo), capi.fptr(dgv), R, nv,
                                                      ctx.n_total, capi.stream()), "batchloss_tail_bwd")
        return dgo, dgv, None


def batchloss_tail(go, gv, n_total):
    """go [R, R] (Gram of the omic rows), gv [nv, R, R] (Grams of the vgrid groups) -> the BatchLoss matrix [R, R]."""
    return _BatchLossTail.apply(go, gv, n_total)


# ------------------------------------------------------------------------------------------------
# generic batched product on the matrix cores: C = alpha * op(A) @ op(B) + beta * R
# (Nystrom sims / landmark products / pinv iteration, co-attention)
# ------------------------------------------------------------------------------------------------
def _op_view(T, trans):
    """(rows, cols, row stride, col stride, batch strides) of op(T) for a contiguous [t0, t1, r, c] tensor."""
    t0, t1, r, c = T.shape
    s0 = t1 * r * c if t0 > 1 else 0
    s1 = r * c if t1 > 1 else 0
    return (c, r, 1, c, s0, s1) if trans else (r, c, c, 1, s0, s1)


class _MatMul(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, B, R, ta, tb, alpha, beta, merged):
        A, B = _c(A), _c(B)
        M, K, sam, sak, sa0, sa1 = _op_view(A, ta)
        K2, N, sbk, sbn, sb0, sb1 = _op_view(B, tb)
        if K != K2:
            raise RuntimeError(f"matmul4: inner dimensions differ ({K} vs {K2})")
        nb0, nb1 = max(A.shape[0], B.shape[0]), max(A.shape[1], B.shape[1])
        if merged:
            Cm = torch.empty(nb0, M, nb1 * N, device=A.device, dtype=torch.float32)
            ldc, sc0, sc1 = nb1 * N, M * nb1 * N, N
        else:
            Cm = torch.empty(nb0, nb1, M, N, device=A.device, dtype=torch.float32)
            ldc, sc0, sc1 = N, nb1 * M * N, M * N
        Rc = None
        if R is not None:
            Rc = _c(R)
            if Rc.shape != Cm.shape:
                raise RuntimeError("matmul4: residual must have the output's shape")
        # few output tiles and a long reduction (attn3 @ v: 256 x 64 outputs over n' keys): the K range is cut into S slices
        # that run as an extra batch dimension into a partial-sum buffer, added up in a fixed order afterwards (no atomics:
        # identical inputs give bit-identical outputs)
        nbt = nb0 * nb1
        full = (A.shape[0] == nb0 and A.shape[1] == nb1 and B.shape[0] == nb0 and B.shape[1] == nb1)
        S = _splitk_for(M, N, K, nbt) if (Rc is None and not merged and full and K >= 1024) else 1
        while S > 1 and (K % S or (K // S) % 4):
            S -= 1
        if S > 1:
            Kc = K // S
            part = torch.empty(S, nb0, nb1, M, N, device=A.device, dtype=torch.float32)
            _gemm(A, B, part, M=M, N=N, K=Kc, sam=sam, sak=sak, sbk=sbk, sbn=sbn, ldc=N, nb0=nbt, nb1=S,
                  sa0=A.shape[2] * A.shape[3], sa1=Kc * sak, sb0=B.shape[2] * B.shape[3], sb1=Kc * sbk,
                  sc0=M * N, sc1=nbt * M * N, alpha=alpha)
            torch.sum(part, dim=0, out=Cm)
        else:
            _gemm(A, B, Cm, M=M, N=N, K=K, sam=sam, sak=sak, sbk=sbk, sbn=sbn, ldc=ldc, residual=Rc, ldr=ldc, nb0=nb0,
                  nb1=nb1, sa0=sa0, sa1=sa1, sb0=sb0, sb1=sb1, sc0=sc0, sc1=sc1, alpha=alpha, beta=beta)
        ctx.cfg = (ta, tb, float(alpha), float(beta), merged, nb0, nb1, M, N, K, R is not None)
        ctx.save_for_backward(A, B)
        return Cm

    @staticmethod
    def backward(ctx, dC):
        A, B = ctx.saved_tensors
        ta, tb, alpha, beta, merged, nb0, nb1, M, N, K, has_r = ctx.cfg
        dC = _c(dC)
        if merged:
            ldc, sc0, sc1 = nb1 * N, M * nb1 * N, N
        else:
            ldc, sc0, sc1 = N, nb1 * M * N, M * N
        _, _, a_sr, a_sc, sa0, sa1 = _op_view(A, ta)       # strides of op(A)[m, k]
        _, _, b_sr, b_sc, sb0, sb1 = _op_view(B, tb)       # strides of op(B)[k, n]
        dA = dB = dR = None

        def run(like, Mx, Nx, Kx, Xa, xa, Xb, xb, ld_out, bcast):
            nb = nb0 * nb1
            splitk = _splitk_for(Mx, Nx, Kx, nb)
            acc = 1 if (bcast or splitk > 1) else 0
            out = torch.zeros_like(like) if acc else torch.empty_like(like)   # zero-fill only what atomics accumulate into
            _gemm(Xa, Xb, out, M=Mx, N=Nx, K=Kx, sam=xa[0], sak=xa[1], sbk=xb[0], sbn=xb[1], ldc=ld_out, nb0=nb0, nb1=nb1,
                  sa0=xa[2], sa1=xa[3], sb0=xb[2], sb1=xb[3],
                  sc0=(out.shape[1] * out.shape[2] * out.shape[3] if out.shape[0] > 1 else 0),
                  sc1=(out.shape[2] * out.shape[3] if out.shape[1] > 1 else 0), alpha=alpha, splitk=splitk,
                  accumulate=acc)
            return out

        if ctx.needs_input_grad[0]:
            bcast = (A.shape[0] < nb0) or (A.shape[1] < nb1)
            if not ta:   # dA[m, k] = sum_n dC[m, n] opB[k, n]
                dA = run(A, M, K, N, dC, (ldc, 1, sc0, sc1), B, (b_sc, b_sr, sb0, sb1), A.shape[3], bcast)
            else:        # A stored [K, M]: dA[k, m] = sum_n opB[k, n] dC[m, n]
                dA = run(A, K, M, N, B, (b_sr, b_sc, sb0, sb1), dC, (1, ldc, sc0, sc1), A.shape[3], bcast)
        if ctx.needs_input_grad[1]:
            bcast = (B.shape[0] < nb0) or (B.shape[1] < nb1)
            if not tb:   # B stored [K, N]: dB[k, n] = sum_m opA[m, k] dC[m, n]
                dB = run(B, K, N, M, A, (a_sc, a_sr, sa0, sa1), dC, (ldc, 1, sc0, sc1), B.shape[3], bcast)
            else:        # B stored [N, K]: dB[n, k] = sum_m dC[m, n] opA[m, k]
                dB = run(B, N, K, M, dC, (1, ldc, sc0, sc1), A, (a_sr, a_sc, sa0, sa1), B.shape[3], bcast)
        if has_r and ctx.needs_input_grad[2]:
            dR = dC * beta if beta != 1.0 else dC
        return dA, dB, dR, None, None, None, None, None


def matmul4(A, B, R=None, *, ta=False, tb=False, alpha=1.0, beta=1.0, merged=False):
    """alpha * op(A) @ op(B) + beta * R for [nb0, nb1, rows, cols] tensors (size-1 batch dims broadcast);
    merged=True lays the result out as [nb0, M, nb1 * N] (heads merged)."""
    return _MatMul.apply(A, B, R, ta, tb, alpha, beta, merged)


class gemm_precision:
    """with gemm_precision(3): ...  - every tiled product launched inside runs with single-term bf16 operands (fp32 storage and
    accumulation); 2 = three-term split (fp32-grade); 0 restores the automatic choice.  Kernel selection happens on the host at
    launch time, so the switch covers exactly the launches issued inside the block (linear(..., prec=mode) re-enters it in its
    backward, so that the gradient products use the same operand precision as the forward one)."""

    def __init__(self, mode: int):
        self.mode = int(mode)

    def __enter__(self):
        L = capi.lib()
        self.prev = L.smml_gemm_get_mode()
        L.smml_gemm_set_mode(self.mode)
        return self

    def __exit__(self, *a):
        capi.lib().smml_gemm_set_mode(self.prev)


class _Attention16(torch.autograd.Function):
    """softmax(scale q k^T) v (+ residual) on the 16-bit matrix pipe, fp32 storage (csrc/attn16.hip)."""

    @staticmethod
    def forward(ctx, q, k, v, residual, scale, fp16, merged):
        q, k, v = _c(q), _c(k), _c(v)
        Bn, H, Lq, D = q.shape
        Lk = k.shape[2]
        if D != 64 or k.shape != (Bn, H, Lk, D) or v.shape != (Bn, H, Lk, D):
            raise RuntimeError("attention16: q [B, h, Lq, 64], k / v [B, h, Lk, 64] expected")
        if residual is not None:
            if not merged or tuple(residual.shape) != (Bn, Lq, H * D):
                raise RuntimeError("attention16: the residual is added in the heads-merged layout [B, Lq, h * 64]")
            out = _c(residual).clone()
        else:
            out = torch.empty((Bn, Lq, H * D) if merged else (Bn, H, Lq, D), device=q.device, dtype=torch.float32)
        lse2 = torch.empty(Bn * H, Lq, device=q.device, dtype=torch.float32)
        L = capi.lib()
        wsb = L.smml_attn16_fwd_workspace_bytes(Bn * H, Lq, Lk)
        ws = torch.empty((wsb + 3) // 4, device=q.device, dtype=torch.float32) if wsb else None
        capi.check(L.smml_attn16_fwd_f32(capi.fptr(q), capi.fptr(k), capi.fptr(v), capi.fptr(out), capi.fptr(lse2), capi.fptr(ws), wsb,
                                         Bn * H, Lq, Lk, D, float(scale), int(bool(fp16)), H if merged else 0,
                                         1 if residual is not None else 0, capi.stream()), "attn16_fwd")
        ctx.cfg = (float(scale), int(bool(fp16)), H if merged else 0, residual is not None)
        ctx.save_for_backward(q, k, v, out, lse2, _c(residual) if residual is not None else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, v, out, lse2, residual = ctx.saved_tensors
        scale, fp16, merged, has_res = ctx.cfg
        Bn, H, Lq, D = q.shape
        Lk = k.shape[2]
        dout = _c(dout)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        L = capi.lib()
        wsb = L.smml_attn16_bwd_workspace_bytes(Bn * H, Lq, Lk)
        ws = torch.empty((wsb + 3) // 4, device=q.device, dtype=torch.float32)
        capi.check(L.smml_attn16_bwd_f32(capi.fptr(q), capi.fptr(k), capi.fptr(v), capi.fptr(out), capi.fptr(residual), capi.fptr(dout),
                                         capi.fptr(lse2), capi.fptr(dq), capi.fptr(dk), capi.fptr(dv), capi.fptr(ws), wsb, Bn * H, Lq, Lk,
                                         D, scale, fp16, merged, capi.stream()), "attn16_bwd")
        return dq, dk, dv, (dout if has_res else None), None, None, None


def attention16(q, k, v, *, scale: float, fp16: bool = False, merged: bool = False, residual=None):
    """softmax(scale q k^T) v for q [B, h, Lq, 64], k / v [B, h, Lk, 64] with bf16 (or fp16) matrix-pipe operands and fp32
    accumulation; no [Lq, Lk] matrix is materialised.  merged=True returns [B, Lq, h * 64]; `residual` (that layout) is added."""
    return _Attention16.apply(q, k, v, residual, scale, fp16, merged)


class _SoftmaxRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        L = x.shape[-1]
        y = torch.empty_like(x)
        capi.check(capi.lib().smml_softmax_fwd_f32(capi.fptr(x), capi.fptr(y), x.numel() // L, L, capi.stream()), "softmax_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        L = y.shape[-1]
        dx = torch.empty_like(y)
        capi.check(capi.lib().smml_softmax_bwd_f32(capi.fptr(y), capi.fptr(_c(dy)), capi.fptr(dx), y.numel() // L, L,
                                                   capi.stream()), "softmax_bwd")
        return dx


def softmax_rows(x):
    return _SoftmaxRows.apply(x)


class _SegmentMean(torch.autograd.Function):
    """x [..., n, d] -> mean over l consecutive tokens -> [..., n / l, d]."""

    @staticmethod
    def forward(ctx, x, l):
        x = _c(x)
        n, d = x.shape[-2:]
        lead = x.numel() // (n * d)
        m = n // l
        ctx.l = l
        ctx.shape = x.shape
        return colsum(x.view(lead * m, l, d), 1.0 / l).view(*x.shape[:-2], m, d)

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        d = dy.shape[-1]
        nb = dy.numel() // d
        dx = torch.empty(ctx.shape, device=dy.device, dtype=torch.float32)
        capi.check(capi.lib().smml_tile_rows_f32(capi.fptr(dy), capi.fptr(dx), nb, ctx.l, d, 1.0 / ctx.l, capi.stream()),
                   "tile_rows")
        return dx, None


def segment_mean(x, l: int):
    return _SegmentMean.apply(x, l)


class _ResConv(torch.autograd.Function):
    """Depthwise convolution along the tokens, per head: v [B, h, n, d], w [h, KW] -> [B, n, h*d] (heads merged)."""

    @staticmethod
    def forward(ctx, v, w):
        v = _c(v)
        w2 = _c(w).reshape(w.shape[0], -1)
        B, H, n, D = v.shape
        out = torch.empty(B, n, H * D, device=v.device, dtype=torch.float32)
        capi.check(capi.lib().smml_resconv_fwd_f32(capi.fptr(v), capi.fptr(w2), capi.fptr(out), B, H, n, D, w2.shape[1],
                                                   capi.stream()), "resconv_fwd")
        ctx.wshape = w.shape
        ctx.save_for_backward(v, w2)
        return out

    @staticmethod
    def backward(ctx, dout):
        v, w2 = ctx.saved_tensors
        B, H, n, D = v.shape
        dv = torch.empty_like(v)
        dw = _zeros_like(w2)
        capi.check(capi.lib().smml_resconv_bwd_f32(capi.fptr(_c(dout)), capi.fptr(v), capi.fptr(w2), capi.fptr(dv),
                                                   capi.fptr(dw), B, H, n, D, w2.shape[1], capi.stream()), "resconv_bwd")
        return dv, dw.reshape(ctx.wshape)


def resconv(v, w):
    return _ResConv.apply(v, w)


class _DwConv7(torch.autograd.Function):
    """Depthwise 7x7 convolution (padding 3) on channel-last maps x [B, H, W, C] with weights wm [C, 49], bias [C]."""

    @staticmethod
    def forward(ctx, x, wm, bias):
        x, wm, bias = _c(x), _c(wm), _c(bias)
        B, H, W, Cc = x.shape
        y = torch.empty_like(x)
        capi.check(capi.lib().smml_dwconv7_fwd_f32(capi.fptr(x), capi.fptr(wm), capi.fptr(bias), capi.fptr(y), B, H, W, Cc, 0,
                                                   capi.stream()), "dwconv7_fwd")
        ctx.save_for_backward(x, wm)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wm = ctx.saved_tensors
        B, H, W, Cc = x.shape
        dy = _c(dy)
        dx = torch.empty_like(x)
        L = capi.lib()
        capi.check(L.smml_dwconv7_fwd_f32(capi.fptr(dy), capi.fptr(wm), None, capi.fptr(dx), B, H, W, Cc, 1, capi.stream()),
                   "dwconv7_bwd_data")
        dwm = _zeros_like(wm)
        db = _ZEROS.zeros((Cc,), x.device)
        capi.check(L.smml_dwconv7_bwd_weight_f32(capi.fptr(x), capi.fptr(dy), capi.fptr(dwm), capi.fptr(db), B, H, W, Cc,
                                                 capi.stream()), "dwconv7_bwd_weight")
        return dx, dwm, db


def dwconv7(x, wm, bias):
    return _DwConv7.apply(x, wm, bias)


class _OrthLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, P, Ph, G, Gh, gamma):
        P, Ph, G, Gh = _c(P), _c(Ph), _c(G), _c(Gh)
        B, D = P.shape
        loss = torch.empty(B, device=P.device, dtype=torch.float32)
        capi.check(capi.lib().smml_orth_loss_f32(capi.fptr(P), capi.fptr(Ph), capi.fptr(G), capi.fptr(Gh), None, capi.fptr(loss),
                                                 None, None, None, None, B, D, float(gamma), capi.stream()), "orth_loss")
        ctx.gamma = float(gamma)
        ctx.save_for_backward(P, Ph, G, Gh)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        P, Ph, G, Gh = ctx.saved_tensors
        B, D = P.shape
        dP, dPh, dG, dGh = (torch.empty_like(t) for t in (P, Ph, G, Gh))
        capi.check(capi.lib().smml_orth_loss_f32(capi.fptr(P), capi.fptr(Ph), capi.fptr(G), capi.fptr(Gh), capi.fptr(_c(dloss)),
                                                 None, capi.fptr(dP), capi.fptr(dPh), capi.fptr(dG), capi.fptr(dGh), B, D,
                                                 ctx.gamma, capi.stream()), "orth_loss_bwd")
        return dP, dPh, dG, dGh, None


def orthogonal_loss(P, Ph, G, Gh, gamma: float = 0.5):
    return _OrthLoss.apply(P, Ph, G, Gh, gamma)

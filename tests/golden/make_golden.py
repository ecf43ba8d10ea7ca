#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/*.npz from the REFERENCE implementation.

Runs only in the build container: it imports /root/reference read-only (with empty stubs for the
unrelated third-party packages the reference imports at module scope but which are not installed:
lifelines, sksurv, imblearn, torchvision, wandb, h5py, cv2, skimage; and with the pip package
`nystrom_attention` aliased to the reference's in-tree copy models/NystromAttention.py).
Weights and inputs come from the portable generator in the package (synth.py), so the fixtures
hold OUTPUTS only (values, selected gradients, vgrid, integer sampling corners) plus checksums.

Usage:  python tests/golden/make_golden.py            (rewrites tests/golden/*.npz)
The fixtures travel to the GPU box; this script and the reference do not need to.
"""
from __future__ import annotations

import argparse
import importlib
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

synth = importlib.import_module("subspace-multimodal-learning_amd.synth")
MAX_KEEP = 4096


def summarize(t: torch.Tensor, t64: torch.Tensor = None) -> dict:
    """Strided subset (<= MAX_KEEP values) + float64 checksums of the full tensor.  `t64` (the same quantity
    from an fp64 run of the oracle) adds `noise`: how far fp32 arithmetic alone moves this tensor, relative to
    its scale.  Tensors the reference's own fp32 arithmetic does not determine to 2e-5 (ReLU-gated sums over 1e6+ pairs
    with cancellation) also carry the fp64 values of the subset (`sub64`), so that the tests can hold the HIP result to
    1.5 x the reference's own distance from fp64 (tests/helpers.py)."""
    f = t.detach().to(torch.float64).flatten()
    step = max(1, -(-f.numel() // MAX_KEEP))
    extra = {}
    if t64 is not None:
        extra["noise"] = np.float64(rel_err(t, t64))
        if extra["noise"] > 2e-5:
            extra["sub64"] = t64.detach().to(torch.float64).flatten()[::step].numpy()
    return {**extra,
        "sub": f[::step].to(torch.float32).numpy(),
        "step": np.int64(step),
        "shape": np.asarray(t.shape, dtype=np.int64),
        "sum": np.float64(f.sum()),
        "l1": np.float64(f.abs().sum()),
        "l2": np.float64(f.pow(2).sum().sqrt()),
    }


class natural_scales:
    """Collects sum |d bias| per position-bias output-bias tensor during the fp64 oracle run (oracle.deform.GRAD_PROBE): the
    natural scale of d rel_pos_bias.mlp.2.bias, a gradient that is exactly 0 in exact arithmetic."""

    def __enter__(self):
        import oracle.deform as od
        self.od = od
        self.d = {}
        od.GRAD_PROBE = self.d
        return self

    def __exit__(self, *a):
        self.od.GRAD_PROBE = None

    def payload(self, p64):
        return {"natural:" + k: np.float64(self.d[id(v)]) for k, v in p64.items()
                if k.endswith("rel_pos_bias.mlp.2.bias") and id(v) in self.d}


def pack(d: dict) -> dict:
    out = {}
    for k, v in d.items():
        if isinstance(v, dict):
            for kk, vv in v.items():
                out[f"{k}/{kk}"] = vv
        else:
            out[k] = v
    return out


def install_stubs():
    class _Any(types.ModuleType):
        def __getattr__(self, name):
            if name.startswith("__"):
                raise AttributeError(name)
            return lambda *a, **k: None

    for name in ["lifelines", "lifelines.utils", "lifelines.statistics", "sksurv", "sksurv.metrics",
                 "imblearn", "imblearn.over_sampling", "imblearn.metrics", "torchvision", "wandb",
                 "h5py", "cv2", "skimage", "skimage.transform"]:
        if name not in sys.modules:
            sys.modules[name] = _Any(name)
    sys.path.insert(0, REF)
    nys = importlib.import_module("models.NystromAttention")
    stub = types.ModuleType("nystrom_attention")
    stub.NystromAttention = nys.NystromAttention
    sys.modules["nystrom_attention"] = stub


def load_synth(module: torch.nn.Module, seed: int, tag: str):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    params = synth.fill_params(shapes, seed=seed, tag=tag)
    module.load_state_dict(params)
    return params


def grads_of(module):
    return {k: p.grad for k, p in module.named_parameters() if p.grad is not None}


def save(name, payload):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **pack(payload))
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


# ---------------------------------------------------------------------------------------------
def case_deform2d(check):
    from models.DeformableAttention2D import DeformCrossAttention2D
    from oracle.deform import deform_cross_attention_2d, sample_positions
    torch.manual_seed(0)
    B, C, N = 2, 128, 2500
    mod = DeformCrossAttention2D(dim=C, dim_head=64, heads=8, dropout=0.1, downsample_factor=4,
                                 offset_scale=4, offset_groups=8, offset_kernel_size=6).eval()
    params = load_synth(mod, 42, "deform2d")
    x1 = synth.normal((B, C, N), 42, "deform2d:x1").requires_grad_()
    x2 = synth.normal((B, C, N), 42, "deform2d:x2").requires_grad_()
    w_out = synth.normal((B, C, N), 42, "deform2d:wout")
    w_vg = synth.normal((B * 8, 2, 12, 12), 42, "deform2d:wvg")
    out, vgrid = mod(x1, x2, return_vgrid=True)
    loss = (out * w_out).sum() + (vgrid * w_vg).sum()
    loss.backward()
    a64 = x1.detach().double().requires_grad_(); b64 = x2.detach().double().requires_grad_()
    p64 = {k: v.double().requires_grad_() for k, v in params.items()}
    with natural_scales() as ns:
        o64, vg64 = deform_cross_attention_2d(a64, b64, p64, grid_hw=(50, 50))
        ((o64 * w_out.double()).sum() + (vg64 * w_vg.double()).sum()).backward()
    payload = {"out": summarize(out, o64), "vgrid": summarize(vgrid, vg64), "loss": np.float64(loss.item()),
               "dx1": summarize(x1.grad, a64.grad), "dx2": summarize(x2.grad, b64.grad), **ns.payload(p64)}
    for k, g in grads_of(mod).items():
        payload["grad:" + k] = summarize(g, p64[k].grad)
    # integer sampling path for the reference's own vgrid (corner indices / masks)
    vs = 2.0 * vgrid.detach() / 11.0 - 1.0
    _, _, corners = sample_positions(vs[:, 0].reshape(B * 8, 144), vs[:, 1].reshape(B * 8, 144), 50, 50)
    payload["corner_x"] = torch.stack([c[0] for c in corners], -1).numpy().astype(np.int32)
    payload["corner_y"] = torch.stack([c[1] for c in corners], -1).numpy().astype(np.int32)
    payload["corner_mask"] = torch.stack([c[3] for c in corners], -1).numpy()
    payload["vgrid_full"] = vgrid.detach().numpy()
    save("deform2d_ref50", payload)
    if check:
        a = x1.detach().clone().requires_grad_(); b = x2.detach().clone().requires_grad_()
        po = {k: v.clone().requires_grad_() for k, v in params.items()}
        o2, vg2 = deform_cross_attention_2d(a, b, po, grid_hw=(50, 50))
        ((o2 * w_out).sum() + (vg2 * w_vg).sum()).backward()
        report("deform2d out", o2, out); report("deform2d vgrid", vg2, vgrid)
        report("deform2d dx1", a.grad, x1.grad); report("deform2d dx2", b.grad, x2.grad)
        for k, g in grads_of(mod).items():
            report("deform2d d" + k, po[k].grad, g)


def case_deform1d(check):
    from models.DeformableAttention1D import DeformCrossAttention1D
    from oracle.deform import deform_cross_attention_1d
    for tag, B, C, n in (("deform1d_n2501", 1, 128, 2501), ("deform1d_n37", 2, 128, 37), ("deform1d_n40", 2, 128, 40)):
        mod = DeformCrossAttention1D(dim=C, downsample_factor=4, offset_scale=2, offset_kernel_size=6).eval()
        params = load_synth(mod, 42, tag)
        x1 = synth.normal((B, C, n), 42, tag + ":x1").requires_grad_()
        x2 = synth.normal((B, C, n), 42, tag + ":x2").requires_grad_()
        w_out = synth.normal((B, C, n), 42, tag + ":wout")
        out, vgrid = mod(x1, x2, return_vgrid=True)
        w_vg = synth.normal(tuple(vgrid.shape), 42, tag + ":wvg")
        loss = (out * w_out).sum() + (vgrid * w_vg).sum()
        loss.backward()
        a64 = x1.detach().double().requires_grad_(); b64 = x2.detach().double().requires_grad_()
        p64 = {k: v.double().requires_grad_() for k, v in params.items()}
        with natural_scales() as ns:
            o64, vg64 = deform_cross_attention_1d(a64, b64, p64, offset_scale=2.0)
            ((o64 * w_out.double()).sum() + (vg64 * w_vg.double()).sum()).backward()
        payload = {"out": summarize(out, o64), "vgrid": summarize(vgrid, vg64), "loss": np.float64(loss.item()),
                   "dx1": summarize(x1.grad, a64.grad), "dx2": summarize(x2.grad, b64.grad), **ns.payload(p64)}
        for k, g in grads_of(mod).items():
            payload["grad:" + k] = summarize(g, p64[k].grad)
        save(tag, payload)
        if check:
            a = x1.detach().clone().requires_grad_(); b = x2.detach().clone().requires_grad_()
            po = {k: v.clone().requires_grad_() for k, v in params.items()}
            o2, vg2 = deform_cross_attention_1d(a, b, po, offset_scale=2.0)
            ((o2 * w_out).sum() + (vg2 * w_vg).sum()).backward()
            report(tag + " out", o2, out); report(tag + " vgrid", vg2, vgrid)
            report(tag + " dx1", a.grad, x1.grad); report(tag + " dx2", b.grad, x2.grad)
            for k, g in grads_of(mod).items():
                report(tag + " d" + k, po[k].grad, g)


def case_nystrom(check):
    from models.NystromAttention import NystromAttention, moore_penrose_iter_pinv
    from oracle.nystrom import nystrom_attention, pinv_newton_schulz
    for tag, B, n, dim, dh, m in (("nystrom_n37_m16", 2, 37, 64, 8, 16), ("nystrom_n64_m16", 2, 64, 64, 8, 16),
                                  ("nystrom_n257_m256", 1, 257, 512, 64, 256)):
        mod = NystromAttention(dim=dim, dim_head=dh, heads=8, num_landmarks=m, pinv_iterations=6,
                               residual=True, dropout=0.1).eval()
        params = load_synth(mod, 42, tag)
        x = synth.normal((B, n, dim), 42, tag + ":x").requires_grad_()
        w_out = synth.normal((B, n, dim), 42, tag + ":wout")
        out = mod(x)
        loss = (out * w_out).sum()
        loss.backward()
        payload = {"out": summarize(out), "loss": np.float64(loss.item()), "dx": summarize(x.grad)}
        for k, g in grads_of(mod).items():
            payload["grad:" + k] = summarize(g)
        save(tag, payload)
        if check:
            a = x.detach().clone().requires_grad_()
            po = {k: v.clone().requires_grad_() for k, v in params.items()}
            o2 = nystrom_attention(a, po, heads=8, dim_head=dh, num_landmarks=m)
            (o2 * w_out).sum().backward()
            report(tag + " out", o2, out); report(tag + " dx", a.grad, x.grad)
            for k, g in grads_of(mod).items():
                report(tag + " d" + k, po[k].grad, g)
    a2 = torch.softmax(synth.normal((2, 3, 16, 16), 42, "pinv:x"), dim=-1)
    z = moore_penrose_iter_pinv(a2, 6)
    save("pinv_m16", {"z": summarize(z)})
    if check:
        report("pinv", pinv_newton_schulz(a2, 6), z)


def option_masks(tag, *shape, frac=0.25):
    """Deterministic bool mask with ~frac True entries (a portable function of the synth stream)."""
    return synth.normal(shape, 42, tag) > 0.6745 * (1.0 if frac == 0.25 else 0.0)


def case_options(check):
    """Arguments of the path's modules that no caller in the reference passes, pinned by the reference all the same: the co-attention masks
    (models/MultiheadAttention.py:206-227,284-296), the NystromAttention mask (models/NystromAttention.py:84,92-96,106-118,127-133) and the raw-offset
    position bias of the 1-D deformable attention (cpb_log_distance=False)."""
    from models.MultiheadAttention import MultiheadAttention
    from models.NystromAttention import NystromAttention
    from oracle.coattn import coattention
    from oracle.nystrom import nystrom_attention
    tag, L, S, B = "coattn_masked_L37_S50", 37, 50, 2
    mod = MultiheadAttention(embed_dim=256, num_heads=1).eval()
    params = load_synth(mod, 42, tag)
    q = synth.normal((L, B, 256), 42, tag + ":q").requires_grad_()
    kv = synth.normal((S, B, 256), 42, tag + ":kv").requires_grad_()
    w_o = synth.normal((L, B, 256), 42, tag + ":wo")
    kpm = torch.zeros(B, S, dtype=torch.bool); kpm[0, -7:] = True; kpm[1, :3] = True          # True = ignored key
    am = option_masks(tag + ":am", L, S); am[:, 10] = False                                    # True = not allowed; no row is fully masked
    out, raw = mod(q, kv, kv, key_padding_mask=kpm, attn_mask=am)
    (out * w_o).sum().backward()
    q64 = q.detach().double().requires_grad_(); kv64 = kv.detach().double().requires_grad_()
    p64 = {k: v.double().requires_grad_() for k, v in params.items()}
    o64, r64 = coattention(q64, kv64, kv64, p64, key_padding_mask=kpm, attn_mask=am)
    (o64 * w_o.double()).sum().backward()
    fin = torch.isfinite(raw)
    assert torch.equal(fin, torch.isfinite(r64)) and not bool(fin.all())
    payload = {"out": summarize(out, o64), "raw_finite": summarize(torch.where(fin, raw, torch.zeros_like(raw)), torch.where(fin, r64, torch.zeros_like(r64))),
               "masked_count": np.int64(int((~fin).sum())), "dq": summarize(q.grad, q64.grad), "dkv": summarize(kv.grad, kv64.grad)}
    for k, g in grads_of(mod).items():
        payload["grad:" + k] = summarize(g, p64[k].grad)
    save(tag, payload)
    if check:
        report(tag + " out", o64.float(), out); report(tag + " dkv", kv64.grad.float(), kv.grad)
    # DeformCrossAttention1D(cpb_log_distance=False): the bias MLP reads the raw offset (models/DeformableAttention1D.py:92,118,148)
    from models.DeformableAttention1D import DeformCrossAttention1D
    from oracle.deform import deform_cross_attention_1d
    for tag, B, C, n in (("deform1d_rawdist_n40", 2, 128, 40), ("deform1d_rawdist_n300", 1, 128, 300)):
        mod = DeformCrossAttention1D(dim=C, downsample_factor=4, offset_scale=2, offset_kernel_size=6, cpb_log_distance=False).eval()
        params = load_synth(mod, 42, tag)
        x1 = synth.normal((B, C, n), 42, tag + ":x1").requires_grad_()
        x2 = synth.normal((B, C, n), 42, tag + ":x2").requires_grad_()
        w_out = synth.normal((B, C, n), 42, tag + ":wout")
        out, vgrid = mod(x1, x2, return_vgrid=True)
        w_vg = synth.normal(tuple(vgrid.shape), 42, tag + ":wvg")
        loss = (out * w_out).sum() + (vgrid * w_vg).sum()
        loss.backward()
        a64 = x1.detach().double().requires_grad_(); b64 = x2.detach().double().requires_grad_()
        p64 = {k: v.double().requires_grad_() for k, v in params.items()}
        with natural_scales() as ns:
            o64, vg64 = deform_cross_attention_1d(a64, b64, p64, offset_scale=2.0, cpb_log_distance=False)
            ((o64 * w_out.double()).sum() + (vg64 * w_vg.double()).sum()).backward()
        payload = {"out": summarize(out, o64), "vgrid": summarize(vgrid, vg64), "loss": np.float64(loss.item()),
                   "dx1": summarize(x1.grad, a64.grad), "dx2": summarize(x2.grad, b64.grad), **ns.payload(p64)}
        for k, g in grads_of(mod).items():
            payload["grad:" + k] = summarize(g, p64[k].grad)
        save(tag, payload)
        if check:
            a = x1.detach().clone().requires_grad_(); b = x2.detach().clone().requires_grad_()
            po = {k: v.clone().requires_grad_() for k, v in params.items()}
            o2, vg2 = deform_cross_attention_1d(a, b, po, offset_scale=2.0, cpb_log_distance=False)
            ((o2 * w_out).sum() + (vg2 * w_vg).sum()).backward()
            report(tag + " out", o2, out); report(tag + " dx1", a.grad, x1.grad)
            for k, g in grads_of(mod).items():
                report(tag + " d" + k, po[k].grad, g)
    for tag, B, n, dim, dh, m in (("nystrom_masked_n37_m16", 2, 37, 64, 8, 16), ("nystrom_masked_n64_m16", 2, 64, 64, 8, 16)):
        mod = NystromAttention(dim=dim, dim_head=dh, heads=8, num_landmarks=m, pinv_iterations=6, residual=True, dropout=0.1).eval()
        params = load_synth(mod, 42, tag)
        x = synth.normal((B, n, dim), 42, tag + ":x").requires_grad_()
        w_out = synth.normal((B, n, dim), 42, tag + ":wout")
        mask = ~option_masks(tag + ":mask", B, n)                   # True = token takes part
        mask[0, :5] = False                                        # a whole segment of bag 0 masked out (an all-masked landmark)
        out = mod(x, mask=mask)
        (out * w_out).sum().backward()
        payload = {"out": summarize(out), "dx": summarize(x.grad), "kept": np.int64(int(mask.sum()))}
        for k, g in grads_of(mod).items():
            payload["grad:" + k] = summarize(g)
        save(tag, payload)
        if check:
            a = x.detach().clone().requires_grad_()
            po = {k: v.clone().requires_grad_() for k, v in params.items()}
            o2 = nystrom_attention(a, po, heads=8, dim_head=dh, num_landmarks=m, mask=mask)
            (o2 * w_out).sum().backward()
            report(tag + " out", o2, out); report(tag + " dx", a.grad, x.grad)
            for k, g in grads_of(mod).items():
                report(tag + " d" + k, po[k].grad, g)


def case_translayer(check):
    from models.cmta_utils import TransLayer, PPEG
    from oracle.nystrom import trans_layer, ppeg
    dim = 64
    mod = TransLayer(dim=dim).eval()
    params = load_synth(mod, 42, "translayer")
    x = synth.normal((2, 1 + 36, dim), 42, "translayer:x").requires_grad_()
    w = synth.normal((2, 37, dim), 42, "translayer:w")
    out = mod(x); (out * w).sum().backward()
    payload = {"out": summarize(out), "dx": summarize(x.grad)}
    for k, g in grads_of(mod).items():
        payload["grad:" + k] = summarize(g)
    save("translayer_d64", payload)
    pp = PPEG(dim=dim).eval()
    pparams = load_synth(pp, 42, "ppeg")
    y = pp(x.detach(), 6, 6)
    save("ppeg_d64", {"out": summarize(y)})
    if check:
        a = x.detach().clone().requires_grad_()
        po = {k: v.clone().requires_grad_() for k, v in params.items()}
        o2 = trans_layer(a, po, dim=dim); (o2 * w).sum().backward()
        report("translayer out", o2, out); report("translayer dx", a.grad, x.grad)
        report("ppeg out", ppeg(x.detach(), 6, 6, pparams), y)


def ref_args(**over):
    a = argparse.Namespace(
        mode="deformpathomic", act_type="Sigmoid", init_type="max", init_gain=0.02, fusion_type="concat",
        skip=0, use_bilinear=1, input_size_omic=431, input_size_omic_tumor=59, input_size_omic_immune=361,
        path_gate=1, omic_gate=1, path_dim=128, omic_dim=128, path_scale=1, omic_scale=1, mmhid=128,
        cut_fuse_grad=False, dropout_rate=0.1, return_grad="False", label_dim=4, task_type="diag2021",
        attn_dim=2, return_vgrid=True, batch_size=2, world_size=1)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def case_pathomic(check):
    from models.model import define_net
    from utils.loss import BatchLoss
    from oracle.mil import deform_pathomic_net
    from oracle.losses import batch_loss
    args = ref_args()
    net = define_net(args).eval()
    params = load_synth(net, 42, "pathomic")
    B = 2
    x_path = synth.bag(B, 2500, 1024, 42, "pathomic:bag")
    x_t = synth.normal((B, 59), 42, "pathomic:tumor")
    x_i = synth.normal((B, 361), 42, "pathomic:immune")
    feats, vt, vi, logits, _, _, _ = net(x_path=x_path, x_omic=None, x_omic_tumor=x_t, x_omic_immune=x_i)
    bl = BatchLoss(B, 1)
    l_t, l_i = bl(logits[3], logits[4]), bl(logits[5], logits[6])
    label = torch.tensor([1, 3])
    ce = torch.nn.functional.cross_entropy(logits[2], label)
    loss = ce + 0.5 * l_t.sum() + 0.5 * l_i.sum()
    loss.backward()
    p64 = {k: (v.double().requires_grad_() if v.dtype.is_floating_point else v) for k, v in params.items()}
    with natural_scales() as ns:
        f64, vt64, vi64, lg64 = deform_pathomic_net(x_path.double(), x_t.double(), x_i.double(), p64, grid_hw=(50, 50))
        lt64, li64 = batch_loss(lg64[3], lg64[4], B), batch_loss(lg64[5], lg64[6], B)
        (torch.nn.functional.cross_entropy(lg64[2], label) + 0.5 * lt64.sum() + 0.5 * li64.sum()).backward()
    payload = {"features": summarize(feats, f64), "vec_t": summarize(vt, vt64), "vec_i": summarize(vi, vi64),
               "haz_t": summarize(logits[0], lg64[0]), "haz_i": summarize(logits[1], lg64[1]),
               "haz": summarize(logits[2], lg64[2]),
               "vgrid_t": summarize(logits[4], lg64[4]), "vgrid_i": summarize(logits[6], lg64[6]),
               "omic_t_row0": summarize(logits[3][:, 0]), "batchloss_t": summarize(l_t, lt64),
               "batchloss_i": summarize(l_i, li64), "loss": np.float64(loss.item()), **ns.payload(p64)}
    g = grads_of(net)
    payload["n_params_with_grad"] = np.int64(len(g))
    for k, v in g.items():
        payload["grad:" + k] = summarize(v, p64[k].grad)
    save("pathomic_ref50", payload)
    if check:
        po = {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in params.items()}
        f2, vt2, vi2, lg2 = deform_pathomic_net(x_path, x_t, x_i, po, grid_hw=(50, 50))
        l2 = (torch.nn.functional.cross_entropy(lg2[2], label) + 0.5 * batch_loss(lg2[3], lg2[4], B).sum()
              + 0.5 * batch_loss(lg2[5], lg2[6], B).sum())
        l2.backward()
        report("pathomic features", f2, feats); report("pathomic haz", lg2[2], logits[2])
        report("pathomic vgrid_t", lg2[4], logits[4]); report("pathomic loss", l2, loss)
        worst = 0.0
        for k, v in g.items():
            worst = max(worst, rel_err(po[k].grad, v))
        print(f"  pathomic worst param-grad rel err = {worst:.3e} over {len(g)} tensors")


def case_losses(check):
    from utils.loss import BatchLoss
    from models.cmta_utils import OrthogonalLoss
    from oracle.losses import batch_loss, orthogonal_loss
    B = 4
    omic = synth.normal((B, 50, 16), 42, "bl:omic").requires_grad_()
    vgrid = synth.normal((B * 8, 2, 3, 3), 42, "bl:vgrid").requires_grad_()
    out = BatchLoss(B, 1)(omic, vgrid); out.sum().backward()
    save("batchloss_b4", {"out": summarize(out), "domic": summarize(omic.grad), "dvgrid": summarize(vgrid.grad)})
    P, Ph, G, Gh = (synth.normal((B, 256), 42, "ol:" + t).requires_grad_() for t in "abcd")
    ol = OrthogonalLoss()(P, Ph, G, Gh); ol.sum().backward()
    save("orthloss_b4", {"out": summarize(ol), "dP": summarize(P.grad), "dPh": summarize(Ph.grad),
                         "dG": summarize(G.grad), "dGh": summarize(Gh.grad)})
    if check:
        report("batchloss", batch_loss(omic.detach(), vgrid.detach(), B), out)
        report("orthloss", orthogonal_loss(P.detach(), Ph.detach(), G.detach(), Gh.detach()), ol)


def case_coattn_fusion(check):
    from models.MultiheadAttention import MultiheadAttention
    from oracle.coattn import bilinear_fusion, coattention
    for tag, L, S, B in (("coattn_L4_S2500", 4, 2500, 2), ("coattn_L2500_S4", 2500, 4, 2), ("coattn_L200_S4096", 200, 4096, 1)):
        mod = MultiheadAttention(embed_dim=256, num_heads=1).eval()
        params = load_synth(mod, 42, tag)
        q = synth.normal((L, B, 256), 42, tag + ":q").requires_grad_()
        kv = synth.normal((S, B, 256), 42, tag + ":kv").requires_grad_()
        w_o = synth.normal((L, B, 256), 42, tag + ":wo"); w_r = synth.normal((B, 1, L, S), 42, tag + ":wr")
        out, raw = mod(q, kv, kv)
        ((out * w_o).sum() + (raw * w_r).sum() * 1e-2).backward()
        q64 = q.detach().double().requires_grad_(); kv64 = kv.detach().double().requires_grad_()
        p64 = {k: v.double().requires_grad_() for k, v in params.items()}
        o64, r64 = coattention(q64, kv64, kv64, p64)
        ((o64 * w_o.double()).sum() + (r64 * w_r.double()).sum() * 1e-2).backward()
        payload = {"out": summarize(out, o64), "raw": summarize(raw, r64), "dq": summarize(q.grad, q64.grad),
                   "dkv": summarize(kv.grad, kv64.grad)}
        for k, g in grads_of(mod).items():
            payload["grad:" + k] = summarize(g, p64[k].grad)
        save(tag, payload)
        if check:
            report(tag + " out", o64.float(), out); report(tag + " raw", r64.float(), raw); report(tag + " dkv", kv64.grad.float(), kv.grad)
    # BilinearFusion (eval mode: BatchNorm running stats, dropout off); forward builds torch.cuda.FloatTensor -> CPU patch
    torch.cuda.FloatTensor = torch.FloatTensor
    from models.fusion import BilinearFusion
    for tag, skip in (("bifusion_skip0", 0), ("bifusion_skip1", 1)):
        mod = BilinearFusion(skip=skip, use_bilinear=1, gate1=1, gate2=1, dim1=128, dim2=128, scale_dim1=1, scale_dim2=1,
                             mmhid=128, dropout_rate=0.1).eval()
        shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items()}
        params = synth.fill_params({k: s for k, s in shapes.items() if "num_batches" not in k}, seed=42, tag=tag)
        for k in list(params):
            if k.endswith("running_var"):
                params[k] = params[k].abs() + 0.5
        sd = dict(mod.state_dict()); sd.update(params); mod.load_state_dict(sd)
        v1 = synth.normal((4, 128), 42, tag + ":v1").requires_grad_(); v2 = synth.normal((4, 128), 42, tag + ":v2").requires_grad_()
        w = synth.normal((4, 128), 42, tag + ":w")
        out = mod(v1, v2); (out * w).sum().backward()
        a64 = v1.detach().double().requires_grad_(); b64 = v2.detach().double().requires_grad_()
        p64 = {k: (v.double().requires_grad_() if "running" not in k else v.double()) for k, v in params.items()}
        o64 = bilinear_fusion(a64, b64, p64, skip=skip); (o64 * w.double()).sum().backward()
        payload = {"out": summarize(out, o64), "dv1": summarize(v1.grad, a64.grad), "dv2": summarize(v2.grad, b64.grad)}
        for k, g in grads_of(mod).items():
            payload["grad:" + k] = summarize(g, p64[k].grad)
        save(tag, payload)
        if check:
            report(tag + " out", o64.float(), out); report(tag + " dv1", a64.grad.float(), v1.grad)


def case_cmta(check):
    """CMTA (the reference's default mode) in eval(): B = 2 bags of 150 x 1024 (wrap-padded to 13 x 13 + cls = 170 tokens ->
    Nystrom front padding to 256, l = 2) and 4 omic signatures.  The reference's Transformer_P/G call .cuda() on the cls
    token (cmta_utils.py:914,940): patched to the identity for this CPU run."""
    global MAX_KEEP
    MAX_KEEP = 1024                                     # 104 parameter tensors: keep the fixture small
    torch.Tensor.cuda = lambda self, *a, **k: self
    from models.model import CMTA
    from oracle.cmta import cmta
    args = argparse.Namespace(label_dim=4)
    mod = CMTA(args).eval()
    params = load_synth(mod, 42, "cmta")
    for k in params:                                   # cls tokens are N(0, 1e-6) at init: use that scale
        if k.endswith("cls_token"):
            params[k] = params[k] * 1e-3
    mod.load_state_dict(params)
    B, n = 2, 150
    x_path = synth.bag(B, n, 1024, 42, "cmta:bag").requires_grad_()
    x_omic = synth.normal((B, 431), 42, "cmta:omic").requires_grad_()
    w = [synth.normal((B, 4), 42, "cmta:w0")] + [synth.normal((B, 256), 42, f"cmta:w{i}") for i in range(1, 5)]
    out = mod(x_path=x_path, x_omic=x_omic)
    loss = (out[0] * w[0]).sum() + sum((out[2 + i] * w[i]).sum() for i in range(1, 5))
    loss.backward()
    p64 = {k: v.double().requires_grad_() for k, v in params.items()}
    a64, b64 = x_path.detach().double().requires_grad_(), x_omic.detach().double().requires_grad_()
    o64 = cmta(a64, b64, p64)
    ((o64[0] * w[0].double()).sum() + sum((o64[2 + i] * w[i].double()).sum() for i in range(1, 5))).backward()
    names = ("logits", "hazards", "S", "cls_p_enc", "cls_p_dec", "cls_g_enc", "cls_g_dec")
    payload = {nm: summarize(o, o6) for nm, o, o6 in zip(names, out, o64)}
    payload["loss"] = np.float64(loss.item())
    payload["dx_path"] = summarize(x_path.grad, a64.grad); payload["dx_omic"] = summarize(x_omic.grad, b64.grad)
    for k, g in grads_of(mod).items():
        payload["grad:" + k] = summarize(g, p64[k].grad)
    save("cmta_n150", payload)
    if check:
        po = {k: v.clone().requires_grad_() for k, v in params.items()}
        a, b = x_path.detach().clone().requires_grad_(), x_omic.detach().clone().requires_grad_()
        o2 = cmta(a, b, po)
        ((o2[0] * w[0]).sum() + sum((o2[2 + i] * w[i]).sum() for i in range(1, 5))).backward()
        for nm, x, y in zip(names, o2, out):
            report("cmta " + nm, x, y)
        report("cmta dx_path", a.grad, x_path.grad)
        worst = max(rel_err(po[k].grad, g) for k, g in grads_of(mod).items())
        print(f"  cmta worst param-grad rel err = {worst:.3e} over {len(grads_of(mod))} tensors")


def gradmod_case(seed):
    """Synthetic inputs of one gradient-modulation case (shared with the tests): B = 8 samples, C = 4 classes, hs = 128."""
    B, C, hs = 8, 4, 128
    tag = f"gradmod:{seed}"
    ft = synth.normal((B, hs), seed, tag + ":ft"); fi = synth.normal((B, hs), seed, tag + ":fi")
    W = synth.normal((C, 2 * hs), seed, tag + ":W") * 0.2; b = synth.normal((C,), seed, tag + ":b") * 0.1
    G = synth.normal((C, 2 * hs), seed, tag + ":G") * 0.05
    label = (synth.normal((B,), seed, tag + ":lab").abs() * 1.7).long().clamp(max=C - 1)
    return ft, fi, W, b, label, G


def case_gradmod(check):
    """The gradient-modulation block of the reference's training loop (train_test.py:87-184) is inline code, not a function:
    its source lines are taken from the imported module at generation time (inspect), dedented and executed on synthetic
    inputs with a stand-in for `model.module.classifier` - the reference's own statements produce the fixture."""
    import inspect, textwrap
    import torch.nn as nn
    import torch.nn.functional as F
    import train_test
    from oracle.trainstep import gradient_modulate
    src = inspect.getsource(train_test.trainDeformPathomicModel).split("\n")
    start = next(i for i, ln in enumerate(src) if "if args.gradient_modulate:" in ln)
    end = next(i for i, ln in enumerate(src) if "# Update parameters based on projected gradients" in ln)
    block = textwrap.dedent("\n".join(src[start:end]))
    payload = {}
    for seed in range(1, 9):
        ft, fi, W, b, label, G = gradmod_case(seed)
        cls = nn.Linear(256, 4)
        cls.weight.data.copy_(W); cls.bias.data.copy_(b); cls.weight.grad = G.clone()
        model = argparse.Namespace(module=argparse.Namespace(classifier=cls))
        lab12 = torch.zeros(8, 12, dtype=torch.long); lab12[:, 5] = label
        ns = dict(torch=torch, F=F, nn=nn, np=np, args=argparse.Namespace(gradient_modulate=True, mmhid=128, task_type="diag2021"),
                  model=model, pathomic_feat_tumor=ft, pathomic_feat_immune=fi, label=lab12,
                  diag2021_loss_func=nn.CrossEntropyLoss(), cosine_similarity=train_test.cosine_similarity)
        exec(block, ns)
        out = cls.weight.grad.detach()
        payload[f"case{seed}/grad"] = out.numpy().copy()
        payload[f"case{seed}/ratio_t"] = np.float64(float(ns["ratio_t"]))
        payload[f"case{seed}/changed_rows"] = (out != G).any(dim=1).numpy()
        if check:
            g2, info = gradient_modulate(ft, fi, W, b, label, G)
            report(f"gradmod case {seed} (ratio_t {float(ns['ratio_t']):.3f}, branches {info['branch']})", g2, out)
    np.savez_compressed(os.path.join(HERE, "gradmod_b8.npz"), **payload)
    print("wrote gradmod_b8.npz")


FIXDIM_CASES = [(n, f) for f in (50, 2500, 10000) for n in (1, 7, 49, 50, 51, 811, 2499, 2500, 2501, 3000, 12345)]


def case_fixdim(check):
    """The fixed-instance-count rule (f3) is embedded in the image-reading loop of the reference's datasets
    (data/dataset.py:142-175, IvYGAP_Dataset.read_img) and needs the patch files to run as it stands.  Here the METHOD'S OWN
    STATEMENTS are executed: its source is taken from the imported class at generation time (inspect), cut at the end of the
    index-selection block (:175, before the image reshapes), and run with stand-ins for the file system only - `np.load` returns a
    synthetic read_details table whose row i is (i, 0), `io.imread` returns the row number encoded in the requested file name,
    `os.listdir` returns nothing.  `patch_all` is then the list of source rows the reference would have read, in order."""
    import inspect, textwrap
    for name in ("skimage.io", "torchvision.transforms"):
        if name not in sys.modules:
            sys.modules[name] = sys.modules["h5py"].__class__(name)
    sys.modules["skimage"].io = sys.modules["skimage.io"]; sys.modules["skimage"].transform = sys.modules["skimage.transform"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].Compose = object
    import data.dataset as ds
    from oracle.bagstore import fixdim_indices
    src = inspect.getsource(ds.IvYGAP_Dataset.read_img).split("\n")
    end = next(i for i, ln in enumerate(src) if "patch_all = np.asarray(patch_all)" in ln)
    body = textwrap.dedent("\n".join(src[1:end]))                 # the statements between `def read_img(self, index):` and :175
    assert "np.around(i * (num_patches / max_num))" in body and "max_num % num_patches" in body

    class _Np:                                                    # numpy with np.load replaced (the only file access through np)
        def __init__(self, table): self._t = table
        def __getattr__(self, k): return getattr(np, k)
        def load(self, *a, **k): return [self._t]

    payload = {}
    for n, fixdim in FIXDIM_CASES:
        table = np.stack((np.arange(n), np.zeros(n, dtype=np.int64)), axis=1)
        io = argparse.Namespace(imread=lambda path: int(os.path.basename(path).split("_")[0]))
        self_ = argparse.Namespace(dataDir="/nowhere/", LIST=np.array([["0", "bag"]]), args=argparse.Namespace(dataDir="/nowhere/", fixdim=fixdim))
        ns = dict(np=_Np(table), io=io, os=argparse.Namespace(listdir=lambda p: [], path=os.path), self=self_, index=0)
        exec(body, ns)
        idx = np.asarray(ns["patch_all"], dtype=np.int64)
        assert idx.shape == (fixdim,), (n, fixdim, idx.shape)
        payload[f"n{n}_f{fixdim}"] = idx.astype(np.int32)
        if check:
            mine = fixdim_indices(n, fixdim)
            print(f"  oracle vs reference  fixdim n={n:<6d} fixdim={fixdim:<6d} equal={bool(np.array_equal(mine, idx))}")
            assert np.array_equal(mine, idx)
    np.savez_compressed(os.path.join(HERE, "fixdim_indices.npz"), **payload)
    print("wrote fixdim_indices.npz", os.path.getsize(os.path.join(HERE, "fixdim_indices.npz")) // 1024, "KiB")


def rel_err(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def report(name, a, b):
    print(f"  oracle vs reference  {name:<42s} rel-max-err {rel_err(a, b):.3e}")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-check", action="store_true", help="skip the oracle-vs-reference report")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 1)
    install_stubs()
    cases = {"deform2d": case_deform2d, "deform1d": case_deform1d, "nystrom": case_nystrom,
             "translayer": case_translayer, "pathomic": case_pathomic, "losses": case_losses,
             "coattn": case_coattn_fusion, "cmta": case_cmta, "gradmod": case_gradmod, "fixdim": case_fixdim, "options": case_options}
    for k, fn in cases.items():
        if a.only and k not in a.only.split(","):
            continue
        print(f"[{k}]")
        fn(not a.no_check)

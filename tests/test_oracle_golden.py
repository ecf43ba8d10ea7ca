"""CPU: pins the oracle (oracle/*.py) against the golden vectors generated from the reference
(tests/golden/make_golden.py).  Tolerance: 1e-4 relative to each tensor's scale (north_star); the oracle
actually agrees to ~2e-6."""
import argparse
import os

import numpy as np
import pytest
import torch

from helpers import Golden, assert_zero_grad, params_for, smml, synth
from oracle.deform import deform_cross_attention_1d, deform_cross_attention_2d, sample_positions
from oracle.losses import batch_loss, orthogonal_loss
from oracle.mil import deform_pathomic_net
from oracle.nystrom import nystrom_attention, pinv_newton_schulz, ppeg, trans_layer

ZERO_GRADS = ("rel_pos_bias.mlp.2.bias",)   # softmax is shift invariant: this gradient is exactly 0 (noise only)


def _req(params):
    return {k: v.clone().requires_grad_(v.dtype.is_floating_point) for k, v in params.items()}


def _check_grads(g, params, prefix="grad:", rtol=1e-4):
    for key in g.keys(prefix):
        name = key[len(prefix):]
        if name.endswith(ZERO_GRADS):
            if g.has("natural:" + name):       # exactly 0 in exact arithmetic: bounded by 1e-4 x sum |d bias|
                assert_zero_grad(f"{g.name}:d{name}", params[name].grad, g.scalar("natural:" + name))
            continue
        assert params[name].grad is not None, name
        g.check(key, params[name].grad, rtol=rtol, what="d" + name)


def test_deform2d_reference_grid():
    g = Golden("deform2d_ref50")
    B, C, N = 2, 128, 2500
    mod = smml.DeformCrossAttention2D(dim=C, dim_head=64, heads=8, dropout=0.1, downsample_factor=4, offset_scale=4,
                                      offset_groups=8, offset_kernel_size=6)
    p = _req(params_for(mod, 42, "deform2d"))
    x1 = synth.normal((B, C, N), 42, "deform2d:x1").requires_grad_()
    x2 = synth.normal((B, C, N), 42, "deform2d:x2").requires_grad_()
    w_out = synth.normal((B, C, N), 42, "deform2d:wout")
    w_vg = synth.normal((B * 8, 2, 12, 12), 42, "deform2d:wvg")
    out, vgrid, aux = deform_cross_attention_2d(x1, x2, p, grid_hw=(50, 50), return_aux=True)
    loss = (out * w_out).sum() + (vgrid * w_vg).sum()
    loss.backward()
    g.check("out", out); g.check("vgrid", vgrid); g.check("dx1", x1.grad); g.check("dx2", x2.grad)
    assert abs(loss.item() - g.scalar("loss")) <= 1e-4 * abs(g.scalar("loss"))
    _check_grads(g, p)
    # integer path: corners / masks computed from the REFERENCE's vgrid must match bit for bit
    vg = torch.from_numpy(g.array("vgrid_full"))
    vs = 2.0 * vg / 11.0 - 1.0
    _, _, corners = sample_positions(vs[:, 0].reshape(16, 144), vs[:, 1].reshape(16, 144), 50, 50)
    assert np.array_equal(torch.stack([c[0] for c in corners], -1).numpy().astype(np.int32), g.array("corner_x"))
    assert np.array_equal(torch.stack([c[1] for c in corners], -1).numpy().astype(np.int32), g.array("corner_y"))
    assert np.array_equal(torch.stack([c[3] for c in corners], -1).numpy(), g.array("corner_mask"))


@pytest.mark.parametrize("tag,B,n", [("deform1d_n37", 2, 37), ("deform1d_n40", 2, 40), ("deform1d_n2501", 1, 2501)])
def test_deform1d(tag, B, n):
    g = Golden(tag)
    C = 128
    mod = smml.DeformCrossAttention1D(dim=C, downsample_factor=4, offset_scale=2, offset_kernel_size=6)
    p = _req(params_for(mod, 42, tag))
    x1 = synth.normal((B, C, n), 42, tag + ":x1").requires_grad_()
    x2 = synth.normal((B, C, n), 42, tag + ":x2").requires_grad_()
    w_out = synth.normal((B, C, n), 42, tag + ":wout")
    out, vgrid = deform_cross_attention_1d(x1, x2, p, offset_scale=2.0)
    w_vg = synth.normal(tuple(vgrid.shape), 42, tag + ":wvg")
    ((out * w_out).sum() + (vgrid * w_vg).sum()).backward()
    g.check("out", out); g.check("vgrid", vgrid); g.check("dx1", x1.grad); g.check("dx2", x2.grad)
    _check_grads(g, p)


@pytest.mark.parametrize("tag,B,n", [("deform1d_rawdist_n40", 2, 40), ("deform1d_rawdist_n300", 1, 300)])
def test_deform1d_raw_distance(tag, B, n):
    """DeformCrossAttention1D(cpb_log_distance=False): the bias MLP reads the raw offset (DeformableAttention1D.py:92; no caller in the reference)."""
    g = Golden(tag)
    C = 128
    mod = smml.DeformCrossAttention1D(dim=C, downsample_factor=4, offset_scale=2, offset_kernel_size=6, cpb_log_distance=False)
    p = _req(params_for(mod, 42, tag))
    x1 = synth.normal((B, C, n), 42, tag + ":x1").requires_grad_()
    x2 = synth.normal((B, C, n), 42, tag + ":x2").requires_grad_()
    w_out = synth.normal((B, C, n), 42, tag + ":wout")
    out, vgrid = deform_cross_attention_1d(x1, x2, p, offset_scale=2.0, cpb_log_distance=False)
    w_vg = synth.normal(tuple(vgrid.shape), 42, tag + ":wvg")
    ((out * w_out).sum() + (vgrid * w_vg).sum()).backward()
    g.check("out", out); g.check("vgrid", vgrid); g.check("dx1", x1.grad); g.check("dx2", x2.grad)
    _check_grads(g, p)


class _NysShapes(torch.nn.Module):
    def __init__(self, dim, dh, heads=8, k=33):
        super().__init__()
        inner = dh * heads
        self.to_qkv = torch.nn.Linear(dim, inner * 3, bias=False)
        self.to_out = torch.nn.Sequential(torch.nn.Linear(inner, dim), torch.nn.Dropout(0.))
        self.res_conv = torch.nn.Conv2d(heads, heads, (k, 1), padding=(k // 2, 0), groups=heads, bias=False)


@pytest.mark.parametrize("tag,B,n,dim,dh,m", [("nystrom_n37_m16", 2, 37, 64, 8, 16), ("nystrom_n64_m16", 2, 64, 64, 8, 16),
                                               ("nystrom_n257_m256", 1, 257, 512, 64, 256)])
def test_nystrom(tag, B, n, dim, dh, m):
    g = Golden(tag)
    p = _req(params_for(_NysShapes(dim, dh), 42, tag))
    x = synth.normal((B, n, dim), 42, tag + ":x").requires_grad_()
    w_out = synth.normal((B, n, dim), 42, tag + ":wout")
    out = nystrom_attention(x, p, heads=8, dim_head=dh, num_landmarks=m)
    (out * w_out).sum().backward()
    g.check("out", out); g.check("dx", x.grad)
    _check_grads(g, p)


def option_masks(tag, *shape):
    """The masks tests/golden/make_golden.py::case_options fed the reference (a portable function of the synth stream)."""
    return synth.normal(shape, 42, tag) > 0.6745


@pytest.mark.parametrize("tag,B,n,dim,dh,m", [("nystrom_masked_n37_m16", 2, 37, 64, 8, 16), ("nystrom_masked_n64_m16", 2, 64, 64, 8, 16)])
def test_nystrom_masked(tag, B, n, dim, dh, m):
    """NystromAttention's `mask` argument (models/NystromAttention.py:84,92-96,106-118,127-133; no caller in the reference passes it), incl. a
    landmark whose whole segment is masked."""
    g = Golden(tag)
    p = _req(params_for(_NysShapes(dim, dh), 42, tag))
    x = synth.normal((B, n, dim), 42, tag + ":x").requires_grad_()
    w_out = synth.normal((B, n, dim), 42, tag + ":wout")
    mask = ~option_masks(tag + ":mask", B, n); mask[0, :5] = False
    assert int(mask.sum()) == int(g.scalar("kept"))
    out = nystrom_attention(x, p, heads=8, dim_head=dh, num_landmarks=m, mask=mask)
    (out * w_out).sum().backward()
    g.check("out", out); g.check("dx", x.grad)
    _check_grads(g, p)


def coattn_masks(tag, L, S, B):
    kpm = torch.zeros(B, S, dtype=torch.bool); kpm[0, -7:] = True; kpm[1, :3] = True
    am = option_masks(tag + ":am", L, S); am[:, 10] = False
    return kpm, am


def test_coattention_masked():
    """key_padding_mask + bool attn_mask of the co-attention (models/MultiheadAttention.py:206-227,284-296); the raw scores returned are the masked
    ones (-inf where masked)."""
    from oracle.coattn import coattention
    tag, L, S, B = "coattn_masked_L37_S50", 37, 50, 2
    g = Golden(tag)
    p = _req(params_for(smml.MultiheadAttention(256, 1), 42, tag))
    q = synth.normal((L, B, 256), 42, tag + ":q").requires_grad_()
    kv = synth.normal((S, B, 256), 42, tag + ":kv").requires_grad_()
    w_o = synth.normal((L, B, 256), 42, tag + ":wo")
    kpm, am = coattn_masks(tag, L, S, B)
    out, raw = coattention(q, kv, kv, p, key_padding_mask=kpm, attn_mask=am)
    (out * w_o).sum().backward()
    fin = torch.isfinite(raw)
    assert int((~fin).sum()) == int(g.scalar("masked_count"))
    g.check("out", out); g.check("raw_finite", torch.where(fin, raw, torch.zeros_like(raw))); g.check("dq", q.grad); g.check("dkv", kv.grad)
    _check_grads(g, p)


def test_pinv_and_translayer_and_ppeg():
    a2 = torch.softmax(synth.normal((2, 3, 16, 16), 42, "pinv:x"), dim=-1)
    Golden("pinv_m16").check("z", pinv_newton_schulz(a2, 6))

    class TL(torch.nn.Module):
        def __init__(self, dim):
            super().__init__()
            self.norm = torch.nn.LayerNorm(dim)
            self.attn = _NysShapes(dim, dim // 8)
    dim = 64
    g = Golden("translayer_d64")
    p = _req(params_for(TL(dim), 42, "translayer"))
    x = synth.normal((2, 37, dim), 42, "translayer:x").requires_grad_()
    w = synth.normal((2, 37, dim), 42, "translayer:w")
    out = trans_layer(x, p, dim=dim)
    (out * w).sum().backward()
    g.check("out", out); g.check("dx", x.grad)
    _check_grads(g, p)

    class PP(torch.nn.Module):
        def __init__(self, dim):
            super().__init__()
            self.proj = torch.nn.Conv2d(dim, dim, 7, 1, 3, groups=dim)
            self.proj1 = torch.nn.Conv2d(dim, dim, 5, 1, 2, groups=dim)
            self.proj2 = torch.nn.Conv2d(dim, dim, 3, 1, 1, groups=dim)
    Golden("ppeg_d64").check("out", ppeg(x.detach(), 6, 6, params_for(PP(dim), 42, "ppeg")))


def test_losses():
    B = 4
    g = Golden("batchloss_b4")
    omic = synth.normal((B, 50, 16), 42, "bl:omic").requires_grad_()
    vgrid = synth.normal((B * 8, 2, 3, 3), 42, "bl:vgrid").requires_grad_()
    out = batch_loss(omic, vgrid, B)
    out.sum().backward()
    g.check("out", out); g.check("domic", omic.grad); g.check("dvgrid", vgrid.grad)
    g = Golden("orthloss_b4")
    P, Ph, G, Gh = (synth.normal((B, 256), 42, "ol:" + t).requires_grad_() for t in "abcd")
    ol = orthogonal_loss(P, Ph, G, Gh)
    ol.sum().backward()
    g.check("out", ol); g.check("dP", P.grad); g.check("dPh", Ph.grad); g.check("dG", G.grad); g.check("dGh", Gh.grad)


def pathomic_args(**over):
    a = argparse.Namespace(
        mode="deformpathomic", act_type="Sigmoid", init_type="max", init_gain=0.02, fusion_type="concat",
        skip=0, use_bilinear=1, input_size_omic=431, input_size_omic_tumor=59, input_size_omic_immune=361,
        input_path_dim=1024, path_gate=1, omic_gate=1, path_dim=128, omic_dim=128, path_scale=1, omic_scale=1,
        mmhid=128, cut_fuse_grad=False, dropout_rate=0.1, return_grad="False", label_dim=4, task_type="diag2021",
        attn_dim=2, return_vgrid=True, batch_size=2, world_size=1)
    for k, v in over.items():
        setattr(a, k, v)
    return a


def test_full_model_reference_grid():
    """DeformPathomicNet + BatchLoss, B = 2, N = 2500 (the only size the reference supports)."""
    g = Golden("pathomic_ref50")
    net = smml.DeformPathomicNet(pathomic_args())
    params = params_for(net, 42, "pathomic")
    p = _req(params)
    B = 2
    x_path = synth.bag(B, 2500, 1024, 42, "pathomic:bag")
    x_t = synth.normal((B, 59), 42, "pathomic:tumor")
    x_i = synth.normal((B, 361), 42, "pathomic:immune")
    feats, vt, vi, lg = deform_pathomic_net(x_path, x_t, x_i, p, grid_hw=(50, 50))
    label = torch.tensor([1, 3])
    l_t, l_i = batch_loss(lg[3], lg[4], B), batch_loss(lg[5], lg[6], B)
    loss = torch.nn.functional.cross_entropy(lg[2], label) + 0.5 * l_t.sum() + 0.5 * l_i.sum()
    loss.backward()
    g.check("features", feats); g.check("vec_t", vt); g.check("vec_i", vi); g.check("haz", lg[2])
    g.check("vgrid_t", lg[4]); g.check("vgrid_i", lg[6]); g.check("batchloss_t", l_t); g.check("batchloss_i", l_i)
    assert abs(loss.item() - g.scalar("loss")) <= 1e-4 * abs(g.scalar("loss"))
    n_with_grad = sum(1 for k in g.keys("grad:"))
    assert n_with_grad == int(g.array("n_params_with_grad"))
    # ill-conditioned gradients (position-bias MLP: ReLU-gated sums over 2.9e7 pairs) carry their own fp32
    # noise in the fixture; Golden.check widens the tolerance to 8 x that noise
    _check_grads(g, p)


@pytest.mark.parametrize("tag,L,S,B", [("coattn_L4_S2500", 4, 2500, 2), ("coattn_L2500_S4", 2500, 4, 2), ("coattn_L200_S4096", 200, 4096, 1)])
def test_coattention(tag, L, S, B):
    from oracle.coattn import coattention
    g = Golden(tag)
    p = _req(params_for(smml.MultiheadAttention(256, 1), 42, tag))
    q = synth.normal((L, B, 256), 42, tag + ":q").requires_grad_()
    kv = synth.normal((S, B, 256), 42, tag + ":kv").requires_grad_()
    w_o = synth.normal((L, B, 256), 42, tag + ":wo"); w_r = synth.normal((B, 1, L, S), 42, tag + ":wr")
    out, raw = coattention(q, kv, kv, p)
    ((out * w_o).sum() + (raw * w_r).sum() * 1e-2).backward()
    g.check("out", out); g.check("raw", raw); g.check("dq", q.grad); g.check("dkv", kv.grad)
    _check_grads(g, p)


def bifusion_params(mod, tag):
    shapes = {k: tuple(v.shape) for k, v in mod.state_dict().items() if "num_batches" not in k}
    params = synth.fill_params(shapes, seed=42, tag=tag)
    for k in list(params):
        if k.endswith("running_var"):
            params[k] = params[k].abs() + 0.5
    return params


@pytest.mark.parametrize("tag,skip", [("bifusion_skip0", 0), ("bifusion_skip1", 1)])
def test_bilinear_fusion(tag, skip):
    from oracle.coattn import bilinear_fusion
    g = Golden(tag)
    mod = smml.BilinearFusion(skip=skip, use_bilinear=1, gate1=1, gate2=1, dim1=128, dim2=128, mmhid=128, dropout_rate=0.1)
    params = bifusion_params(mod, tag)
    p = {k: (v.clone().requires_grad_() if "running" not in k else v) for k, v in params.items()}
    v1 = synth.normal((4, 128), 42, tag + ":v1").requires_grad_(); v2 = synth.normal((4, 128), 42, tag + ":v2").requires_grad_()
    w = synth.normal((4, 128), 42, tag + ":w")
    out = bilinear_fusion(v1, v2, p, skip=skip); (out * w).sum().backward()
    g.check("out", out); g.check("dv1", v1.grad); g.check("dv2", v2.grad)
    _check_grads(g, p)


def cmta_inputs():
    B, n = 2, 150
    x_path = synth.bag(B, n, 1024, 42, "cmta:bag")
    x_omic = synth.normal((B, 431), 42, "cmta:omic")
    w = [synth.normal((B, 4), 42, "cmta:w0")] + [synth.normal((B, 256), 42, f"cmta:w{i}") for i in range(1, 5)]
    return x_path, x_omic, w


def cmta_params(mod):
    params = params_for(mod, 42, "cmta")
    for k in params:
        if k.endswith("cls_token"):
            params[k] = params[k] * 1e-3
    return params


CMTA_OUTPUTS = ("logits", "hazards", "S", "cls_p_enc", "cls_p_dec", "cls_g_enc", "cls_g_dec")


def test_cmta():
    """CMTA (Transformer_P / Transformer_G encoders and decoders, the co-attention pair, concat fusion) vs the reference."""
    from oracle.cmta import cmta
    g = Golden("cmta_n150")
    mod = smml.CMTA(argparse.Namespace(label_dim=4))
    p = _req(cmta_params(mod))
    x_path, x_omic, w = cmta_inputs()
    x_path.requires_grad_(); x_omic.requires_grad_()
    out = cmta(x_path, x_omic, p)
    loss = (out[0] * w[0]).sum() + sum((out[2 + i] * w[i]).sum() for i in range(1, 5))
    loss.backward()
    for nm, o in zip(CMTA_OUTPUTS, out):
        g.check(nm, o)
    g.check("dx_path", x_path.grad); g.check("dx_omic", x_omic.grad)
    assert abs(loss.item() - g.scalar("loss")) <= 1e-4 * abs(g.scalar("loss"))
    _check_grads(g, p)


def gradmod_case(seed):
    """Inputs of tests/golden/make_golden.py::gradmod_case (B = 8, C = 4, hs = 128)."""
    B, C, hs = 8, 4, 128
    tag = f"gradmod:{seed}"
    ft = synth.normal((B, hs), seed, tag + ":ft"); fi = synth.normal((B, hs), seed, tag + ":fi")
    W = synth.normal((C, 2 * hs), seed, tag + ":W") * 0.2; b = synth.normal((C,), seed, tag + ":b") * 0.1
    G = synth.normal((C, 2 * hs), seed, tag + ":G") * 0.05
    label = (synth.normal((B,), seed, tag + ":lab").abs() * 1.7).long().clamp(max=C - 1)
    return ft, fi, W, b, label, G


def test_gradient_modulation_block():
    """oracle/trainstep.py vs the output of the reference's own statements (train_test.py:87-184) on 8 synthetic cases
    that take every branch (no conflict, tumor half projected, immune half projected)."""
    from oracle.trainstep import gradient_modulate
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gradmod_b8.npz"))
    seen = set()
    for seed in range(1, 9):
        ft, fi, W, b, label, G = gradmod_case(seed)
        g, info = gradient_modulate(ft, fi, W, b, label, G)
        ref = torch.from_numpy(z[f"case{seed}/grad"])
        assert float((g - ref).abs().max()) <= 1e-6 * float(ref.abs().max()), seed
        assert abs(info["ratio_t"] - float(z[f"case{seed}/ratio_t"])) <= 1e-6 * info["ratio_t"]
        assert [bool(x) for x in info["branch"]] == list(z[f"case{seed}/changed_rows"])
        seen.update(info["branch"])
    assert seen == {0, 1, 2}


def test_imposed_decisions_reproduce_the_plain_oracle():
    """oracle.deform.DECISIONS (the hook the GPU parity tests use to impose the kernels' ReLU masks and sampler cells): with the
    oracle's OWN decisions imposed, values and gradients must equal the plain run bit for bit - 2-D and 1-D module."""
    import oracle.deform as od

    class Own:
        """decisions computed from the oracle's own fp32 evaluation (same interface as tests/helpers.py Decisions)"""
        def __init__(self, vs, gq, p, W, H):
            vx, vy = (vs[..., 0], vs[..., 1]) if vs.shape[-1] == 2 else (vs[..., 0], torch.zeros_like(vs[..., 0]))
            ix, iy, _ = sample_positions(vx, vy, W, H)
            self.cells = (torch.floor(ix).long(), torch.floor(iy).long())
            pos = od.signed_log(gq[None, :, None, :] - vs[:, None, :, :])
            x1 = pos @ p["rel_pos_bias.mlp.0.0.weight"].t() + p["rel_pos_bias.mlp.0.0.bias"]
            x2 = torch.relu(x1) @ p["rel_pos_bias.mlp.1.0.weight"].t() + p["rel_pos_bias.mlp.1.0.bias"]
            self.m1, self.m2 = x1 > 0, x2 > 0

        def relu_masks(self, i0, i1):
            return self.m1[:, i0:i1], self.m2[:, i0:i1]

    for dim in (2, 1):
        B, C = 2, 128
        if dim == 2:
            Hh, Ww = 12, 16
            N = Hh * Ww
            mod = smml.DeformCrossAttention2D(dim=C, grid_hw=(Hh, Ww))
            fn = lambda a, b, p: deform_cross_attention_2d(a, b, p, grid_hw=(Hh, Ww), q_chunk=50, return_aux=True)
        else:
            N = 61
            mod = smml.DeformCrossAttention1D(dim=C, downsample_factor=4, offset_scale=2, offset_kernel_size=6)
            fn = lambda a, b, p: deform_cross_attention_1d(a, b, p, offset_scale=2.0, q_chunk=50, return_aux=True)
        params = params_for(mod, 5, f"imposed{dim}")
        x1 = synth.normal((B, C, N), 5, f"imposed{dim}:x1"); x2 = synth.normal((B, C, N), 5, f"imposed{dim}:x2")
        wo = synth.normal((B, C, N), 5, f"imposed{dim}:wo")
        res = []
        dec = None
        for imposed in (False, True):
            p = {k: v.clone().requires_grad_() for k, v in params.items()}
            a, b = x1.clone().requires_grad_(), x2.clone().requires_grad_()
            od.DECISIONS = [dec] if imposed else None
            out, vg, aux = fn(a, b, p)
            assert not od.DECISIONS
            (out * wo).sum().backward()
            res.append([out.detach(), vg.detach(), a.grad, b.grad] + [p[k].grad for k in sorted(p) if p[k].grad is not None])
            if not imposed:
                with torch.no_grad():
                    if dim == 2:
                        vs = torch.stack((aux["vsx"], aux["vsy"]), dim=-1)
                        qx = 2.0 * torch.arange(Ww, dtype=torch.float32) / max(Hh - 1, 1) - 1.0
                        qy = 2.0 * torch.arange(Hh, dtype=torch.float32) / max(Ww - 1, 1) - 1.0
                        gq = torch.stack((qx.view(1, Ww).expand(Hh, Ww), qy.view(Hh, 1).expand(Hh, Ww)), dim=-1).reshape(N, 2)
                        dec = Own(vs, gq, params, Ww, Hh)
                    else:
                        gq = (2.0 * torch.arange(N, dtype=torch.float32) / max(N - 1, 1) - 1.0).view(N, 1)
                        dec = Own(aux["vs"].unsqueeze(-1), gq, params, 1, N)
        assert len(res[0]) == len(res[1]) > 10
        for i, (u, v) in enumerate(zip(*res)):
            assert torch.equal(u, v), f"{dim}-D module, tensor {i}: imposing the oracle's own decisions changed the result"


def test_concordance_index_known_answers():
    """The oracle's restatement of scikit-survival's concordance_index_censored (absent from this image: parity unpinned) on hand-checked
    cases: a perfectly ordered cohort, a reversed one, a tie in risk (one half), a censored sample at an event's time (comparable), two events
    at the same time (not comparable), a censored sample before every event (never the first of a pair)."""
    from oracle.trainstep import concordance_index_censored as cic
    assert cic([True, True, True], [1, 2, 3], [3.0, 2.0, 1.0])[0] == 1.0
    assert cic([True, True, True], [1, 2, 3], [1.0, 2.0, 3.0])[0] == 0.0
    c, con, dis, tie, comp = cic([True, True, False], [1, 2, 3], [2.0, 2.0, 1.0])
    assert (con, dis, tie, comp) == (2, 0, 1, 3) and abs(c - 2.5 / 3) < 1e-12
    # event and censoring at the same time: the event sample is comparable with the censored one
    c, con, dis, tie, comp = cic([True, False], [5, 5], [1.0, 0.0])
    assert (con, comp) == (1, 1) and c == 1.0
    # two events at the same time are not comparable; the later sample gives each one pair
    c, con, dis, tie, comp = cic([True, True, False], [5, 5, 9], [2.0, 1.0, 0.0])
    assert comp == 2 and c == 1.0
    # a censored sample never opens a pair
    c, con, dis, tie, comp = cic([False, True, True], [1, 2, 3], [0.0, 2.0, 1.0])
    assert comp == 1 and c == 1.0
    import pytest
    with pytest.raises(ZeroDivisionError):
        cic([False, True], [1, 2], [0.0, 1.0])

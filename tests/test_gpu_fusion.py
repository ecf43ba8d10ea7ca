"""GPU parity of the co-attention (a10), BilinearFusion (a11) and OrthogonalLoss (a13) rows against golden vectors
generated from the reference."""
import pytest
import torch

from helpers import Golden, params_for, smml, synth
from test_gpu_parity import _assert_close, _load
from test_oracle_golden import bifusion_params

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag,L,S,B", [("coattn_L4_S2500", 4, 2500, 2), ("coattn_L2500_S4", 2500, 4, 2), ("coattn_L200_S4096", 200, 4096, 1)])
def test_coattention_golden(cuda, tag, L, S, B):
    g = Golden(tag)
    mod = smml.MultiheadAttention(embed_dim=256, num_heads=1)
    mod = _load(mod, params_for(mod, 42, tag), cuda)
    q = synth.normal((L, B, 256), 42, tag + ":q").to(cuda).requires_grad_()
    kv = synth.normal((S, B, 256), 42, tag + ":kv").to(cuda).requires_grad_()
    w_o = synth.normal((L, B, 256), 42, tag + ":wo").to(cuda); w_r = synth.normal((B, 1, L, S), 42, tag + ":wr").to(cuda)
    out, raw = mod(q, kv, kv)
    assert out.shape == (L, B, 256) and raw.shape == (B, 1, L, S)
    ((out * w_o).sum() + (raw * w_r).sum() * 1e-2).backward()
    g.check("out", out); g.check("raw", raw); g.check("dq", q.grad); g.check("dkv", kv.grad)
    for k, p in mod.named_parameters():
        g.check("grad:" + k, p.grad, what="d" + k)


def test_coattention_masked_golden(cuda):
    """key_padding_mask + bool attn_mask against the reference's own output (tests/golden/make_golden.py::case_options)."""
    from test_oracle_golden import coattn_masks
    tag, L, S, B = "coattn_masked_L37_S50", 37, 50, 2
    g = Golden(tag)
    mod = smml.MultiheadAttention(embed_dim=256, num_heads=1)
    mod = _load(mod, params_for(mod, 42, tag), cuda)
    q = synth.normal((L, B, 256), 42, tag + ":q").to(cuda).requires_grad_()
    kv = synth.normal((S, B, 256), 42, tag + ":kv").to(cuda).requires_grad_()
    w_o = synth.normal((L, B, 256), 42, tag + ":wo").to(cuda)
    kpm, am = coattn_masks(tag, L, S, B)
    out, raw = mod(q, kv, kv, key_padding_mask=kpm.to(cuda), attn_mask=am.to(cuda))
    (out * w_o).sum().backward()
    fin = torch.isfinite(raw)
    assert int((~fin).sum()) == int(g.scalar("masked_count"))
    g.check("out", out); g.check("raw_finite", torch.where(fin, raw, torch.zeros_like(raw))); g.check("dq", q.grad); g.check("dkv", kv.grad)
    for k, p in mod.named_parameters():
        g.check("grad:" + k, p.grad, what="d" + k)


def test_coattention_float_mask_and_multi_head_vs_torch(cuda):
    """Additive float attn_mask, 3-D per-head mask and key_padding_mask with four heads against torch.nn.MultiheadAttention."""
    torch.manual_seed(1)
    ref = torch.nn.MultiheadAttention(64, 4)
    mod = smml.MultiheadAttention(64, 4)
    mod.load_state_dict(ref.state_dict())
    mod = mod.to(cuda).eval()
    L, S, B = 10, 33, 3
    q, k = torch.randn(L, B, 64), torch.randn(S, B, 64)
    kpm = torch.zeros(B, S, dtype=torch.bool); kpm[:, -4:] = True
    for am in (torch.randn(L, S), torch.randn(B * 4, L, S) > 1.0):
        o_ref, w_ref = ref(q, k, k, need_weights=True, key_padding_mask=kpm, attn_mask=am)
        o, w = mod(q.to(cuda), k.to(cuda), k.to(cuda), need_raw=False, key_padding_mask=kpm.to(cuda), attn_mask=am.to(cuda))
        _assert_close("masked mha out", o, o_ref, 1e-5); _assert_close("masked mha weights", w, w_ref, 1e-5)


@pytest.mark.parametrize("bias_kv,zero_attn", [(True, False), (False, True), (True, True)])
def test_coattention_bias_kv_and_zero_attn_vs_torch(cuda, bias_kv, zero_attn):
    """add_bias_kv / add_zero_attn (MultiheadAttention.py:236-243,271-279 - the reference's file is a copy of torch's functional form, so
    torch.nn.MultiheadAttention is the pin), with masks, four heads: outputs, averaged weights and every gradient."""
    torch.manual_seed(2)
    ref = torch.nn.MultiheadAttention(64, 4, add_bias_kv=bias_kv, add_zero_attn=zero_attn)
    mod = smml.MultiheadAttention(64, 4, add_bias_kv=bias_kv, add_zero_attn=zero_attn)
    mod.load_state_dict(ref.state_dict())
    mod = mod.to(cuda).eval()
    L, S, B = 9, 21, 3
    q, k = torch.randn(L, B, 64), torch.randn(S, B, 64)
    kpm = torch.zeros(B, S, dtype=torch.bool); kpm[:, -3:] = True
    am = torch.randn(L, S) > 1.0
    wo = torch.randn(L, B, 64)
    qr, kr = q.clone().requires_grad_(), k.clone().requires_grad_()
    o_ref, w_ref = ref(qr, kr, kr, need_weights=True, key_padding_mask=kpm, attn_mask=am)
    (o_ref * wo).sum().backward()
    qd, kd = q.to(cuda).requires_grad_(), k.to(cuda).requires_grad_()
    o, w = mod(qd, kd, kd, need_raw=False, key_padding_mask=kpm.to(cuda), attn_mask=am.to(cuda))
    (o * wo.to(cuda)).sum().backward()
    _assert_close("out", o, o_ref, 1e-5); _assert_close("weights", w, w_ref, 1e-5)
    _assert_close("dq", qd.grad, qr.grad, 1e-5); _assert_close("dk", kd.grad, kr.grad, 1e-5)
    rp = dict(ref.named_parameters())
    for name, p in mod.named_parameters():
        _assert_close("d" + name, p.grad, rp[name].grad, 1e-5)
    _, raw = mod(qd, kd, kd, key_padding_mask=kpm.to(cuda), attn_mask=am.to(cuda))
    assert raw.shape == (B, 4, L, S + int(bias_kv) + int(zero_attn))


def test_coattention_separate_kv_dims_vs_torch(cuda):
    """kdim / vdim != embed_dim (separate q / k / v projection weights, MultiheadAttention.py:372-379) against torch.nn.MultiheadAttention."""
    torch.manual_seed(3)
    ref = torch.nn.MultiheadAttention(64, 2, kdim=48, vdim=80)
    mod = smml.MultiheadAttention(64, 2, kdim=48, vdim=80)
    mod.load_state_dict(ref.state_dict())
    mod = mod.to(cuda).eval()
    L, S, B = 7, 19, 2
    q, k, v = torch.randn(L, B, 64), torch.randn(S, B, 48), torch.randn(S, B, 80)
    wo = torch.randn(L, B, 64)
    qr, kr, vr = (t.clone().requires_grad_() for t in (q, k, v))
    o_ref, w_ref = ref(qr, kr, vr, need_weights=True)
    (o_ref * wo).sum().backward()
    qd, kd, vd = (t.to(cuda).requires_grad_() for t in (q, k, v))
    o, w = mod(qd, kd, vd, need_raw=False)
    (o * wo.to(cuda)).sum().backward()
    _assert_close("out", o, o_ref, 1e-5); _assert_close("weights", w, w_ref, 1e-5)
    for a, b, n in ((qd, qr, "dq"), (kd, kr, "dk"), (vd, vr, "dv")):
        _assert_close(n, a.grad, b.grad, 1e-5)
    rp = dict(ref.named_parameters())
    for name, p in mod.named_parameters():
        _assert_close("d" + name, p.grad, rp[name].grad, 1e-5)


def test_coattention_multi_head_vs_torch(cuda):
    torch.manual_seed(0)
    ref = torch.nn.MultiheadAttention(64, 4)
    mod = smml.MultiheadAttention(64, 4)
    mod.load_state_dict(ref.state_dict())
    mod = mod.to(cuda).eval()
    q, k = torch.randn(10, 3, 64), torch.randn(33, 3, 64)
    o_ref, w_ref = ref(q, k, k, need_weights=True)
    o, w = mod(q.to(cuda), k.to(cuda), k.to(cuda), need_raw=False)
    _assert_close("mha out", o, o_ref, 1e-5); _assert_close("mha weights", w, w_ref, 1e-5)


@pytest.mark.parametrize("tag,skip", [("bifusion_skip0", 0), ("bifusion_skip1", 1)])
def test_bilinear_fusion_golden(cuda, tag, skip):
    g = Golden(tag)
    mod = smml.BilinearFusion(skip=skip, use_bilinear=1, gate1=1, gate2=1, dim1=128, dim2=128, mmhid=128, dropout_rate=0.1)
    sd = dict(mod.state_dict()); sd.update(bifusion_params(mod, tag)); mod.load_state_dict(sd)
    mod = mod.to(cuda).eval()
    v1 = synth.normal((4, 128), 42, tag + ":v1").to(cuda).requires_grad_(); v2 = synth.normal((4, 128), 42, tag + ":v2").to(cuda).requires_grad_()
    w = synth.normal((4, 128), 42, tag + ":w").to(cuda)
    out = mod(v1, v2); (out * w).sum().backward()
    g.check("out", out); g.check("dv1", v1.grad); g.check("dv2", v2.grad)
    for k, p in mod.named_parameters():
        g.check("grad:" + k, p.grad, what="d" + k)


def test_orthogonal_loss_golden(cuda):
    g = Golden("orthloss_b4")
    P, Ph, G, Gh = (synth.normal((4, 256), 42, "ol:" + t).to(cuda).requires_grad_() for t in "abcd")
    ol = smml.OrthogonalLoss()(P, Ph, G, Gh)
    ol.sum().backward()
    g.check("out", ol); g.check("dP", P.grad); g.check("dPh", Ph.grad); g.check("dG", G.grad); g.check("dGh", Gh.grad)

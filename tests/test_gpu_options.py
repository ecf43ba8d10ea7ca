"""GPU: the corrected-semantics switches.  All are OFF by default - the default paths are pinned to the
reference by the golden tests - so these tests check (a) HIP vs the oracle's restatement of the corrected semantics and
(b) the property each switch is there for."""
import argparse

import pytest
import torch

import oracle.deform as odeform
from helpers import assert_calibrated, assert_close, decision_tap, params_for, smml, synth
from oracle.deform import deform_cross_attention_1d, deform_cross_attention_2d, sample_positions
from oracle.nystrom import nystrom_attention
from test_gpu_parity import _compare_param_grads, _load, cpb_probe

pytestmark = pytest.mark.gpu
Fh = smml.functional


def test_consistent_grid_norm_2d(cuda):
    B, Hh, Ww, C = 2, 20, 28, 128
    N = Hh * Ww
    tag = "opt:cgn"
    mod = smml.DeformCrossAttention2D(dim=C, dropout=0.1, grid_hw=(Hh, Ww), consistent_grid_norm=True)
    params = params_for(mod, 31, tag)
    mod = _load(mod, params, cuda)
    x1 = synth.normal((B, C, N), 31, tag + ":x1"); x2 = synth.normal((B, C, N), 31, tag + ":x2")
    wo = synth.normal((B, C, N), 31, tag + ":wo")
    ad, bd = x1.to(cuda).requires_grad_(), x2.to(cuda).requires_grad_()
    with decision_tap() as tap:
        o, vg = mod(ad, bd, return_vgrid=True)
    (o * wo.to(cuda)).sum().backward()
    run = {}
    with cpb_probe() as probe:
        for dt in (torch.float32, torch.float64):
            pr = {k: v.clone().to(dt).requires_grad_() for k, v in params.items()}
            a, b = x1.clone().to(dt).requires_grad_(), x2.clone().to(dt).requires_grad_()
            odeform.DECISIONS = tap.decisions()
            o_r, vg_r = deform_cross_attention_2d(a, b, pr, grid_hw=(Hh, Ww), consistent_grid_norm=True)
            (o_r * wo.to(dt)).sum().backward()
            run[dt] = (o_r, vg_r, a.grad, b.grad, pr)
    r32, r64 = run[torch.float32], run[torch.float64]
    for name, got, i in (("out", o, 0), ("vgrid", vg, 1), ("dx1", ad.grad, 2), ("dx2", bd.grad, 3)):
        assert_calibrated("cgn " + name, got, r32[i], r64[i])
    _compare_param_grads(mod, r32[4], r64[4], probe=probe)
    # what the switch is for: with zero offsets every sample sits on the centre of its r x r block, inside the map
    th, tw = vg.shape[-2:]
    gx = torch.arange(tw, dtype=torch.float32).view(1, tw).expand(th, tw).reshape(1, -1)
    gy = torch.arange(th, dtype=torch.float32).view(th, 1).expand(th, tw).reshape(1, -1)
    _, _, corners = sample_positions((2 * gx + 1) / tw - 1, (2 * gy + 1) / th - 1, Ww, Hh)
    assert all(bool(c[3].all()) for c in corners), "pixel-centre positions must have all four corners in bounds"
    _, _, corners_ref = sample_positions(2 * gx / max(th - 1, 1) - 1, 2 * gy / max(tw - 1, 1) - 1, Ww, Hh)
    assert not all(bool(c[3].all()) for c in corners_ref), "the reference normalisation puts corners outside the map"


def test_true_1d_sampling(cuda):
    B, n, C = 2, 130, 128
    tag = "opt:t1d"
    mod = smml.DeformCrossAttention1D(dim=C, downsample_factor=4, offset_scale=2, offset_kernel_size=6, true_1d_sampling=True)
    params = params_for(mod, 33, tag)
    mod = _load(mod, params, cuda)
    x1 = synth.normal((B, C, n), 33, tag + ":x1"); x2 = synth.normal((B, C, n), 33, tag + ":x2")
    wo = synth.normal((B, C, n), 33, tag + ":wo")
    ad, bd = x1.to(cuda).requires_grad_(), x2.to(cuda).requires_grad_()
    with decision_tap() as tap:
        o, vg = mod(ad, bd, return_vgrid=True)
    (o * wo.to(cuda)).sum().backward()
    run = {}
    with cpb_probe() as probe:
        for dt in (torch.float32, torch.float64):
            pr = {k: v.clone().to(dt).requires_grad_() for k, v in params.items()}
            a, b = x1.clone().to(dt).requires_grad_(), x2.clone().to(dt).requires_grad_()
            odeform.DECISIONS = tap.decisions()
            o_r, vg_r = deform_cross_attention_1d(a, b, pr, offset_scale=2.0, true_1d_sampling=True)
            (o_r * wo.to(dt)).sum().backward()
            run[dt] = (o_r, vg_r, a.grad, b.grad, pr)
    r32, r64 = run[torch.float32], run[torch.float64]
    for name, got, i in (("out", o, 0), ("vgrid", vg, 1), ("dx1", ad.grad, 2), ("dx2", bd.grad, 3)):
        assert_calibrated("t1d " + name, got, r32[i], r64[i])
    _compare_param_grads(mod, r32[4], r64[4], probe=probe)
    # what the switch is for: the keys depend on MORE than the centre token (in the reference's layout d out / d x2 is
    # non-zero at the centre token only)
    touched = (bd.grad.abs().sum(dim=(0, 1)) > 0).sum().item()
    assert touched > n // 2
    ref_mod = smml.DeformCrossAttention1D(dim=C, downsample_factor=4, offset_scale=2, offset_kernel_size=6)
    ref_mod = _load(ref_mod, params, cuda)
    b2 = x2.to(cuda).requires_grad_()
    (ref_mod(x1.to(cuda), b2) * wo.to(cuda)).sum().backward()
    assert (b2.grad.abs().sum(dim=(0, 1)) > 0).sum().item() <= 2


def test_per_bag_pinv_scale(cuda):
    B, n, dim, dh, m = 3, 200, 128, 16, 64
    tag = "opt:pbp"
    mod = smml.NystromAttention(dim=dim, dim_head=dh, heads=8, num_landmarks=m, per_bag_pinv_scale=True)
    params = params_for(mod, 35, tag)
    mod = _load(mod, params, cuda)
    x = synth.normal((B, n, dim), 35, tag + ":x") * torch.tensor([0.3, 1.0, 2.5]).view(3, 1, 1)
    wo = synth.normal((B, n, dim), 35, tag + ":wo")
    run = {}
    for dt in (torch.float32, torch.float64):
        pr = {k: v.clone().to(dt).requires_grad_() for k, v in params.items()}
        xr = x.clone().to(dt).requires_grad_()
        o = nystrom_attention(xr, pr, heads=8, dim_head=dh, num_landmarks=m, per_bag_pinv_scale=True)
        (o * wo.to(dt)).sum().backward()
        run[dt] = (o, xr.grad, pr)
    xd = x.to(cuda).requires_grad_()
    out = mod(xd)
    (out * wo.to(cuda)).sum().backward()
    r32, r64 = run[torch.float32], run[torch.float64]
    assert_calibrated("pbp out", out, r32[0], r64[0]); assert_calibrated("pbp dx", xd.grad, r32[1], r64[1])
    for k, p in mod.named_parameters():
        assert_calibrated("pbp d" + k, p.grad, r32[2][k].grad, r64[2][k].grad)
    # what the switch is for: a bag's output no longer depends on its batch mates
    with torch.no_grad():
        alone = mod(x[1:2].to(cuda))
        assert_close("bag independent of its batch", out[1:2].detach(), alone, 1e-5)
        ref_mod = _load(smml.NystromAttention(dim=dim, dim_head=dh, heads=8, num_landmarks=m), params, cuda)
        coupled = float((ref_mod(x.to(cuda))[1:2] - ref_mod(x[1:2].to(cuda))).abs().max())
        assert coupled > 1e-6, "with the reference's batch-global max the bags are coupled"


def test_wrap_pad_to_square(cuda):
    """A bag of 390 instances (not a square) through the 2-D branch: equals the run on the explicitly wrap-padded 20 x 20
    bag, cropped; without the switch the module refuses."""
    args = argparse.Namespace(path_dim=128, attn_dim=2, return_vgrid=True, input_path_dim=64, wrap_pad_to_square=True)
    mil = smml.DeformCrossTransMIL(args)
    params = params_for(mil, 37, "opt:wrap")
    mil = _load(mil, params, cuda)
    B, N = 2, 390
    path = synth.bag(B, N, 64, 37, "opt:wrap:bag").to(cuda); omic = torch.relu(synth.normal((B, 128), 37, "opt:wrap:omic")).to(cuda)
    enc, logits, _, omic_t, vg = mil(path, omic)
    assert omic_t.shape == (B, N, 128) and vg.shape == (B * 8, 2, 5, 5)
    # reference construction by hand: fc1 -> fusion on the N tokens, wrap-pad both streams, attention on 20 x 20, crop, pool
    with torch.no_grad():
        p = Fh.linear(path, mil._fc1[0].weight, mil._fc1[0].bias, act=Fh.ACT_RELU)
        h = mil.fusion_layer(p, omic)
        h2 = torch.cat((h, h[:, :10]), 1); p2 = torch.cat((p, p[:, :10]), 1)
        h2 = mil.layer3(h2, p2, 2, False)[:, :N]
        avg = Fh.layer_norm_token_mean(h2, mil.norm.weight, mil.norm.bias, mil.norm.eps)
        e2 = Fh.linear(Fh.linear(avg, mil.pooler.dense.weight, mil.pooler.dense.bias, act=Fh.ACT_TANH),
                       mil.multimodal_projection.weight, mil.multimodal_projection.bias)
    assert_close("wrap-pad encoded", enc.detach(), e2, 1e-6)
    enc.sum().backward()
    assert all(torch.isfinite(q.grad).all() for q in mil.parameters() if q.grad is not None)
    args2 = argparse.Namespace(path_dim=128, attn_dim=2, return_vgrid=True, input_path_dim=64)
    mil2 = _load(smml.DeformCrossTransMIL(args2), params, cuda)
    with pytest.raises(ValueError):
        mil2(path, omic)


def test_dropout_mask_changes_between_graph_replays(cuda):
    """A training-mode DeformCrossAttention2D captured in a hipGraph: the host seed is baked into the captured launches, the
    device-resident offset (include/smml.h: SmmlDeformOpts.seed_offset; functional.graph_seed_offset) makes every replay draw
    a new dropout mask, and forward / backward of one call agree on it (the gradient of a replay matches an eager call that is
    given the replay's effective seed)."""
    import torch
    Fh = smml.functional
    torch.manual_seed(3)
    gen = torch.Generator().manual_seed(11)
    B, N, J, H, G, PD = 1, 96, 40, 8, 8, 2
    rn = lambda *s: torch.randn(*s, generator=gen)
    t = dict(q=rn(B, N, 512) * 0.4, k=rn(B, J, 512) * 0.4, v=rn(B, J, 512), vs=torch.rand(B * G, J, PD, generator=gen) * 2.4 - 1.2,
             gq=torch.rand(N, PD, generator=gen) * 2 - 1, w1=rn(32, PD) * 0.7, b1=rn(32) * 0.3, w2=rn(32, 32) * 0.25, b2=rn(32) * 0.2,
             w3=rn(H // G, 32) * 0.3, b3=rn(H // G) * 0.1)
    names = ("q", "k", "v", "vs", "gq", "w1", "b1", "w2", "b2", "w3", "b3")
    dev = {n: x.to(cuda).requires_grad_(n == "q") for n, x in t.items()}
    Fh.graph_seed_offset(cuda, allocate_only=True)                       # what the first eager dropout call of a model does
    out_s = torch.zeros(B, N, 512, device=cuda); dq_s = torch.zeros(B, N, 512, device=cuda); off_s = torch.zeros(1, device=cuda, dtype=torch.int64)
    seed = 12345
    # the half-width of the region tables is given (as the modules do): a data-derived one would need a host sync, which a capture forbids
    # (such a call keeps the per-pair kernels) - the replay and the eager calls below then run the SAME kernels, region path included
    pm = dict(cpb_region_pmax=Fh.table_pmax(1.0, 1.2))

    def call():
        off = Fh.graph_seed_offset(cuda)
        o = Fh.deform_attention(*(dev[n] for n in names), heads=H, groups=G, scale=0.125, dropout_p=0.25, dropout_seed=seed,
                                dropout_seed_offset=off, **pm)
        dev["q"].grad = None
        o.sum().backward()
        out_s.copy_(o.detach()); dq_s.copy_(dev["q"].grad); off_s.copy_(off)

    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        g = torch.cuda.CUDAGraph()
        # warm-up outside the capture (no offset in eager mode)
        o = Fh.deform_attention(*(dev[n] for n in names), heads=H, groups=G, scale=0.125, dropout_p=0.25, dropout_seed=seed, **pm)
        o.sum().backward()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            call()
    torch.cuda.current_stream().wait_stream(s)
    seen = []
    for _ in range(3):
        g.replay(); torch.cuda.synchronize()
        off = int(off_s)
        seen.append((off, out_s.clone(), dq_s.clone()))
        # an eager call with the effective seed reproduces the replay (forward and backward agree on the mask)
        dev["q"].grad = None
        o = Fh.deform_attention(*(dev[n] for n in names), heads=H, groups=G, scale=0.125, dropout_p=0.25,
                                dropout_seed=_mix64(seed ^ _mix64(off)), **pm)
        o.sum().backward()
        assert torch.equal(o.detach(), out_s), "forward of the replay differs from the eager call with its effective seed"
        assert_close("dq of a graph replay vs eager with the same effective seed", dq_s, dev["q"].grad, 1e-6)
    assert seen[0][0] != seen[1][0] != seen[2][0]
    assert not torch.equal(seen[0][1], seen[1][1]) and not torch.equal(seen[1][1], seen[2][1]), "replays must draw different masks"


def _mix64(z):
    """splitmix64 finaliser, as csrc/deform_attn.hip mix64"""
    M = (1 << 64) - 1
    z = (z + 0x9E3779B97F4A7C15) & M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    return z ^ (z >> 31)


def test_dropout_masks_of_graph_replays_are_uncorrelated(cuda):
    """with the replay offset ADDED to the seed the mask of replay r was the mask of replay 0 shifted by (calls per step)
    x r elements along the key axis.  The offset is now hashed into the key: masks of different offsets must agree with each
    other - at every small index shift - no more often than independent masks do (p^2 + (1 - p)^2), and keep the right density."""
    Fh = smml.functional
    B, N, J, H, p, seed = 1, 300, 80, 8, 0.5, 777
    masks = []
    for off in (0, 1, 2, 5):
        m = Fh.deform_attention_dropout_mask(B, N, J, H, p, seed, cuda, seed_offset=torch.tensor([off], device=cuda, dtype=torch.int64))
        assert abs(float(m.mean()) - (1 - p)) < 0.01
        masks.append(m.flatten().bool())
    n = masks[0].numel()
    for a in range(len(masks)):
        for b in range(a + 1, len(masks)):
            for shift in range(-12, 13):
                lo, hi = max(0, shift), min(n, n + shift)
                agree = float((masks[a][lo - shift:hi - shift] == masks[b][lo:hi]).float().mean())
                assert abs(agree - 0.5) < 0.01, f"offsets {a},{b} shift {shift}: masks agree on {agree:.3f} of the elements"


def test_accumulated_gradients_under_graph_replay(cuda):
    """Weight / bias gradients are accumulated with atomics into zero-initialised buffers; the buffers of small tensors come from a pre-zeroed
    pool (functional._ZeroPool).  Inside a hipGraph capture the pool must not be used - a pooled slice is zero only once - so every replay of a
    captured forward + backward has to return the same gradients as an eager call, not a running sum."""
    import torch
    Fh = smml.functional
    gen = torch.Generator().manual_seed(21)
    x = (torch.randn(4, 300, 128, generator=gen) * 0.5).to(cuda)
    w = (torch.randn(64, 128, generator=gen) * 0.1).to(cuda).requires_grad_()
    bias = (torch.randn(64, generator=gen) * 0.1).to(cuda).requires_grad_()
    gam = torch.ones(64, device=cuda, requires_grad=True); bet = torch.zeros(64, device=cuda, requires_grad=True)
    outs = [torch.zeros_like(t) for t in (w, bias, gam, bet)]

    def step():
        for t in (w, bias, gam, bet):
            t.grad = None
        y = Fh.layer_norm(Fh.linear(x, w, bias), gam, bet)
        (y * y).sum().backward()
        for o, t in zip(outs, (w, bias, gam, bet)):
            o.copy_(t.grad)

    step(); torch.cuda.synchronize()
    ref = [o.clone() for o in outs]
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            step()
    torch.cuda.current_stream().wait_stream(s)
    for r in range(3):
        g.replay(); torch.cuda.synchronize()
        for name, o, e in zip(("dW", "dbias", "dgamma", "dbeta"), outs, ref):
            assert_close(f"{name} of graph replay {r} vs eager", o, e, 1e-5)

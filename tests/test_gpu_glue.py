"""GPU unit tests of the glue pieces that the model-level parity runs only cover indirectly:
  * functional.GradFork - the offsets network's backward ADDS its gradient of q into the buffer the attention's backward returned
    instead of letting autograd add two [B, N, 512] tensors: results must be identical with the fork on and off, for the 2-D module
    and for the functional API with a leaf q;
  * functional.dual_linear_relu - both branches' relu(_fc1(bag)) in one batched launch per direction - against two linear() calls;
  * functional.batchloss_tail - everything of BatchLoss after its two Gram products in one launch per direction - against the torch-level
    tail (the reference's op sequence, utils/loss.py:26-40)."""
import pytest
import torch

from helpers import assert_close, smml, synth

pytestmark = pytest.mark.gpu
Fh = smml.functional


def test_grad_fork_on_and_off_give_the_same_gradients(cuda):
    """2-D module: d x1 (through to_q's two consumers) and every parameter gradient, fork on vs off; functional API: a LEAF q consumed by
    offsets() and deform_attention()."""
    import importlib
    da = importlib.import_module(smml.__name__ + ".deform_attention")
    torch.manual_seed(5)
    mod = smml.DeformCrossAttention2D(dim=128, dropout=0.0).to(cuda).train()
    B, S = 2, 20
    x1 = torch.randn(B, 128, S * S, device=cuda) * 0.5
    x2 = torch.randn(B, 128, S * S, device=cuda) * 0.5
    wo = torch.randn(B, 128, S * S, device=cuda)
    res = {}
    for fork in (True, False):
        da._GRAD_FORK = fork
        try:
            a = x1.clone().requires_grad_()
            b = x2.clone().requires_grad_()
            mod.zero_grad(set_to_none=True)
            out, vgrid = mod(a, b, return_vgrid=True)
            ((out * wo).sum() + 1e-2 * vgrid.pow(2).sum()).backward()
            res[fork] = {"dx1": a.grad.clone(), "dx2": b.grad.clone(), **{"d" + k: p.grad.clone() for k, p in mod.named_parameters() if p.grad is not None}}
        finally:
            da._GRAD_FORK = True
    assert res[True].keys() == res[False].keys()
    for k in res[True]:
        # the same kernels compute the same contributions; only WHERE the two gradients of q are added differs (in the offsets kernel's
        # gather vs an elementwise add): fp32 addition of two terms is commutative, so the results agree to the last bit or an ulp of
        # the atomically reduced weight gradients
        assert_close("fork on vs off " + k, res[True][k], res[False][k], 1e-6)

    # functional API, leaf q
    G, H, dg = 8, 8, 64
    gen = torch.Generator().manual_seed(9)
    q0 = torch.randn(B, S, S, G * dg, generator=gen).to(cuda) * 0.4
    w0 = (torch.randn(dg, 1, 6, 6, generator=gen) * 0.15).to(cuda).requires_grad_()
    b0 = (torch.randn(dg, generator=gen) * 0.1).to(cuda).requires_grad_()
    w2 = (torch.randn(2, dg, generator=gen) * 0.1).to(cuda).requires_grad_()
    cp = [t.to(cuda) for t in (torch.randn(32, 2, generator=gen) * 0.7, torch.randn(32, generator=gen) * 0.3, torch.randn(32, 32, generator=gen) * 0.25,
                               torch.randn(32, generator=gen) * 0.2, torch.randn(1, 32, generator=gen) * 0.3, torch.randn(1, generator=gen) * 0.1)]
    ax = 2.0 * torch.arange(S, dtype=torch.float32) / (S - 1) - 1.0
    gq = torch.stack((ax.view(1, S).expand(S, S), ax.view(S, 1).expand(S, S)), dim=-1).reshape(S * S, 2).contiguous().to(cuda)
    got = {}
    for use_fork in (True, False):
        q = q0.clone().requires_grad_()
        fork = Fh.GradFork() if use_fork else None
        vgrid, vs = Fh.offsets(q, w0, b0, w2, groups=G, ks=6, r=4, posdim=2, offset_scale=4.0, fork=fork)
        J = vs.shape[1]
        kv = torch.randn(B, J, H * 64, generator=torch.Generator().manual_seed(1)).to(cuda)
        o = Fh.deform_attention(q.view(B, S * S, -1), kv, kv * 0.5, vs, gq, *cp, heads=H, groups=G, scale=0.125, fork=fork,
                                cpb_region_pmax=Fh.table_pmax(1.0, 2.0))
        for t in (w0, b0, w2):
            t.grad = None
        (o.sum() + 1e-2 * vgrid.pow(2).sum()).backward()
        got[use_fork] = (q.grad.clone(), w0.grad.clone(), w2.grad.clone())
    for name, u, v in zip(("dq (leaf)", "d w0", "d w2"), got[True], got[False]):
        assert_close("functional fork on vs off " + name, u, v, 1e-6)


def test_dual_linear_relu_matches_two_linear_calls(cuda):
    gen = torch.Generator().manual_seed(3)
    B, N, K, C = 2, 777, 512, 128
    x = torch.clamp(torch.randn(B, N, K, generator=gen) * 0.5, min=0).to(cuda)
    ws = [(torch.randn(C, K, generator=gen) / K ** 0.5).to(cuda).requires_grad_() for _ in range(2)]
    bs = [(torch.randn(C, generator=gen) * 0.1).to(cuda).requires_grad_() for _ in range(2)]
    wo = [torch.randn(B, N, C, generator=gen).to(cuda) for _ in range(2)]
    y0, y1 = Fh.dual_linear_relu(x, ws[0], bs[0], ws[1], bs[1])
    ((y0 * wo[0]).sum() + (y1 * wo[1]).sum()).backward()
    got = [t.grad.clone() for t in (*ws, *bs)]
    for t in (*ws, *bs):
        t.grad = None
    r0 = Fh.linear(x, ws[0], bs[0], act=Fh.ACT_RELU)
    r1 = Fh.linear(x, ws[1], bs[1], act=Fh.ACT_RELU)
    ((r0 * wo[0]).sum() + (r1 * wo[1]).sum()).backward()
    assert_close("dual_linear_relu y0", y0, r0, 1e-6)
    assert_close("dual_linear_relu y1", y1, r1, 1e-6)
    for name, g, t in zip(("dw0", "dw1", "db0", "db1"), got, (*ws, *bs)):
        assert_close("dual_linear_relu " + name, g, t.grad, 2e-5)          # split-K weight gradients: atomically reduced, order differs


def test_batchloss_tail_matches_the_torch_tail(cuda):
    """BatchLoss with the fused tail against BatchLoss with the torch-level tail (SMML_BATCHLOSS_TAIL=0 path), forward and backward."""
    import importlib
    losses = importlib.import_module(smml.__name__ + ".losses")
    gen = torch.Generator().manual_seed(4)
    for Nb in (2, 8, 17):
        omic0 = torch.randn(Nb, 300, 128, generator=gen).to(cuda)
        vgrid0 = torch.randn(Nb * 8, 2, 6, 6, generator=gen).to(cuda)
        res = {}
        for fused in (True, False):
            losses._FUSED_TAIL = fused
            try:
                omic, vgrid = omic0.clone().requires_grad_(), vgrid0.clone().requires_grad_()
                out = smml.BatchLoss(Nb, 1)(omic, vgrid)
                assert out.shape == (Nb, Nb)
                w = torch.linspace(0.5, 1.5, Nb * Nb, device=cuda).view(Nb, Nb)
                (out * w).sum().backward()
                res[fused] = (out.detach().clone(), omic.grad.clone(), vgrid.grad.clone())
            finally:
                losses._FUSED_TAIL = True
        for name, u, v in zip(("loss matrix", "d omic", "d vgrid"), res[True], res[False]):
            assert_close(f"batchloss_tail N_b={Nb} {name}", u, v, 2e-5)

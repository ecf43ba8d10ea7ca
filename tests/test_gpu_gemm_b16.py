"""GPU: the bf16-storage GEMM (csrc/gemm_b16.hip) against an fp64 product of the SAME bf16 operands (the kernel's only arithmetic
difference is fp32 accumulation order: 1e-5 relative; a bf16 result is additionally rounded once: one ulp = 2^-8 relative), ragged
tiles, K tails, split-K, and linear_b16's forward / backward against an fp64 linear of the bf16-rounded tensors."""
import pytest
import torch

from helpers import smml

pytestmark = pytest.mark.gpu
Fh = smml.functional


def _rand(shape, seed):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * 0.5).to(torch.bfloat16)


def _check(got, ref, out_bf16, what):
    scale = ref.abs().max().clamp_min(1e-30)
    err = float((got.double().cpu() - ref).abs().max() / scale)
    tol = 2.0 ** -8 if out_bf16 else 2e-5
    assert err <= tol, (what, err, tol)


@pytest.fixture(params=[1, 2], ids=["tile128", "tile256"])
def tile(request):
    """both kernels: 1 = the 128 x 128 tile only, 2 = the 256-row eight-wave tile wherever M >= 256 and N >= 128"""
    L = smml.lib()
    L.smml_gemm_b16_set_tile(request.param)
    yield request.param
    L.smml_gemm_b16_set_tile(-1)


@pytest.mark.parametrize("out_bf16", [False, True])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 200, 72), (1, 8, 8), (1024, 1536, 512), (257, 136, 1000), (4096, 512, 1536), (700, 520, 136)])
def test_gemm_b16_nt(cuda, tile, M, N, K, out_bf16):
    a, b = _rand((M, K), M + K), _rand((N, K), N + 7 * K)
    bias = torch.randn(N, generator=torch.Generator().manual_seed(3))
    ref = a.double() @ b.double().t() + bias.double()
    c = torch.full((M, N), float("nan"), device=cuda, dtype=torch.bfloat16 if out_bf16 else torch.float32)
    Fh.gemm_b16(a.to(cuda), b.to(cuda), c, M=M, N=N, K=K, lda=K, ldb=K, ldc=N, bias=bias.to(cuda))
    _check(c, ref, out_bf16, ("nt", M, N, K))
    # the identity with an asymmetric second operand: catches a transposed accumulator map
    if K == N and M == K:
        eye = torch.eye(K).to(torch.bfloat16)
        Fh.gemm_b16(eye.to(cuda), b.to(cuda), c, M=M, N=N, K=K, lda=K, ldb=K, ldc=N)
        assert torch.equal(c.float().cpu(), b.float().t())


@pytest.mark.parametrize("splitk", [0, 1, 3, 16])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (264, 136, 1000), (8, 8, 1), (512, 1536, 5000), (512, 512, 4096), (776, 264, 333)])
def test_gemm_b16_tn(cuda, tile, M, N, K, splitk):
    a, b = _rand((K, M), M + K), _rand((K, N), N + 7 * K)
    ref = a.double().t() @ b.double()
    c = torch.zeros(M, N, device=cuda)
    Fh.gemm_b16(a.to(cuda), b.to(cuda), c, M=M, N=N, K=K, lda=M, ldb=N, ldc=N, trans=True, splitk=splitk)
    _check(c, ref, False, ("tn", M, N, K, splitk))
    if splitk == 1:                         # (0 = the library's choice: needs the zeroed fp32 output like any split)
        cb = torch.empty(M, N, device=cuda, dtype=torch.bfloat16)
        Fh.gemm_b16(a.to(cuda), b.to(cuda), cb, M=M, N=N, K=K, lda=M, ldb=N, ldc=N, trans=True)
        _check(cb, ref, True, ("tn bf16", M, N, K))


def test_gemm_b16_rejects_bad_operands(cuda):
    a, b = _rand((16, 12), 1).to(cuda), _rand((16, 12), 2).to(cuda)
    c = torch.empty(16, 16, device=cuda)
    with pytest.raises(RuntimeError):
        Fh.gemm_b16(a, b, c, M=16, N=16, K=12, lda=12, ldb=12, ldc=16)            # K % 8
    with pytest.raises(RuntimeError):
        Fh.gemm_b16(a.float(), b, c, M=16, N=16, K=12, lda=12, ldb=12, ldc=16)    # fp32 operand
    with pytest.raises(RuntimeError):
        Fh.gemm_b16(a, b, c.to(torch.bfloat16), M=16, N=16, K=8, lda=12, ldb=12, ldc=16, splitk=2)     # split-K into bf16


@pytest.mark.parametrize("out_bf16", [False, True])
def test_linear_b16_forward_backward(cuda, out_bf16):
    B, n, K, N = 2, 700, 512, 1536
    x = _rand((B, n, K), 5)
    w = torch.randn(N, K, generator=torch.Generator().manual_seed(6)) * 0.05
    bias = torch.randn(N, generator=torch.Generator().manual_seed(7))
    wo = _rand((B, n, N), 8)
    xr = x.double().requires_grad_(); wr = w.to(torch.bfloat16).double().requires_grad_(); br = bias.double().requires_grad_()
    yr = xr @ wr.t() + br
    (yr * wo.double()).sum().backward()
    xd = x.to(cuda).requires_grad_(); wd = w.to(cuda).requires_grad_(); bd = bias.to(cuda).requires_grad_()
    y = Fh.linear_b16(xd, wd, bd, out_bf16=out_bf16)
    assert y.dtype == (torch.bfloat16 if out_bf16 else torch.float32)
    (y.float() * wo.to(cuda).float()).sum().backward()
    _check(y, yr.detach(), out_bf16, "y")
    assert xd.grad.dtype == torch.bfloat16 and wd.grad.dtype == torch.float32
    _check(xd.grad, xr.grad, True, "dx")
    _check(wd.grad, wr.grad, False, "dw")
    _check(bd.grad, br.grad, False, "db")


def test_gemm_b16_batched_row_maps(cuda, tile):
    """Batch items addressed in place (the bags of a padded buffer): NT with offset / strided outputs, TN whose batches add up in one output."""
    b, n0, pad, K, N = 3, 300, 56, 64, 136
    n = n0 + pad
    x = _rand((b, n0, K), 11); w = _rand((N, K), 12)
    y = torch.full((b, n, N), float("nan"), device=cuda, dtype=torch.bfloat16)
    Fh.gemm_b16(x.to(cuda), w.to(cuda), y, M=n0, N=N, K=K, lda=K, ldb=K, ldc=N, nb=b, sa=n0 * K, sc=n * N, c_off=pad * N)
    ref = x.double() @ w.double().t()
    _check(y[:, pad:], ref, True, "batched nt")
    assert torch.isnan(y[:, :pad].float()).all()                      # rows in front of a bag are not touched
    g = _rand((b, n, N), 13)
    dw = torch.zeros(N, K, device=cuda)
    Fh.gemm_b16(g.to(cuda), x.to(cuda), dw, M=N, N=K, K=n0, lda=N, ldb=K, ldc=K, trans=True, splitk=2, nb=b, sa=n * N, sb=n0 * K, sc=0,
                a_off=pad * N)
    refw = torch.einsum("bnm,bnk->mk", g[:, pad:].double(), x.double())
    _check(dw, refw, False, "batched tn")
    with pytest.raises(RuntimeError):                                  # batches that add up need an fp32 output
        Fh.gemm_b16(g.to(cuda), x.to(cuda), dw.to(torch.bfloat16), M=N, N=K, K=n0, lda=N, ldb=K, ldc=K, trans=True, nb=b, sa=n * N, sb=n0 * K, sc=0)

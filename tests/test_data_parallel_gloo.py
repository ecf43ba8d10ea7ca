"""CPU, world_size 2 over gloo: the whole-bag data-parallel step (gradient averaging that completes inside
backward(), grad-less parameters, the gather layer of BatchLoss).  The collectives are device agnostic; on
the GPU box the same code runs over RCCL / xGMI."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

from helpers import smml
from oracle.losses import batch_loss


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


class _Toy(nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.a = nn.Linear(6, 5)
        self.unused = nn.Linear(3, 3)          # never receives a gradient (as attn1d.* with attn_dim = 2)
        self.b = nn.Linear(5, 2)
        self.shared = nn.LayerNorm(5)          # used twice per step (as layer3.norm)

    def forward(self, x):
        return self.b(self.shared(torch.tanh(self.shared(self.a(x)))))


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)
    net = _Toy()
    if rank == 1:                              # desynchronise on purpose: the wrapper must broadcast rank 0's weights
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    dp = smml.BagDataParallel(net, bucket_bytes=32)
    xs = [torch.randn(4, 6, generator=torch.Generator().manual_seed(7 + r)) for r in range(world)]
    out = {}
    for step in range(2):                      # twice: bucket state must re-arm
        net.zero_grad(set_to_none=True)
        dp(xs[rank]).pow(2).sum().backward()
        # gradients are final as soon as backward() returns (train_test.py:158 reads .grad right away)
        out[f"ga{step}"] = net.a.weight.grad.clone()
        out[f"in_bwd{step}"] = dp.stats["launched_in_backward"]
    out["unused_none"] = net.unused.weight.grad is None
    out["buckets"] = dp.stats["buckets"]; out["skipped"] = dp.stats["skipped"]
    # a third step that keeps the gradient tensors (zero_grad(set_to_none=False)): they are the bucket's views by now, autograd
    # accumulates into them in place and the wrapper must neither copy nor clear them
    net.zero_grad(set_to_none=False)
    dp(xs[rank]).pow(2).sum().backward()
    out["ga2"] = net.a.weight.grad.clone()
    # single-process reference: mean over ranks of the per-rank gradients with rank 0's weights
    ref = _Toy()
    g = []
    for r in range(world):
        ref.zero_grad()
        ref(xs[r]).pow(2).sum().backward()
        g.append(ref.a.weight.grad.clone())
    out["ref"] = sum(g) / world
    out["has_module"] = dp.module is net

    # BatchLoss across ranks: gather layer forward / backward
    B = 2
    omic = torch.randn(B, 5, 4, generator=torch.Generator().manual_seed(20 + rank)).requires_grad_()
    vgrid = torch.randn(B * 8, 2, 2, 2, generator=torch.Generator().manual_seed(30 + rank)).requires_grad_()
    o_all = torch.cat(smml.GatherLayer.apply(omic), dim=0)
    v_all = torch.cat(smml.GatherLayer.apply(vgrid), dim=0)
    loss = batch_loss(o_all, v_all, B * world).sum()
    loss.backward()
    out["bl"] = loss.detach()
    out["domic"] = omic.grad.clone()
    # reference: full tensors on one process
    os_ = [torch.randn(B, 5, 4, generator=torch.Generator().manual_seed(20 + r)) for r in range(world)]
    vs_ = [torch.randn(B * 8, 2, 2, 2, generator=torch.Generator().manual_seed(30 + r)) for r in range(world)]
    of = torch.cat(os_, 0).requires_grad_(); vf = torch.cat(vs_, 0).requires_grad_()
    lref = batch_loss(of, vf, B * world).sum(); lref.backward()
    out["bl_ref"] = lref.detach()
    out["domic_ref"] = of.grad[rank * B:(rank + 1) * B].clone()
    q.put((rank, {k: (v.numpy() if torch.is_tensor(v) else v) for k, v in out.items()}))   # by value
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    res = {r: {k: (torch.from_numpy(v) if hasattr(v, 'dtype') else v) for k, v in o.items()} for r, o in res.items()}
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in range(world):
        o = res[r]
        assert o["has_module"] and o["unused_none"]
        # the reductions must start DURING backward.  Step 0 learns the grad-less set (the bucket holding
        # `unused` can only be flushed at the end); from step 1 on every bucket that carries gradients is launched from a hook
        # before backward() ends and the bucket made only of grad-less parameters is skipped
        assert o["buckets"] >= 4 and o["skipped"] >= 1
        assert o["in_bwd1"] == o["buckets"] - o["skipped"], (o["in_bwd0"], o["in_bwd1"], o["buckets"], o["skipped"])
        for step in range(3):
            assert torch.allclose(o[f"ga{step}"], o["ref"], rtol=1e-5, atol=1e-6), f"rank {r} step {step}"
        assert torch.allclose(o["bl"], o["bl_ref"], rtol=1e-5)
        assert torch.allclose(o["domic"], o["domic_ref"], rtol=1e-5, atol=1e-7)
    assert torch.equal(res[0]["ga0"], res[1]["ga0"])


class _Branchy(nn.Module):
    """`extra` receives a gradient only where `use_extra` is set - ranks that disagree in the first step"""
    def __init__(self):
        super().__init__()
        torch.manual_seed(1)
        self.a = nn.Linear(6, 5)
        self.extra = nn.Linear(5, 5)
        self.b = nn.Linear(5, 2)
        self.use_extra = False

    def forward(self, x):
        h = torch.tanh(self.a(x))
        if self.use_extra:
            h = h + self.extra(h)
        return self.b(h)


def _worker_mismatch(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    net = _Branchy()
    net.use_extra = (rank == 1)                # a data-dependent branch taken on one rank only
    dp = smml.BagDataParallel(net, bucket_bytes=32)
    x = torch.randn(4, 6, generator=torch.Generator().manual_seed(3 + rank))
    grads = []
    for step in range(3):
        net.zero_grad(set_to_none=True)
        dp(x).pow(2).sum().backward()
        grads.append(net.extra.weight.grad.clone() if net.extra.weight.grad is not None else None)
    q.put((rank, {"mismatch": dp.stats["used_mismatch"], "skipped": dp.stats["skipped"], "hook_ms": dp.stats["hook_host_ms"],
                  "extra": [g.numpy() if g is not None else None for g in grads],
                  "a": net.a.weight.grad.numpy()}))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_that_disagree_on_the_gradless_set_do_not_hang():
    """the grad-less set is learned per rank from the first backward; if the ranks disagree (a data-dependent branch,
    a warm-up on one rank) they would issue different numbers of all-reduces and hang.  The sets are exchanged once: the union is
    used everywhere (the rank without a gradient contributes zeros), the mismatch is counted in stats."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_mismatch, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0]["mismatch"] == res[1]["mismatch"] == 2          # extra.weight, extra.bias
    assert (res[0]["a"] == res[1]["a"]).all()
    # rank 1's gradient of `extra` is halved (the mean with rank 0's zeros) and identical in every step
    e1 = res[1]["extra"]
    assert e1[0] is not None and (e1[0] == e1[1]).all() and (e1[1] == e1[2]).all()
    assert res[0]["hook_ms"] >= 0.0


def test_default_buckets_split_the_mil_model():
    """DeformCrossTransMIL has 1.96 MB of parameters: the default bucket size must give several buckets (one 2 MiB bucket
    could only ever be reduced after backward had finished)."""
    import argparse
    mil = smml.DeformCrossTransMIL(argparse.Namespace(path_dim=128, attn_dim=2, return_vgrid=True, input_path_dim=512))
    dp = smml.BagDataParallel(mil)
    assert dp.stats["buckets"] >= 3
    sizes = [b.numel * 4 for b in dp._buckets]
    assert max(sizes) < (1 << 20) + (512 << 10)


class _FakeWork:
    waited = 0

    def wait(self):
        _FakeWork.waited += 1
        return True


def test_rccl_branch_of_the_reducer_with_a_fake_backend(monkeypatch):
    """the `nccl` (RCCL) branch of BagDataParallel - ReduceOp.AVG, no pre-scaling - has only ever been skipped on a
    one-GPU box.  Walked here in ONE process against stand-ins for torch.distributed (world 2, backend 'nccl', all_reduce that records
    its arguments and averages with a pretend second rank holding 3 x this rank's gradient): the first RCCL run is then not also the
    first execution of that branch."""
    calls = []

    def fake_all_reduce(t, op=None, group=None, async_op=False):
        calls.append({"op": op, "async": async_op, "numel": t.numel(), "dtype": t.dtype, "pre": t.clone()})
        if op == dist.ReduceOp.AVG:
            t.mul_(2.0)                       # (g + 3 g) / 2
        elif op == dist.ReduceOp.MAX:
            pass                              # the used / unused bitmask exchange: both ranks agree
        else:
            raise AssertionError(f"unexpected reduce op {op} on the RCCL branch")
        return _FakeWork() if async_op else None

    monkeypatch.setattr(dist, "is_initialized", lambda: True)
    monkeypatch.setattr(dist, "get_world_size", lambda group=None: 2)
    monkeypatch.setattr(dist, "get_backend", lambda group=None: "nccl")
    monkeypatch.setattr(dist, "broadcast", lambda t, src=0, group=None: None)
    monkeypatch.setattr(dist, "all_reduce", fake_all_reduce)
    _FakeWork.waited = 0
    net = _Toy()
    dp = smml.BagDataParallel(net, bucket_bytes=32)
    x = torch.randn(4, 6, generator=torch.Generator().manual_seed(11))
    ref = _Toy()
    ref(x).pow(2).sum().backward()
    for step in range(2):
        calls.clear()
        net.zero_grad(set_to_none=True)
        dp(x).pow(2).sum().backward()
        red = [c for c in calls if c["op"] == dist.ReduceOp.AVG]
        assert len(red) == dp.stats["buckets"] - (dp.stats["skipped"] if step else 0)
        assert all(c["async"] and c["dtype"] == torch.float32 for c in red)
        # AVG carries the 1 / world factor: the payload is the raw gradient, not a pre-scaled one
        assert torch.allclose(net.a.weight.grad, 2.0 * ref.a.weight.grad, rtol=1e-6)
        assert torch.allclose(net.b.bias.grad, 2.0 * ref.b.bias.grad, rtol=1e-6)
        assert net.unused.weight.grad is None
    assert dp.stats["skipped"] >= 1 and dp.stats["launched_in_backward"] == dp.stats["buckets"] - dp.stats["skipped"]
    assert _FakeWork.waited >= 2 * (dp.stats["buckets"] - dp.stats["skipped"])
    raw = [c for c in calls if c["op"] == dist.ReduceOp.AVG and c["numel"] >= net.a.weight.numel()]
    assert raw, "a bucket holding a.weight was reduced"

"""GPU, world_size 2 on ONE MI355X (gloo transport, both ranks on cuda:0 - RCCL refuses two ranks per device): the
data-parallel training step of the real model on the HIP kernels - gradient averaging that completes inside backward(),
the BatchLoss exchange with the tile hint - against a single-process run over the concatenated batch."""
import argparse
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _build(smml, S, in_dim):
    args = argparse.Namespace(path_dim=128, attn_dim=2, return_vgrid=True, input_path_dim=in_dim, grid_hw=(S, S))
    mil = smml.DeformCrossTransMIL(args)
    shapes = {k: tuple(v.shape) for k, v in mil.state_dict().items()}
    mil.load_state_dict(smml.synth.fill_params(shapes, 5, "dp"))
    return mil.cuda().eval()


def _loss(smml, model, bl, path, omic, label):
    enc, logits, _, omic_t, vgrid = model(path, omic)
    return torch.nn.functional.cross_entropy(logits, label) + torch.sum(bl(omic_t, vgrid))


def _data(smml, rank, B, S, in_dim):
    path = smml.synth.bag(B, S * S, in_dim, 70 + rank, "dp:bag").cuda()
    omic = torch.relu(smml.synth.normal((B, 128), 70 + rank, "dp:omic")).cuda()
    label = torch.tensor([(rank + i) % 4 for i in range(B)]).cuda()
    return path, omic, label


def _worker(rank, world, port, q, backend="gloo"):
    import importlib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = rank if backend == "nccl" else 0        # RCCL wants one device per rank; the gloo rehearsal shares cuda:0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    smml = importlib.import_module("subspace-multimodal-learning_amd")
    B, S, in_dim = 2, 16, 64
    mil = _build(smml, S, in_dim)
    dp = smml.BagDataParallel(mil, bucket_bytes=1 << 18)
    bl = smml.BatchLoss(B, world)
    path, omic, label = _data(smml, rank, B, S, in_dim)
    for step in range(2):                          # step 0 learns the grad-less set, step 1 overlaps (same data: same gradients)
        mil.zero_grad(set_to_none=True)
        loss = _loss(smml, dp, bl, path, omic, label)
        loss.backward()
    grads = {k: p.grad.detach().cpu().numpy() for k, p in mil.named_parameters() if p.grad is not None}
    q.put((rank, float(loss.item()), grads, dict(dp.stats)))
    dist.barrier()
    dist.destroy_process_group()


def _run_two_ranks(smml, backend):
    import helpers
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, backend)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, l, g, st = q.get(timeout=300)
        res[r] = (l, g)
        # every bucket that carries gradients was launched from a hook, i.e. while backward was still running
        assert st["buckets"] >= 3 and st["launched_in_backward"] == st["buckets"] - st["skipped"] >= 2, st
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process: both ranks' bags in one batch.  Every rank's loss = CE(own bags) + BatchLoss(all bags); the gather
    # layer's backward keeps only the own-rank slice (utils/gather.py:16-20), so after the wrapper's averaging the
    # parameter gradient is (sum_r dCE_r + dBatchLoss) / world - exactly what DDP + GatherLayer give in the reference
    B, S, in_dim = 2, 16, 64
    mil = _build(smml, S, in_dim)
    data = [_data(smml, r, B, S, in_dim) for r in range(world)]
    path = torch.cat([d[0] for d in data]); omic = torch.cat([d[1] for d in data]); label = torch.cat([d[2] for d in data])
    enc, logits, _, omic_t, vgrid = mil(path, omic)
    blv = torch.sum(smml.BatchLoss(B * world, 1)(omic_t, vgrid))
    ce = [torch.nn.functional.cross_entropy(logits[r * B:(r + 1) * B], label[r * B:(r + 1) * B]) for r in range(world)]
    total = (sum(ce) + blv) / world
    total.backward()
    for r in range(world):
        assert abs(res[r][0] - float((ce[r] + blv).item())) <= 1e-4 * abs(res[r][0]), "per-rank loss"
    ref = {k: p.grad.detach().cpu() for k, p in mil.named_parameters() if p.grad is not None}
    assert set(ref) == set(res[0][1]) == set(res[1][1])
    worst = []
    for k, v in ref.items():
        g0, g1 = torch.from_numpy(res[0][1][k]), torch.from_numpy(res[1][1][k])
        assert torch.equal(g0, g1), f"ranks disagree on {k}"
        if k.endswith("rel_pos_bias.mlp.2.bias"):           # exactly 0 in exact arithmetic: both evaluations are rounding noise,
            scale = float(ref["layer3.attn2d.rel_pos_bias.mlp.2.weight"].abs().max())      # bounded against the layer's weight gradient
            assert float(g0.abs().max()) <= 1e-4 * scale and float(v.abs().max()) <= 1e-4 * scale, k
            continue
        err = float((g0 - v).abs().max() / v.abs().max().clamp_min(1e-30))
        helpers.record(f"dp2[{backend}] d{k}", err, None, DP_TOL, "max vs single process")
        if err > DP_TOL:
            worst.append(f"{k}: {err:.2e}")
    assert not worst, "two-rank gradients differ from the single-process run:\n  " + "\n  ".join(worst)


# Both sides are HIP fp32 evaluations of the same sums in different orders (2 + 2 bags vs 4 bags per launch): they agree to fp32
# rounding amplified by the conditioning of each gradient; every tensor's distance is recorded in the parity report
DP_TOL = 1e-4


def test_two_ranks_match_single_process(cuda, smml):
    """2 ranks on ONE device over gloo (RCCL refuses two ranks per device)."""
    _run_two_ranks(smml, "gloo")


def test_two_ranks_over_rccl(cuda, smml):
    """The same comparison over RCCL, one device per rank - runs wherever two GPUs are visible (the driver's 8-GPU node)."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    _run_two_ranks(smml, "nccl")


def _worker_rccl_one_rank(port, q):
    import importlib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    smml = importlib.import_module("subspace-multimodal-learning_amd")
    B, S, in_dim = 2, 16, 64
    out = {}
    for wrapped in (False, True):
        mil = _build(smml, S, in_dim)
        model = smml.BagDataParallel(mil, bucket_bytes=1 << 18, collectives_at_world_1=True) if wrapped else mil
        bl = smml.BatchLoss(B, 1)
        path, omic, label = _data(smml, 0, B, S, in_dim)
        for step in range(2):
            mil.zero_grad(set_to_none=True)
            loss = _loss(smml, model, bl, path, omic, label)
            loss.backward()
        out[wrapped] = ({k: p.grad.detach().cpu() for k, p in mil.named_parameters() if p.grad is not None}, float(loss),
                        dict(model.stats) if wrapped else None)
    # the gather layer of BatchLoss over RCCL (all_gather + its own-rank backward)
    t = torch.randn(3, 5, device="cuda", requires_grad=True)
    g, = smml.losses.GatherLayer.apply(t)
    (g * 2.0).sum().backward()
    gather_ok = bool(torch.equal(g.detach(), t.detach())) and bool(torch.equal(t.grad, torch.full_like(t, 2.0)))
    # two runs of the same step differ in the last bits of the weight gradients that are reduced with float atomics: compared
    # to 1e-5 of each tensor's scale, not bit for bit
    worst = max(float((out[True][0][k] - v).abs().max() / v.abs().max().clamp_min(1e-30)) for k, v in out[False][0].items()
                if not k.endswith("rel_pos_bias.mlp.2.bias"))
    same = worst <= 1e-5 and set(out[True][0]) == set(out[False][0])
    q.put((same, out[True][1], out[False][1], out[True][2], gather_ok, dist.get_backend()))
    dist.barrier()
    dist.destroy_process_group()


def test_reducer_over_rccl_one_rank(cuda, smml):
    """RCCL under the data-parallel reducer on a one-GPU box: a ONE-rank nccl group (`init_process_group(device_id=...)`), parameters and buffers
    broadcast, every bucket all-reduced with ReduceOp.AVG from the gradient hooks while backward runs, the grad-less-set exchange, the
    end-of-backward wait, and BatchLoss's GatherLayer - the average over one rank is the identity, so the gradients must equal the unwrapped
    model's (to the run-to-run noise of the atomically reduced weight gradients).  (Two ranks over RCCL need two devices: test_two_ranks_over_rccl.)"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_worker_rccl_one_rank, args=(_free_port(), q))
    p.start()
    same, l_wrapped, l_plain, st, gather_ok, backend = q.get(timeout=300)
    p.join(timeout=60)
    assert p.exitcode == 0 and backend == "nccl"
    assert same and l_wrapped == l_plain, "the one-rank RCCL reducer changed the gradients"
    assert st["buckets"] >= 3 and st["launched_in_backward"] == st["buckets"] - st["skipped"] >= 2, st
    assert gather_ok


# ------------------------------------------------------------------------------------------------
# SyncBatchNorm + BagDataParallel (reference: main.py:118-119 converts BN to SyncBN, then wraps) - only reached with
# fusion_type 'pofusion' (BilinearFusion's two BatchNorm1d, models/fusion.py:44-45)
# ------------------------------------------------------------------------------------------------
def _pofusion_net(smml, S, in_dim):
    from test_oracle_golden import pathomic_args
    args = pathomic_args(fusion_type="pofusion", skip=1, input_path_dim=in_dim, grid_hw=(S, S), batch_size=2, mmhid=128)
    net = smml.DeformPathomicNet(args)
    # parameters from the portable generator; BatchNorm's buffers (running statistics, the 0-dim step counter) keep their defaults
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items() if v.dtype.is_floating_point and v.dim() >= 1 and "running_" not in k}
    missing = net.load_state_dict(smml.synth.fill_params(shapes, 9, "dp:po"), strict=False)
    assert all("running_" in k or "num_batches_tracked" in k for k in missing.missing_keys) and not missing.unexpected_keys
    return net


def _po_mode(net):
    """dropout off everywhere, BatchNorm on batch statistics: the only train-mode behaviour whose two-rank result has a single-process twin"""
    net.eval()
    for m in net.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.train()
    return net


def _po_data(smml, rank, B, S, in_dim):
    path = smml.synth.bag(B, S * S, in_dim, 80 + rank, "dp:po:bag").cuda()
    xt = smml.synth.normal((B, 59), 80 + rank, "dp:po:t").cuda()
    xi = smml.synth.normal((B, 361), 80 + rank, "dp:po:i").cuda()
    label = torch.tensor([(rank + 2 * i) % 4 for i in range(B)]).cuda()
    return path, xt, xi, label


def _worker_syncbn(rank, world, port, q):
    import importlib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    smml = importlib.import_module("subspace-multimodal-learning_amd")
    B, S, in_dim = 2, 12, 64
    net = _pofusion_net(smml, S, in_dim).cuda()
    if rank == 1:                                  # desynchronise rank 1's BN buffers on purpose: the wrap must bring rank 0's over
        with torch.no_grad():
            for m in net.modules():
                if isinstance(m, torch.nn.BatchNorm1d):
                    m.running_mean.add_(3.0); m.running_var.mul_(5.0)
    net = torch.nn.SyncBatchNorm.convert_sync_batchnorm(net)          # main.py:118
    n_sync = sum(isinstance(m, torch.nn.SyncBatchNorm) for m in net.modules())
    dp = smml.BagDataParallel(_po_mode(net), bucket_bytes=1 << 18)    # main.py:119
    path, xt, xi, label = _po_data(smml, rank, B, S, in_dim)
    for step in range(2):
        net.zero_grad(set_to_none=True)
        out = dp(x_path=path, x_omic=None, x_omic_tumor=xt, x_omic_immune=xi)
        loss = torch.nn.functional.cross_entropy(out[3][2], label)
        loss.backward()
    grads = {k: p.grad.detach().cpu().numpy() for k, p in net.named_parameters() if p.grad is not None}
    bufs = {k: b.detach().cpu().numpy() for k, b in net.named_buffers() if "running" in k}
    q.put((rank, float(loss.item()), grads, bufs, n_sync, dict(dp.stats)))
    dist.barrier()
    dist.destroy_process_group()


def test_syncbn_pofusion_two_ranks(cuda, smml):
    """fusion_type 'pofusion' -> SyncBatchNorm.convert_sync_batchnorm -> BagDataParallel on two ranks (gloo, both on
    cuda:0): SyncBN's own collectives inside forward / backward interleave with the bucket all-reduces; buffers are broadcast at wrap
    time.  Gradients and running statistics equal a single-process BatchNorm1d run over the concatenated batch."""
    import helpers
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_syncbn, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, l, g, bufs, n_sync, st = q.get(timeout=300)
        res[r] = (l, g, bufs)
        assert n_sync == 2, "BilinearFusion's two BatchNorm1d must have been converted"
        assert st["launched_in_backward"] == st["buckets"] - st["skipped"] >= 2, st
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    B, S, in_dim = 2, 12, 64
    net = _po_mode(_pofusion_net(smml, S, in_dim).cuda())
    data = [_po_data(smml, r, B, S, in_dim) for r in range(world)]
    path, xt, xi, label = (torch.cat([d[i] for d in data]) for i in range(4))
    for step in range(2):                          # two steps: the running statistics take two momentum updates on both sides
        net.zero_grad(set_to_none=True)
        out = net(x_path=path, x_omic=None, x_omic_tumor=xt, x_omic_immune=xi)
        ce = [torch.nn.functional.cross_entropy(out[3][2][r * B:(r + 1) * B], label[r * B:(r + 1) * B]) for r in range(world)]
        (sum(ce) / world).backward()
    for r in range(world):
        assert abs(res[r][0] - float(ce[r].item())) <= 1e-4 * abs(res[r][0]), "per-rank loss"
    ref = {k: p.grad.detach().cpu() for k, p in net.named_parameters() if p.grad is not None}
    assert set(ref) == set(res[0][1]) == set(res[1][1])
    assert any("fusion.encoder1.1" in k for k in ref), "the SyncBN affine parameters carry gradients"
    worst = []
    for k, v in ref.items():
        g0, g1 = torch.from_numpy(res[0][1][k]), torch.from_numpy(res[1][1][k])
        assert torch.equal(g0, g1), f"ranks disagree on {k}"
        if k.endswith("rel_pos_bias.mlp.2.bias"):
            continue
        if k in ("fusion.encoder1.0.bias", "fusion.encoder2.0.bias"):
            # the bias of a Linear that feeds a BatchNorm: its gradient is exactly zero in exact arithmetic (the batch mean removes it) -
            # both sides are rounding noise, held against the scale of the layer's weight gradient
            scale = float(ref[k.replace(".bias", ".weight")].abs().max())
            assert float(g0.abs().max()) <= 1e-4 * scale and float(v.abs().max()) <= 1e-4 * scale, k
            continue
        err = float((g0 - v).abs().max() / v.abs().max().clamp_min(1e-30))
        helpers.record(f"syncbn dp2 d{k}", err, None, 2e-4, "max vs single process")
        if err > 2e-4:
            worst.append(f"{k}: {err:.2e}")
    assert not worst, "two-rank SyncBN gradients differ from the single-process BatchNorm run:\n  " + "\n  ".join(worst)
    refb = {k: b.detach().cpu() for k, b in net.named_buffers() if "running" in k}
    for k, v in refb.items():
        b0, b1 = torch.from_numpy(res[0][2][k]), torch.from_numpy(res[1][2][k])
        assert torch.equal(b0, b1), f"running statistics differ between ranks: {k}"
        assert torch.allclose(b0, v, rtol=1e-4, atol=1e-6), f"{k}: running statistics differ from the single-process run"

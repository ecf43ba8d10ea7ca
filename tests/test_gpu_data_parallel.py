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


def _worker(rank, world, port, q):
    import importlib
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    smml = importlib.import_module("subspace-multimodal-learning_amd")
    B, S, in_dim = 2, 16, 64
    mil = _build(smml, S, in_dim)
    dp = smml.BagDataParallel(mil, bucket_bytes=1 << 18)
    bl = smml.BatchLoss(B, world)
    path, omic, label = _data(smml, rank, B, S, in_dim)
    loss = _loss(smml, dp, bl, path, omic, label)
    loss.backward()
    grads = {k: p.grad.detach().cpu().numpy() for k, p in mil.named_parameters() if p.grad is not None}
    q.put((rank, float(loss.item()), grads))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_match_single_process(cuda, smml):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        r, l, g = q.get(timeout=300)
        res[r] = (l, g)
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
    for k, v in ref.items():
        if k.endswith("rel_pos_bias.mlp.2.bias"):
            continue
        g0, g1 = torch.from_numpy(res[0][1][k]), torch.from_numpy(res[1][1][k])
        assert torch.equal(g0, g1), f"ranks disagree on {k}"
        err = float((g0 - v).abs().max() / v.abs().max().clamp_min(1e-30))
        assert err <= 2e-3, f"{k}: {err:.2e}"      # the BatchLoss gradient of identical-value tiles is ill-conditioned at B = 4

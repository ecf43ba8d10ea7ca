"""Host-side mirror of the reference's batch-similarity / orthogonality losses and its gather layer.

  BatchLoss      utils/loss.py:7-40        same ctor (batch_size, world_size) and forward(omic, vgrid)
  GatherLayer    utils/gather.py:5-20      all_gather with the reference's backward (own-rank slice)
  OrthogonalLoss models/cmta_utils.py:1212-1228

The long contraction (Gram matrix of the [N_b, N*C] tiled omic matrix, K = 320 000 ... 1 280 000) runs
through the split-K matrix-core kernel; everything after it is [N_b, N_b]-sized."""
from __future__ import annotations

import torch
import torch.distributed as dist
from torch import nn

from . import functional as Fh


class GatherLayer(torch.autograd.Function):
    """Gather tensors from all ranks (RCCL all_gather over xGMI on GPU tensors, gloo on CPU tensors);
    backward returns this rank's slice of the incoming gradients, as utils/gather.py:16-20 does
    (every rank computes the same full loss and the data-parallel wrapper averages the parameter grads)."""

    @staticmethod
    def forward(ctx, input):
        world = dist.get_world_size()
        out = torch.empty((world,) + tuple(input.shape), dtype=input.dtype, device=input.device)
        dist.all_gather(list(out.unbind(0)), input.contiguous())     # contiguous slices of one buffer
        return tuple(out.unbind(0))

    @staticmethod
    def backward(ctx, *grads):
        return grads[dist.get_rank()].contiguous()


_FUSED_TAIL = __import__("os").environ.get("SMML_BATCHLOSS_TAIL", "1") != "0"      # measurement switch: 0 = the torch-level tail of round 3


def _row_normalised_gram(x3: torch.Tensor) -> torch.Tensor:
    g = Fh.gram(x3)                                        # [nb, R, R] on the matrix cores
    return g / g.norm(dim=2, keepdim=True)


class BatchLoss(nn.Module):
    """Same constructor / forward as utils/loss.py:7-40.  `use_tile_hint` (additive, default on): when `omic` is the
    token tile produced by DeformCrossTransMIL (it then carries the un-tiled [B, C] vector), the exchange and the Gram
    run on the vector - 20 MB/rank of all-gather at B = 4, N = 10 000 shrink to 2 KB, values equal up to rounding
    (SURVEY.md C4).  Any other tensor takes the reference's full path (gather the tile, Gram over N*C columns)."""

    def __init__(self, batch_size, world_size, use_tile_hint=True):
        super().__init__()
        self.batch_size = batch_size
        self.world_size = world_size
        self.use_tile_hint = use_tile_hint

    def forward(self, omic, vgrid):
        N = self.batch_size * self.world_size
        compact = getattr(omic, "_smml_compact", None) if self.use_tile_hint else None
        if compact is not None and compact.dim() == 2 and compact.shape[0] == omic.shape[0]:
            omic = compact
        if self.world_size > 1:
            omic = torch.cat(GatherLayer.apply(omic), dim=0)
            vgrid = torch.cat(GatherLayer.apply(vgrid), dim=0)
        omic = omic.reshape(1, N, -1)
        vgrid = vgrid.reshape(8, N, -1)                    # reinterprets the (b g)-major buffer, utils/loss.py:23
        if omic.is_cuda and N <= 64 and _FUSED_TAIL:
            # the two Gram products on the matrix cores, everything after them (row norms, mean over the 8 groups, difference, square) in ONE launch
            return Fh.batchloss_tail(Fh.gram(omic)[0], Fh.gram(vgrid), N)
        similarity = _row_normalised_gram(omic)[0]
        mean_vgrid_sim = _row_normalised_gram(vgrid).mean(dim=0)
        return (similarity - mean_vgrid_sim) ** 2 / N


class OrthogonalLoss(nn.Module):
    def __init__(self, gamma=0.5):
        super().__init__()
        self.gamma = gamma

    def forward(self, P, P_hat, G, G_hat):
        """[B, d] x 4 -> [B]; the five cosine terms and their gradients are wavefront reductions in one kernel
        (`.detach()` placements of the reference are built into its backward)."""
        return Fh.orthogonal_loss(P, P_hat, G, G_hat, self.gamma)

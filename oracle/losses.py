"""Oracle (TEST INFRASTRUCTURE ONLY): the batch-similarity and orthogonality losses.

Plain PyTorch fp32 restatement of
  utils/loss.py:7-40            BatchLoss      (+ utils/gather.py:5-20 when world_size > 1)
  models/cmta_utils.py:1212-1228 OrthogonalLoss
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def _row_normalised_gram(x: torch.Tensor) -> torch.Tensor:
    g = x @ x.t()
    return g / g.norm(dim=1, keepdim=True)       # utils/loss.py:26-28


def batch_loss(omic: torch.Tensor, vgrid: torch.Tensor, n_total: int) -> torch.Tensor:
    """omic [N_b, ...], vgrid [(N_b * 8), ...] (already gathered over ranks) -> [N_b, N_b].

    ``vgrid.view(8, N_b, -1)`` (utils/loss.py:23) reinterprets the (b g)-major buffer: slice k,
    row r is flat row k*N_b + r - not a per-group split; kept bit-for-bit."""
    N = n_total
    s = _row_normalised_gram(omic.reshape(N, -1))
    vg = vgrid.reshape(8, N, -1)
    v = torch.stack([_row_normalised_gram(vg[i]) for i in range(8)]).mean(dim=0)
    return (s - v) ** 2 / N


def orthogonal_loss(P, P_hat, G, G_hat, gamma: float = 0.5) -> torch.Tensor:
    """cmta_utils.py:1217-1228; each input [B, d] -> [B]."""
    cs = lambda a, b: F.cosine_similarity(a, b, dim=1).abs()
    pos = (1 - cs(P.detach(), P_hat)) + (1 - cs(G.detach(), G_hat))
    neg = cs(P, G) + cs(P.detach(), G_hat) + cs(G.detach(), P_hat)
    return pos + gamma * neg

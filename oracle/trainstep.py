"""Oracle (TEST INFRASTRUCTURE ONLY): the gradient-modulation block of the reference's training loop.

Plain PyTorch restatement of train_test.py:87-184 for the non-survival task types (diag2021 / grade / subtype):
given the two branch vectors, the fused classifier's weight / bias, the labels and classifier.weight.grad, returns the
modulated gradient and (score_t, score_i, ratio_t, ratio_i, per-row cosine similarity, per-row branch)."""
from __future__ import annotations

import torch


def gradient_modulate(feat_t, feat_i, weight, bias, label, grad):
    hs = weight.shape[1] // 2
    out_t = feat_t @ weight[:, :hs].t() + bias / 2                                   # :90-93
    out_i = feat_i @ weight[:, hs:].t() + bias / 2
    score_t = sum(torch.softmax(out_t, dim=1)[i][label[i]] for i in range(out_t.size(0)))   # :119-120
    score_i = sum(torch.softmax(out_i, dim=1)[i][label[i]] for i in range(out_i.size(0)))
    ratio_t = score_t / score_i                                                      # :150-152
    ratio_i = 1 / ratio_t
    g = grad.clone()
    sims, branches = [], []
    for r in range(g.shape[0]):                                                      # :158-183
        gt, gi = g[r, :hs].clone(), g[r, hs:].clone()
        sim = torch.dot(gt, gi) / (gt.norm() * gi.norm())
        branch = 0
        if sim < 0:
            if ratio_t < 1:
                proj = torch.dot(gt, gi) / gi.norm() ** 2 * gi
                a = gt - proj
                perpen = a - proj
                g[r, :hs] = a.norm() * (perpen / perpen.norm())
                branch = 1
            elif ratio_i < 1:
                proj = torch.dot(gi, gt) / gt.norm() ** 2 * gt
                a = gi - proj
                perpen = a - proj
                g[r, hs:] = a.norm() * (perpen / perpen.norm())
                branch = 2
        sims.append(float(sim)); branches.append(branch)
    return g, dict(score_t=float(score_t), score_i=float(score_i), ratio_t=float(ratio_t), ratio_i=float(ratio_i),
                   sim=sims, branch=branches)

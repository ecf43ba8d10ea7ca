"""Oracle (TEST INFRASTRUCTURE ONLY): the gradient-modulation block of the reference's training loop.

`gradient_modulate_survival` restates the survival branch (train_test.py:99-102,121-149).  Its C-index comes from scikit-survival
(`concordance_index_censored`, called at utils/utils.py:315-317), a third-party dependency that is ABSENT from this image and
unpinned in the reference (no requirements file): the published pair rule of that function (sksurv.metrics, `_get_comparable` /
`_estimate_concordance_index`, 0.2x series) is restated in `concordance_index_censored` below - PARITY UNPINNED for this branch.

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


def concordance_index_censored(event, time, estimate, tied_tol=1e-8):
    """scikit-survival's rule: a sample with an event is comparable with every sample of later time and with every sample censored at the
    same time; a comparable pair is concordant if the event sample's estimate is larger, tied if the two are within tied_tol (counted 1/2).
    -> (cindex, concordant, discordant, tied_risk, comparable); raises ZeroDivisionError without a comparable pair (sksurv:
    NoComparablePairException)."""
    n = len(time)
    con = dis = tie = 0
    for i in range(n):
        if not event[i]:
            continue
        for j in range(n):
            if j == i:
                continue
            if time[j] > time[i] or (time[j] == time[i] and not event[j]):
                if abs(estimate[i] - estimate[j]) <= tied_tol:
                    tie += 1
                elif estimate[i] > estimate[j]:
                    con += 1
                else:
                    dis += 1
    comp = con + dis + tie
    return (con + 0.5 * tie) / comp, con, dis, tie, comp


def gradient_modulate_survival(feat_t, feat_i, weight, bias, censor, survtime, grad):
    """train_test.py:90-93 (logits), :99-102 (hazards, S), :123-134 (risk, C-index), :143-183 (ratios, row edits)."""
    hs = weight.shape[1] // 2
    out_t = feat_t @ weight[:, :hs].t() + bias / 2
    out_i = feat_i @ weight[:, hs:].t() + bias / 2
    S_t = torch.cumprod(1 - torch.sigmoid(out_t), dim=1)
    S_i = torch.cumprod(1 - torch.sigmoid(out_i), dim=1)
    risk_t, risk_i = -torch.sum(S_t, dim=1), -torch.sum(S_i, dim=1)
    g = grad.clone()
    info = dict(cindex_t=None, cindex_i=None, branch=[0] * g.shape[0])
    if float(censor.float().mean()) == 1:                                            # :127-133: all censored -> no modulation
        return g, info
    ev = [bool(1 - int(c)) for c in censor.tolist()]
    try:
        ct = concordance_index_censored(ev, survtime.tolist(), risk_t.tolist())[0]
        ci = concordance_index_censored(ev, survtime.tolist(), risk_i.tolist())[0]
    except ZeroDivisionError:
        return g, info
    info["cindex_t"], info["cindex_i"] = ct, ci
    ratio_t = ct / ci if ci != 0 else float("inf")
    ratio_i = 1 / ratio_t if ratio_t != 0 else float("inf")
    for r in range(g.shape[0]):
        gt, gi = g[r, :hs].clone(), g[r, hs:].clone()
        sim = torch.dot(gt, gi) / (gt.norm() * gi.norm())
        if sim < 0:
            if ratio_t < 1:
                proj = torch.dot(gt, gi) / gi.norm() ** 2 * gi
                a = gt - proj
                perpen = a - proj
                g[r, :hs] = a.norm() * (perpen / perpen.norm())
                info["branch"][r] = 1
            elif ratio_i < 1:
                proj = torch.dot(gi, gt) / gt.norm() ** 2 * gt
                a = gi - proj
                perpen = a - proj
                g[r, hs:] = a.norm() * (perpen / perpen.norm())
                info["branch"][r] = 2
    return g, info

"""Host-side mirror of the two-branch assembly around DeformCrossTransMIL:

  MaxNet            models/model.py:142-187   omic SNN encoder (4 x Linear+ELU+AlphaDropout, ReLU, classifier)
  DeformPathomicNet models/model.py:440-544   two omic encoders + two DeformCrossTransMIL branches over the same
                                              bag, concat fusion, three classifiers; 7-tuple output
  define_net        models/model.py:49-79     (mode 'deformpathomic' only)

Same constructor arguments (an `args` namespace with the keys of config/config_mine.yaml), forward kwargs,
return tuple and parameter names.  fusion_type 'concat' (the shipped default, config_mine.yaml:17) and 'pofusion'
(BilinearFusion, fusion.py) are built."""
from __future__ import annotations

import math

import torch
from torch import nn
from torch.nn import Parameter

from . import functional as Fh
from .deform_cross_trans_mil import DeformCrossTransMIL
from .fusion import define_bifusion


def init_max_weights(module):
    """utils/utils.py:214-219: N(0, 1/sqrt(fan_in)) weights, zero biases on every nn.Linear."""
    for m in module.modules():
        if type(m) == nn.Linear:
            stdv = 1. / math.sqrt(m.weight.size(1))
            m.weight.data.normal_(0, stdv)
            m.bias.data.zero_()


class MaxNet(nn.Module):
    def __init__(self, input_dim=59, omic_dim=32, return_grad='False', dropout_rate=0.25, label_dim=1, init_max=True):
        super().__init__()
        hidden = [64, 48, 32, 32]
        self.return_grad = return_grad
        dims = [input_dim, hidden[0], hidden[1], hidden[2], omic_dim]
        self.encoder = nn.Sequential(*[
            nn.Sequential(nn.Linear(dims[i], dims[i + 1]), nn.ELU(), nn.AlphaDropout(p=dropout_rate, inplace=False))
            for i in range(4)])
        self.relu = nn.ReLU(inplace=False)
        self.classifier = nn.Sequential(nn.Linear(omic_dim, label_dim))
        if init_max:
            init_max_weights(self)
        self.output_range = Parameter(torch.FloatTensor([6]), requires_grad=False)
        self.output_shift = Parameter(torch.FloatTensor([-3]), requires_grad=False)

    def forward(self, **kwargs):
        h = kwargs['x_omic'].float()
        for blk in self.encoder:
            h = blk[2](blk[1](Fh.linear(h, blk[0].weight, blk[0].bias)))     # [B, <=64]-sized activations
        features = self.relu(h)
        logits = Fh.linear(features, self.classifier[0].weight, self.classifier[0].bias)
        return features, logits, None


class DeformPathomicNet(nn.Module):
    def __init__(self, args):
        super().__init__()
        init_max = True if args.init_type == "max" else False
        self.args = args
        self.omic_net_tumor = MaxNet(input_dim=args.input_size_omic_tumor, omic_dim=args.omic_dim,
                                     return_grad=args.return_grad, dropout_rate=args.dropout_rate,
                                     label_dim=args.label_dim, init_max=init_max)
        self.omic_net_immune = MaxNet(input_dim=args.input_size_omic_immune, omic_dim=args.omic_dim,
                                      return_grad=args.return_grad, dropout_rate=args.dropout_rate,
                                      label_dim=args.label_dim, init_max=init_max)
        self.pathomic_net_tumor = DeformCrossTransMIL(args)
        self.pathomic_net_immune = DeformCrossTransMIL(args)
        self.bilinear_dim = 20
        if args.fusion_type != "concat":
            self.fusion = define_bifusion(fusion_type=args.fusion_type, skip=args.skip, use_bilinear=args.use_bilinear,
                                          gate1=args.path_gate, gate2=args.omic_gate, dim1=args.path_dim, dim2=args.omic_dim,
                                          scale_dim1=args.path_scale, scale_dim2=args.omic_scale, mmhid=args.mmhid,
                                          dropout_rate=args.dropout_rate)
            self.classifier = nn.Sequential(nn.Linear(args.mmhid, args.label_dim))
        else:
            self.classifier = nn.Linear(args.mmhid * 2, args.label_dim)
        # The reference defines the two per-branch heads only in the 'concat' branch (model.py:463-469) while its
        # forward reads them unconditionally (:522-523), i.e. 'pofusion' cannot run there; they are defined for both here.
        self.classifier_tumor = nn.Sequential(nn.Linear(args.mmhid, args.label_dim))
        self.classifier_immune = nn.Sequential(nn.Linear(args.mmhid, args.label_dim))
        self.return_grad = args.return_grad
        self.cut_fuse_grad = args.cut_fuse_grad
        self.fusion_type = args.fusion_type
        self.output_range = Parameter(torch.FloatTensor([6]), requires_grad=False)
        self.output_shift = Parameter(torch.FloatTensor([-3]), requires_grad=False)

    def forward(self, **kwargs):
        x_path = kwargs['x_path']
        # both branches start with relu(_fc1(x_path)) on the SAME bag (model.py:488-497): one batched launch per direction
        ft, fi = self.pathomic_net_tumor._fc1[0], self.pathomic_net_immune._fc1[0]
        shared = ft.weight.shape == fi.weight.shape and x_path.is_cuda and not x_path.requires_grad
        if shared:
            self.pathomic_net_tumor.prefetch(x_path.shape[1])       # parameter-only work of both branches, beside the shared first layer
            self.pathomic_net_immune.prefetch(x_path.shape[1])
            pf_t, pf_i = Fh.dual_linear_relu(x_path.float(), ft.weight, ft.bias, fi.weight, fi.bias)
        omic_vec_tumor, _, _ = self.omic_net_tumor(x_omic=kwargs['x_omic_tumor'])
        rt = self.pathomic_net_tumor.forward_features(pf_t, omic_vec_tumor) if shared else self.pathomic_net_tumor(path=x_path, omic=omic_vec_tumor)
        omic_vec_immune, _, _ = self.omic_net_immune(x_omic=kwargs['x_omic_immune'])
        ri = self.pathomic_net_immune.forward_features(pf_i, omic_vec_immune) if shared else self.pathomic_net_immune(path=x_path, omic=omic_vec_immune)
        pathomic_vec_tumor, pathomic_grads_tumor = rt[0], rt[2]
        pathomic_vec_immune, pathomic_grads_immune = ri[0], ri[2]
        a, b = pathomic_vec_tumor, pathomic_vec_immune
        if self.cut_fuse_grad:
            a, b = a.clone().detach(), b.clone().detach()
        if self.fusion_type == "concat":
            features = torch.cat((a, b), 1)
            hazard = Fh.linear(features, self.classifier.weight, self.classifier.bias)
        else:
            features = self.fusion(a, b)
            hazard = Fh.linear(features, self.classifier[0].weight, self.classifier[0].bias)
        hazard_tumor = Fh.linear(pathomic_vec_tumor, self.classifier_tumor[0].weight, self.classifier_tumor[0].bias)
        hazard_immune = Fh.linear(pathomic_vec_immune, self.classifier_immune[0].weight, self.classifier_immune[0].bias)
        if self.return_grad == "True":
            raise NotImplementedError("return_grad='True' (get_grad_embedding) is outside the accelerated path")
        fuse_grads = None
        if self.args.task_type == "survival":
            hazard = torch.sigmoid(hazard)
            hazard_tumor = torch.sigmoid(hazard_tumor)
            hazard_immune = torch.sigmoid(hazard_immune)
        if self.args.return_vgrid:
            logits = [hazard_tumor, hazard_immune, hazard, rt[3], rt[4], ri[3], ri[4]]
        else:
            logits = [hazard_tumor, hazard_immune, hazard]
        return features, pathomic_vec_tumor, pathomic_vec_immune, logits, fuse_grads, pathomic_grads_tumor, pathomic_grads_immune


def define_net(args):
    """mode 'deformpathomic' of models/model.py:49-79 (init_net with init_type 'max' / 'none' leaves the
    constructor's initialisation in place, utils/utils.py:222-240)."""
    if args.mode != "deformpathomic":
        raise NotImplementedError(f"model [{args.mode}] is not part of the accelerated path")
    return DeformPathomicNet(args=args)

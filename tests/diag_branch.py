"""Diagnostic: ONE DeformCrossTransMIL branch with cfg4's immune parameters at B = 2, S = 100; loss variants; HIP vs fp64 oracle."""
import importlib, sys, time
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import params_for, smml, synth, rel_err, l2_err
from oracle.losses import batch_loss
from oracle.mil import deform_cross_trans_mil, max_net
from oracle.nystrom import _sub
from test_oracle_golden import pathomic_args
dev = torch.device("cuda:0")
B, S = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 100
branch = sys.argv[2] if len(sys.argv) > 2 else "immune"
args = pathomic_args(input_path_dim=512, batch_size=B)
net = smml.DeformPathomicNet(args)
params = params_for(net, 17, "cfg4")
x_path = synth.bag(B, S * S, 512, 17, "cfg4:bag")
x_o = synth.normal((B, 361), 17, "cfg4:immune") if branch == "immune" else synth.normal((B, 59), 17, "cfg4:tumor")
pm = {k: v for k, v in _sub(params, f"pathomic_net_{branch}.").items()}
omic, _ = max_net(x_o, _sub(params, f"omic_net_{branch}."))
mil = smml.DeformCrossTransMIL(args)
mil.load_state_dict(pm); mil = mil.to(dev).eval()
label = torch.tensor([2, 0])
for variant in ("ce", "ce+bl", "bl"):
    def total(enc, logits, omic_t, vg, bl):
        l = 0
        if "ce" in variant:
            l = l + torch.nn.functional.cross_entropy(logits, label.to(logits.device)) + enc.sum() * 0.01
        if "bl" in variant:
            l = l + 0.5 * bl(omic_t, vg).sum()
        return l
    t0 = time.time()
    p64 = {k: v.clone().double().requires_grad_() for k, v in pm.items()}
    o64 = omic.clone().double().requires_grad_()
    enc, logits, omic_t, vg = deform_cross_trans_mil(x_path.double(), o64, p64, grid_hw=(S, S), q_chunk=1024)
    total(enc, logits, omic_t, vg, lambda o, v: batch_loss(o, v, B)).backward()
    mil.zero_grad(set_to_none=True)
    od = omic.clone().to(dev).requires_grad_()
    e, lg, _, ot, vgd = mil(x_path.to(dev), od)
    total(e, lg, ot, vgd, smml.BatchLoss(B, 1)).backward()
    print(f"--- {branch} S={S} loss = {variant}  ({time.time() - t0:.0f} s)")
    print(f"   domic  l2 {l2_err(od.grad, o64.grad):.2e}")
    for k in ("_fc1.0.weight", "fusion_layer.fusion_layer.weight", "layer3.norm.weight", "layer3.attn2d.to_q.weight", "layer3.attn2d.to_offsets.2.weight",
              "layer3.attn2d.to_k.weight", "layer3.attn2d.rel_pos_bias.mlp.1.0.weight", "pooler.dense.weight"):
        g = dict(mil.named_parameters())[k].grad
        if g is not None and p64[k].grad is not None:
            print(f"   d{k:<45s} l2 {l2_err(g, p64[k].grad):.2e} max {rel_err(g, p64[k].grad):.2e}")

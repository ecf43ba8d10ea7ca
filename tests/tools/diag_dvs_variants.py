"""Diagnostic, step B (GPU box, seconds per variant): the cfg4 tumor branch's d vs of the loaded library (SMML_LIB) against the cached fp64
reference of tests/diag_dvs_ref.py: total, sampler part, position-bias part, and the error of d to_offsets.2.weight they imply."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from helpers import l2_err, params_for, rel_err, smml, synth
from test_oracle_golden import pathomic_args
S = 100
cuda = torch.device("cuda:0")
Fh = smml.functional
ref = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "cfg4_tumor_ref.pt"))
args = pathomic_args(input_path_dim=512, batch_size=1)
net = smml.DeformPathomicNet(args)
params = params_for(net, 17, "cfg4")
net.load_state_dict(params); net = net.to(cuda).eval()
x_path = synth.bag(1, S * S, 512, 17, "cfg4:bag"); x_o = synth.normal((1, 59), 17, "cfg4:tumor")
w_enc = synth.normal((1, 128), 17, "diag:wenc")
cap = {}
ob = Fh._Sample.backward
def sb(ctx, dkv):
    r = ob(ctx, dkv); cap["sampler"] = r[1].detach().clone(); return r
Fh._Sample.backward = staticmethod(sb)
oa = Fh._DeformAttn.backward
def ab(ctx, dout):
    r = oa(ctx, dout); cap["cpb"] = r[3].detach().clone(); return r
Fh._DeformAttn.backward = staticmethod(ab)
enc, logits, _, omic_t, vg = net.pathomic_net_tumor(x_path.to(cuda), net.omic_net_tumor(x_omic=x_o.to(cuda))[0])
(enc * w_enc.to(cuda)).sum().backward()
tot = (cap["sampler"] + cap["cpb"]).cpu().double()
r_tot, r_s = ref["dvs"], ref["dvs_sampler"]
r_c = r_tot - r_s
J = ref["J_w2"]
push = lambda d: J.t() @ torch.cat((d[..., 0].reshape(-1), d[..., 1].reshape(-1)))
g64 = push(r_tot)
name = os.path.basename(os.environ.get("SMML_LIB", "default"))
print(f"{name:14s} d vs total l2 {l2_err(tot, r_tot):.2e} | sampler part l2 {l2_err(cap['sampler'].cpu(), r_s):.2e} | cpb part l2 {l2_err(cap['cpb'].cpu(), r_c):.2e} "
      f"(|cpb| / |sampler| = {float(r_c.norm() / r_s.norm()):.2f}) || d to_offsets.2.weight via J: total {rel_err(push(tot), g64):.2e}, "
      f"sampler-part error alone {rel_err(push(cap['sampler'].cpu().double() - r_s), g64):.2e}, cpb-part error alone {rel_err(push(cap['cpb'].cpu().double() - r_c), g64):.2e}")
# where does the sampler-part error sit?
e = (cap["sampler"].cpu().double() - r_s)
flat = e.abs().reshape(-1)
top = torch.topk(flat, 6)
print("largest |error| entries of the sampler part (index, error, reference value); total error l2", float(e.norm()), " without the largest:", float((e.reshape(-1)[flat < top.values[0]]).norm()))
for v, i in zip(top.values.tolist(), top.indices.tolist()):
    print(f"   entry {i} (bg {i // (625 * 2)}, key {(i // 2) % 625}, coord {i % 2}): err {v:.3e}  ref {float(r_s.reshape(-1)[i]):+.3e}")
# logit magnitudes per head (does exp2(l log2e - lse log2e) lose bits to cancellation?)
qk = {}
od = Fh.deform_attention
def tap_qk(q, k, v, vs, gq, *a, **kw):
    qk["q"], qk["k"] = q.detach(), k.detach()
    return od(q, k, v, vs, gq, *a, **kw)
smml.deform_attention.Fh.deform_attention = tap_qk
with torch.no_grad():
    net.pathomic_net_tumor(x_path.to(cuda), net.omic_net_tumor(x_omic=x_o.to(cuda))[0])
q, k = qk["q"].view(1, -1, 8, 64).permute(0, 2, 1, 3), qk["k"].view(1, -1, 8, 64).permute(0, 2, 1, 3)
sc = 0.125 * (q @ k.transpose(-1, -2))
print("max |scale q k^T| per head:", [round(float(sc[0, h].abs().max()), 1) for h in range(8)], " row-max mean per head:", [round(float(sc[0, h].amax(-1).mean()), 1) for h in range(8)])

"""Diagnostic, step A (GPU box, ~2 min): fp64 oracle run of the cfg4 tumor branch (100 x 100) with the default build's decisions imposed;
saves to tests/cache/cfg4_tumor_ref.pt the total d vs, its sampler part, and the Jacobian of vs w.r.t. to_offsets.2.weight, so
that step B (tests/tools/diag_dvs_variants.py) can judge any library variant's d vs in seconds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from helpers import decision_tap, params_for, smml, synth
import oracle.deform as odeform
import oracle.mil as omil
from oracle.mil import deform_cross_trans_mil, max_net
from test_oracle_golden import pathomic_args

S = 100
cuda = torch.device("cuda:0")
Fh = smml.functional
args = pathomic_args(input_path_dim=512, batch_size=1)
net = smml.DeformPathomicNet(args)
params = params_for(net, 17, "cfg4")
net.load_state_dict(params); net = net.to(cuda).eval()
x_path = synth.bag(1, S * S, 512, 17, "cfg4:bag"); x_o = synth.normal((1, 59), 17, "cfg4:tumor")
mil, onet = net.pathomic_net_tumor, net.omic_net_tumor
pm = {k[len("pathomic_net_tumor."):]: v for k, v in params.items() if k.startswith("pathomic_net_tumor.")}
po = {k[len("omic_net_tumor."):]: v for k, v in params.items() if k.startswith("omic_net_tumor.")}
w_enc = synth.normal((1, 128), 17, "diag:wenc")
with decision_tap() as tap:
    enc, logits, _, omic_t, vg = mil(x_path.to(cuda), onet(x_omic=x_o.to(cuda))[0])
(enc * w_enc.to(cuda)).sum().backward()
dt = torch.float64
oc = {}
orig_o = omil.deform_cross_attention_2d
def wrapped(a, b, p, **kw):
    out, vgrid, aux = orig_o(a, b, p, return_aux=True, **kw)
    oc["handles"] = [aux[n].register_hook(lambda g, n=n: oc.__setitem__(n, g.detach().clone())) for n in ("vsx", "vsy", "kv")]
    oc["nodes"] = (aux["vsx"], aux["vsy"], aux["kv"], p["to_offsets.2.weight"])
    return out, vgrid
omil.deform_cross_attention_2d = wrapped
p = {k: v.clone().to(dt).requires_grad_() for k, v in pm.items()}
odeform.DECISIONS = tap.decisions()
e, lg, _, vgr = deform_cross_trans_mil(x_path.to(dt), max_net(x_o.to(dt), {k: v.to(dt) for k, v in po.items()})[0], p, grid_hw=(S, S), q_chunk=1024)
(e * w_enc.to(dt)).sum().backward(retain_graph=True)
for h in oc["handles"]:
    h.remove()
vsx, vsy, kv, w2 = oc["nodes"]
dvs = torch.stack((oc["vsx"], oc["vsy"]), -1)
sx, sy = torch.autograd.grad(kv, [vsx, vsy], oc["kv"], retain_graph=True)
dvs_sampler = torch.stack((sx, sy), -1)
# Jacobian of vs (flattened x, y) w.r.t. to_offsets.2.weight: one backward per parameter would be 128 passes; do it per OUTPUT basis instead
# through vjp with random probes is not exact - use 128 forward differences in exact arithmetic?  vs is smooth in w2: take autograd per column.
J = []
flat = torch.cat((vsx.reshape(-1), vsy.reshape(-1)))
n_out = flat.numel()
# J^T d for a batch of basis vectors is what we need later: store the full J via 128 reverse passes on w2's elements is impossible
# (reverse mode goes output -> input); instead store R = [vsx, vsy] graph-free Jacobian by is_grads_batched over outputs in chunks
eye_chunk = 500
rows = []
for i0 in range(0, n_out, eye_chunk):
    i1 = min(n_out, i0 + eye_chunk)
    go = torch.zeros(i1 - i0, n_out, dtype=dt); go[torch.arange(i1 - i0), torch.arange(i0, i1)] = 1.0
    (jw,) = torch.autograd.grad(flat, [w2], go, retain_graph=True, is_grads_batched=True)
    rows.append(jw.reshape(i1 - i0, -1))
J = torch.cat(rows, 0)                                   # [2 * BG * J, 128]
os.makedirs(os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache"), exist_ok=True)
torch.save({"dvs": dvs, "dvs_sampler": dvs_sampler, "J_w2": J, "dw2": w2.grad.detach()},
           os.path.join(os.path.dirname(os.path.abspath(__file__)), "cache", "cfg4_tumor_ref.pt"))
chk = (J.t() @ torch.cat((dvs[..., 0].reshape(-1), dvs[..., 1].reshape(-1)))).reshape(w2.shape)
print("saved; J^T dvs vs autograd d to_offsets.2.weight:", float((chk - w2.grad).abs().max() / w2.grad.abs().max()))

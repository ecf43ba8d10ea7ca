"""Diagnostic: the 2-D attention module on the ACTUAL layer-3 inputs of cfg4's immune / tumor branch (B = 2, S = 100):
gradients of the intermediate tensors (q, vs, kv) HIP vs fp64 oracle."""
import importlib, sys, time
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from helpers import params_for, smml, synth, rel_err, l2_err
from oracle.mil import max_net, linear
from oracle.nystrom import _sub, layer_norm
import oracle.deform as od
from test_oracle_golden import pathomic_args
Fh = smml.functional
dev = torch.device("cuda:0")
B, S = 2, int(sys.argv[1]) if len(sys.argv) > 1 else 100
branch = sys.argv[2] if len(sys.argv) > 2 else "immune"
args = pathomic_args(input_path_dim=512, batch_size=B)
net = smml.DeformPathomicNet(args)
params = params_for(net, 17, "cfg4")
x_path = synth.bag(B, S * S, 512, 17, "cfg4:bag").double()
x_o = (synth.normal((B, 361), 17, "cfg4:immune") if branch == "immune" else synth.normal((B, 59), 17, "cfg4:tumor")).double()
pm = {k: v.double() for k, v in _sub(params, f"pathomic_net_{branch}.").items()}
omic, _ = max_net(x_o, {k: v.double() for k, v in _sub(params, f"omic_net_{branch}.").items()})
path = torch.relu(linear(x_path, pm, "_fc1.0."))
N = path.shape[1]
h = linear(torch.cat((path, omic.unsqueeze(1).repeat(1, N, 1)), dim=-1), pm, "fusion_layer.fusion_layer.")
pl = _sub(pm, "layer3.")
a = layer_norm(h, pl, "norm."); b = layer_norm(path, pl, "norm.")          # token-major [B, N, C]
pa = _sub(pl, "attn2d.")
wo = synth.normal((B, N, 128), 3, "diag:wo").double()
print(f"branch {branch}: |a| max {float(a.abs().max()):.3f}  |b| max {float(b.abs().max()):.3f}  omic max {float(omic.max()):.3f}")
# ---- oracle fp64 with intermediates
a64 = a.clone().requires_grad_(); b64 = b.clone().requires_grad_()
p64 = {k: v.clone().requires_grad_() for k, v in pa.items()}
o, vg, aux = od.deform_cross_attention_2d(a64.transpose(1, 2), b64.transpose(1, 2), p64, grid_hw=(S, S), q_chunk=1024, return_aux=True)
for t in (aux["q"], aux["vsx"], aux["vsy"], aux["kv"]):
    t.retain_grad()
(o.transpose(1, 2) * wo).sum().backward()
# ---- HIP with hooks on the intermediates
mod = smml.DeformCrossAttention2D(dim=128, dropout=0.1, grid_hw=(S, S))
mod.load_state_dict({k: v.float() for k, v in pa.items()}); mod = mod.to(dev).eval()
cap = {}
orig_offsets, orig_sample, orig_gp = Fh.offsets, Fh.bilinear_sample, Fh.grouped_pointwise
def offsets(*a_, **k_):
    vgrid, vs = orig_offsets(*a_, **k_); vs.retain_grad(); cap["vs"] = vs; return vgrid, vs
def sample(*a_, **k_):
    kv = orig_sample(*a_, **k_); kv.retain_grad(); cap["kv"] = kv; return kv
first = [True]
def gp(x, w, g=1):
    y = orig_gp(x, w, g)
    if first[0]:
        first[0] = False; y.retain_grad(); cap["q"] = y
    return y
Fh.offsets, Fh.bilinear_sample, Fh.grouped_pointwise = offsets, sample, gp
ad = a.float().to(dev).requires_grad_(); bd = b.float().to(dev).requires_grad_()
out = mod.forward_tokens(ad, bd, False)
(out * wo.float().to(dev)).sum().backward()
vs64 = torch.stack((aux["vsx"].grad, aux["vsy"].grad), -1)
print(f"   out   l2 {l2_err(out, o.transpose(1, 2)):.2e}")
print(f"   dq    l2 {l2_err(cap['q'].grad, aux['q'].grad):.2e}   max {rel_err(cap['q'].grad, aux['q'].grad):.2e}")
print(f"   dkv   l2 {l2_err(cap['kv'].grad, aux['kv'].grad):.2e}")
print(f"   dvs   l2 {l2_err(cap['vs'].grad, vs64):.2e}   max {rel_err(cap['vs'].grad, vs64):.2e}")
g, g64 = cap["vs"].grad.double().cpu(), vs64
e = (g - g64).abs() / g64.abs().max()
idx = torch.nonzero(e > 1e-4)
print("   dvs entries with err > 1e-4 of scale:", idx.shape[0], idx[:10].tolist(), [f"{float(e[tuple(i)]):.1e}" for i in idx[:10]])
print(f"   dx1   l2 {l2_err(ad.grad, a64.grad):.2e}   dx2 l2 {l2_err(bd.grad, b64.grad):.2e}")
for k, p in mod.named_parameters():
    if p.grad is not None and p64[k].grad is not None and float(p64[k].grad.abs().max()) > 0:
        print(f"   d{k:<32s} l2 {l2_err(p.grad, p64[k].grad):.2e}")
vsd = cap["vs"].detach().cpu()
gq = smml.deform_attention._grid_queries_2d(S, S, "cpu")
for (bg, key, comp) in idx[:5].tolist():
    v = vsd[bg, key, comp]
    hit = (gq[:, comp] == v)
    print(f"   entry {bg, key, comp}: vs = {float(v):.9f}; exact coincidences with the query grid: {int(hit.sum())}; nearest grid distance {float((gq[:, comp] - v).abs().min()):.3e}")
    ix = ((v + 1.0) * S - 1.0) / 2.0
    print(f"      pixel coordinate {float(ix):.7f} (distance to an integer {abs(float(ix) - round(float(ix))):.2e})")

"""Diagnostic (GPU, ~2 min): which half of d vs carries the error that leaves the tumor branch's to_offsets.* gradients at ~4.5 x the
CPU-fp32 noise in the full-size cfg4 case?  d vs = (position-bias backward) + (bilinear sampler backward): the two
consumers of vs get their own copy of the tensor on the HIP side, so each part is captured separately, and each is pushed through the
fp64 offsets network to see what it does to d to_offsets.2.weight.  Loss: enc . w only (NO direct vgrid term: B = 1 makes BatchLoss
identically zero in the cfg4 test too).  Usage: python tests/tools/diag_r4_dvs_split.py [tumor|immune] [S]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from helpers import decision_tap, l2_err, params_for, rel_err, smml, synth
import oracle.deform as odeform
import oracle.mil as omil
from oracle.mil import deform_cross_trans_mil, max_net
from test_oracle_golden import pathomic_args

branch = sys.argv[1] if len(sys.argv) > 1 else "tumor"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 100
SEED = int(sys.argv[3]) if len(sys.argv) > 3 else 17          # 17 = the cfg4 test's parameters and bag; other seeds: other realisations
cuda = torch.device("cuda:0")
Fh = smml.functional
args = pathomic_args(input_path_dim=512, batch_size=1)
net = smml.DeformPathomicNet(args)
params = params_for(net, SEED, "cfg4")
net.load_state_dict(params); net = net.to(cuda).eval()
x_path = synth.bag(1, S * S, 512, SEED, "cfg4:bag")
x_o = synth.normal((1, 59), SEED, "cfg4:tumor") if branch == "tumor" else synth.normal((1, 361), SEED, "cfg4:immune")
mil, onet = getattr(net, f"pathomic_net_{branch}"), getattr(net, f"omic_net_{branch}")
pm = {k[len(f"pathomic_net_{branch}."):]: v for k, v in params.items() if k.startswith(f"pathomic_net_{branch}.")}
po = {k[len(f"omic_net_{branch}."):]: v for k, v in params.items() if k.startswith(f"omic_net_{branch}.")}
w_enc = synth.normal((1, 128), SEED, "diag:wenc")

cap = {}
orig_a, orig_s = Fh.deform_attention, Fh.bilinear_sample
def tapped_a(q, k, v, vs, gq, *a, **kw):
    vs2 = vs.clone()                                                # own node: its gradient is the position-bias part only
    vs2.register_hook(lambda g: cap.__setitem__("cpb", g.detach().clone()))
    for n, t in (("dq", q), ("dk", k), ("dv", v)):
        t.register_hook(lambda g, n=n: cap.__setitem__(n, g.detach().clone()))
    return orig_a(q, k, v, vs2, gq, *a, **kw)
def tapped_s(x, vs, **kw):
    vs3 = vs.clone()
    vs3.register_hook(lambda g: cap.__setitem__("sampler", g.detach().clone()))
    cap["_sampler_in"] = (x.detach(), vs.detach(), dict(kw))
    out = orig_s(x, vs3, **kw)
    out.register_hook(lambda g: cap.__setitem__("dkv", g.detach().clone()))
    return out
Fh.deform_attention = tapped_a; Fh.bilinear_sample = tapped_s
with decision_tap() as tap:
    enc, logits, _, omic_t, vg = mil(x_path.to(cuda), onet(x_omic=x_o.to(cuda))[0])
(enc * w_enc.to(cuda)).sum().backward()
Fh.deform_attention = orig_a; Fh.bilinear_sample = orig_s
sampler_in = cap.pop("_sampler_in")
hip = {k: v.double().cpu() for k, v in cap.items()}

res = {}
orig_o = omil.deform_cross_attention_2d
for dt in (torch.float32, torch.float64):
    oc = {}
    def wrapped(a, b, p, **kw):
        out, vgrid, aux = orig_o(a, b, p, return_aux=True, **kw)
        for n in ("vsx", "vsy", "kv", "k", "v"):
            aux[n].register_hook(lambda g, n=n: oc.__setitem__(n, g.detach().clone()))
        oc["nodes"] = (aux["vsx"], aux["vsy"], aux["kv"], p["to_offsets.2.weight"])
        oc["kvnodes"] = (aux["k"], aux["v"])
        return out, vgrid
    omil.deform_cross_attention_2d = wrapped
    p = {k: v.clone().to(dt).requires_grad_() for k, v in pm.items()}
    odeform.DECISIONS = tap.decisions()
    e, lg, _, vgr = deform_cross_trans_mil(x_path.to(dt), max_net(x_o.to(dt), {k: v.to(dt) for k, v in po.items()})[0], p, grid_hw=(S, S), q_chunk=1024)
    (e * w_enc.to(dt)).sum().backward(retain_graph=True)
    vsx, vsy, kv, w2 = oc["nodes"]
    tot = torch.stack((oc["vsx"], oc["vsy"]), -1)
    sx, sy = torch.autograd.grad(kv, [vsx, vsy], oc["kv"], retain_graph=True)
    smp = torch.stack((sx, sy), -1)
    res[dt] = dict(total=tot.double(), sampler=smp.double(), cpb=(tot - smp).double(), dkv=oc["kv"].double(), w2g=w2.grad.detach().double(),
                   nodes=(vsx, vsy, w2), kv=kv, knode=oc["kvnodes"][0], vnode=oc["kvnodes"][1], dk=oc["k"].double(), dv=oc["v"].double())
omil.deform_cross_attention_2d = orig_o
o32, o64 = res[torch.float32], res[torch.float64]
vsx, vsy, w2 = o64["nodes"]

def through(dvs):            # d to_offsets.2.weight that a given d vs produces through the fp64 offsets network
    (g,) = torch.autograd.grad([vsx, vsy], [w2], [dvs[..., 0].reshape(vsx.shape), dvs[..., 1].reshape(vsy.shape)], retain_graph=True)
    return g
print(f"branch {branch} S {S} seed {SEED}; shapes: d vs {tuple(o64['total'].shape)}")
hip["total"] = hip["cpb"] + hip["sampler"]
print("part       | HIP vs fp64 (max-rel / l2)  | CPU fp32 oracle vs fp64      | |part| / |total|")
for n in ("total", "cpb", "sampler", "dkv"):
    a = hip[n].reshape(o64[n].shape)
    sc = float(o64[n].abs().max() / o64["total"].abs().max()) if n != "dkv" else float("nan")
    print(f"  {n:8s} | {rel_err(a, o64[n]):.2e} / {l2_err(a, o64[n]):.2e}       | {rel_err(o32[n], o64[n]):.2e} / {l2_err(o32[n], o64[n]):.2e}        | {sc:.2f}")
ref = through(o64["total"])
print("d to_offsets.2.weight through the fp64 offsets network, error relative to its own max:")
for label, d in (("HIP total", hip["total"].reshape(o64["total"].shape)), ("fp32-oracle total", o32["total"]),
                 ("HIP cpb + exact sampler", hip["cpb"].reshape(o64["cpb"].shape) + o64["sampler"]),
                 ("exact cpb + HIP sampler", o64["cpb"] + hip["sampler"].reshape(o64["sampler"].shape)),
                 ("fp32-oracle cpb + exact sampler", o32["cpb"] + o64["sampler"]), ("exact cpb + fp32-oracle sampler", o64["cpb"] + o32["sampler"])):
    print(f"  {label:34s} {rel_err(through(d), ref):.2e}")
# is the HIP error of a part a plain multiple of the part (a scale error) or of the other part?
for n in ("cpb", "sampler"):
    a = hip[n].reshape(o64[n].shape); err = (a - o64[n]).reshape(-1)
    for m in ("cpb", "sampler", "total"):
        b = o64[m].reshape(-1)
        c = float(err @ b / (b @ b))
        print(f"  projection of the HIP {n} error on the exact {m}: coefficient {c:+.2e}, explains {float((c * b).norm() / err.norm()):.2f} of the error norm")

# ---- where does the sampler part's error come from?  (B) the fp64 sampler backward fed with HIP's d kv; (A) HIP's sampler kernel fed with
# the exact d kv; and d kv itself rebuilt in fp64 from HIP's dk / dv one at a time
kv64, kn, vn = o64["kv"], o64["knode"], o64["vnode"]
def sampler64(dkv):
    sx, sy = torch.autograd.grad(kv64, [vsx, vsy], dkv.reshape(kv64.shape), retain_graph=True)
    return torch.stack((sx, sy), -1)
def dkv64(dk, dv):
    (g,) = torch.autograd.grad([kn, vn], [kv64], [dk.reshape(kn.shape), dv.reshape(vn.shape)], retain_graph=True)
    return g
x_in, vs_in, kw_in = sampler_in
vs_l = vs_in.clone().requires_grad_()
out2 = orig_s(x_in, vs_l, **kw_in)
out2.backward(o64["dkv"].reshape(out2.shape).float().to(cuda))
dvs_A = vs_l.grad.double().cpu().reshape(o64["sampler"].shape)
print("sampler part rebuilt (d to_offsets.2.weight error through the fp64 offsets network, exact cpb part added):")
for label, d in (("(A) HIP sampler kernel on the EXACT d kv", dvs_A), ("(B) fp64 sampler on HIP's d kv", sampler64(hip["dkv"])),
                 ("    fp64 sampler on the fp32 oracle's d kv", sampler64(o32["dkv"])),
                 ("(C) fp64 sampler + fp64 to_k/to_v bwd on HIP dk, exact dv", sampler64(dkv64(hip["dk"], o64["dv"]))),
                 ("(D) fp64 sampler + fp64 to_k/to_v bwd on exact dk, HIP dv", sampler64(dkv64(o64["dk"], hip["dv"]))),
                 ("    same with the fp32 oracle's dk, exact dv", sampler64(dkv64(o32["dk"], o64["dv"]))),
                 ("    same with exact dk, the fp32 oracle's dv", sampler64(dkv64(o64["dk"], o32["dv"])))):
    print(f"  {label:62s} {rel_err(through(o64['cpb'] + d), ref):.2e}")
print(f"  HIP dk vs fp64 {rel_err(hip['dk'].reshape(o64['dk'].shape), o64['dk']):.2e} (fp32 oracle {rel_err(o32['dk'], o64['dk']):.2e});  "
      f"HIP dv vs fp64 {rel_err(hip['dv'].reshape(o64['dv'].shape), o64['dv']):.2e} (fp32 oracle {rel_err(o32['dv'], o64['dv']):.2e})")

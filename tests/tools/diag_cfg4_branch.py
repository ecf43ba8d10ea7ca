"""Diagnostic (GPU): where does the tumor branch of the full-size config-4 case lose accuracy?  Runs ONE DeformCrossTransMIL branch
with the cfg4 parameters / inputs at 100 x 100 on the HIP path and on the oracle (fp32 + fp64, the kernels' decisions imposed) and
prints, for the attention module's intermediate tensors, the gradient errors of both against fp64.
Usage: python tests/tools/diag_cfg4_branch.py [tumor|immune] [S]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import helpers
from helpers import decision_tap, l2_err, params_for, rel_err, smml, synth
import oracle.deform as odeform
import oracle.mil as omil
from oracle.mil import deform_cross_trans_mil, max_net
from oracle.nystrom import _sub
from test_oracle_golden import pathomic_args

branch = sys.argv[1] if len(sys.argv) > 1 else "tumor"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 100
VG_W = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3          # weight of the direct vgrid term of the diagnostic loss
cuda = torch.device("cuda:0")
Fh = smml.functional
B = 1
args = pathomic_args(input_path_dim=512, batch_size=B)
net = smml.DeformPathomicNet(args)
params = params_for(net, 17, "cfg4")
net.load_state_dict(params); net = net.to(cuda).eval()
x_path = synth.bag(B, S * S, 512, 17, "cfg4:bag")
x_o = synth.normal((B, 59), 17, "cfg4:tumor") if branch == "tumor" else synth.normal((B, 361), 17, "cfg4:immune")
mil = getattr(net, f"pathomic_net_{branch}")
onet = getattr(net, f"omic_net_{branch}")
pm = {k[len(f"pathomic_net_{branch}."):]: v for k, v in params.items() if k.startswith(f"pathomic_net_{branch}.")}
po = {k[len(f"omic_net_{branch}."):]: v for k, v in params.items() if k.startswith(f"omic_net_{branch}.")}
w_enc = synth.normal((B, 128), 17, "diag:wenc")

# ---- HIP, with the inputs of the fused core captured
cap = {}
orig = Fh.deform_attention
def tapped(q, k, v, vs, gq, *a, **kw):
    for n, t in (("q", q), ("k", k), ("v", v), ("vs_attn", vs)):
        if n == "vs_attn":
            cap["vs_value"] = vs.detach().clone()
        t.register_hook(lambda g, n=n: cap.__setitem__(n, g.detach().clone()))
    return orig(q, k, v, vs, gq, *a, **kw)
Fh.deform_attention = tapped
smml.deform_attention.Fh.deform_attention = tapped
orig_s = Fh.bilinear_sample
def tapped_s(x, vs, **kw):
    vs.register_hook(lambda g: cap.__setitem__("vs_sample", g.detach().clone()))
    x.register_hook(lambda g: cap.__setitem__("x2map", g.detach().clone()))
    return orig_s(x, vs, **kw)
Fh.bilinear_sample = tapped_s
with decision_tap() as tap:
    feat_o = onet(x_omic=x_o.to(cuda))[0]
    enc, logits, _, omic_t, vg = mil(x_path.to(cuda), feat_o)
((enc * w_enc.to(cuda)).sum() + VG_W * vg.pow(2).sum()).backward()
hip = {k: v.cpu() for k, v in cap.items()}
hip_p = {k: p.grad.detach().cpu() for k, p in mil.named_parameters() if p.grad is not None}

# ---- oracle with the same captures
res = {}
orig_o = omil.deform_cross_attention_2d
for dt in (torch.float32, torch.float64):
    oc = {}
    def wrapped(a, b, p, **kw):
        out, vgrid, aux = orig_o(a, b, p, return_aux=True, **kw)
        for n in ("q", "k", "v", "vsx", "vsy", "kv"):
            aux[n].register_hook(lambda g, n=n: oc.__setitem__(n, g.detach().clone()))
        oc["_vs_nodes"] = (aux["vsx"], aux["vsy"], p["to_offsets.2.weight"], p["to_offsets.0.weight"])
        return out, vgrid
    omil.deform_cross_attention_2d = wrapped
    p = {k: v.clone().to(dt).requires_grad_() for k, v in pm.items()}
    pom = {k: v.clone().to(dt) for k, v in po.items()}
    odeform.DECISIONS = tap.decisions()
    fo, _ = max_net(x_o.to(dt), pom)
    e, lg, _, vgr = deform_cross_trans_mil(x_path.to(dt), fo, p, grid_hw=(S, S), q_chunk=1024)
    ((e * w_enc.to(dt)).sum() + VG_W * vgr.pow(2).sum()).backward(retain_graph=(dt == torch.float64))
    if dt == torch.float64:
        # push a given d vs through the fp64 offsets network: which part of the weight-gradient error does HIP's d vs explain?
        vsx_n, vsy_n, w2_n, w0_n = oc["_vs_nodes"]
        def through(dvs):
            gx, gy = dvs[..., 0].to(dt), dvs[..., 1].to(dt)
            return torch.autograd.grad([vsx_n, vsy_n], [w2_n, w0_n], [gx, gy], retain_graph=True)
        oc["_through"] = through
    res[dt] = (oc, {k: v.grad for k, v in p.items() if v.grad is not None}, e.detach(), vgr.detach())
omil.deform_cross_attention_2d = orig_o
o32, o64 = res[torch.float32], res[torch.float64]
print(f"branch {branch} S {S}: enc err {rel_err(enc, o64[2]):.2e} (oracle32 {rel_err(o32[2], o64[2]):.2e})  vgrid {rel_err(vg, o64[3]):.2e}")
print("intermediate gradients (max-rel / l2): HIP vs fp64 | oracle fp32 vs fp64")
for n in ("q", "k", "v"):
    print(f"  d{n:10s} {rel_err(hip[n], o64[0][n]):.2e} / {l2_err(hip[n], o64[0][n]):.2e} | {rel_err(o32[0][n], o64[0][n]):.2e} / {l2_err(o32[0][n], o64[0][n]):.2e}")
print("per-head l2 errors (HIP | torch fp32):")
for n in ("q", "k", "v"):
    hh = hip[n].double().view(1, -1, 8, 64); a32 = o32[0][n].double().view(1, -1, 8, 64); a64 = o64[0][n].view(1, -1, 8, 64)
    print(f"  d{n}: " + "  ".join(f"h{h} {float((hh[:, :, h] - a64[:, :, h]).norm() / a64[:, :, h].norm()):.1e}|{float((a32[:, :, h] - a64[:, :, h]).norm() / a64[:, :, h].norm()):.1e}" for h in range(8)))
print("parameter gradients with err / noise > 2:")
for k in sorted(hip_p):
    if k in o64[1]:
        e, nz = rel_err(hip_p[k], o64[1][k]), rel_err(o32[1][k], o64[1][k])
        if e > 2 * nz and e > 2e-5:
            print(f"  {k:60s} {e:.2e}  noise {nz:.2e}  ratio {e / max(nz, 1e-30):.1f}")
thr = o64[0]["_through"]
dvs64 = torch.stack((o64[0]["vsx"], o64[0]["vsy"]), dim=-1)
dvs32 = torch.stack((o32[0]["vsx"], o32[0]["vsy"]), dim=-1)
a64, a_hip, a_32 = thr(dvs64), thr(hip["vs_attn"]), thr(dvs32)
for i, nm in enumerate(("to_offsets.2.weight", "to_offsets.0.weight")):
    print(f"  {nm} from d vs pushed through the fp64 offsets network: HIP d vs {rel_err(a_hip[i], a64[i]):.2e} | fp32-oracle d vs {rel_err(a_32[i], a64[i]):.2e}   "
          f"(the part of d {nm} that comes through d vs only)")
dvs_hip = hip["vs_attn"]                                   # hooks on the same tensor: the total d vs [(B G), J, 2]
for nm, oo in (("fp64", o64), ("fp32", o32)):
    oo[0]["vs"] = torch.stack((oo[0]["vsx"], oo[0]["vsy"]), dim=-1)
print(f"  d vs (total) {rel_err(dvs_hip, o64[0]['vs']):.2e} / {l2_err(dvs_hip, o64[0]['vs']):.2e} | {rel_err(o32[0]['vs'], o64[0]['vs']):.2e} / {l2_err(o32[0]['vs'], o64[0]['vs']):.2e}   max |d vs| {float(o64[0]['vs'].abs().max()):.3e}")


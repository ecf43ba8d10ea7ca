"""Diagnostic (GPU): the full cfg4[1-100] comparison with a THIRD evaluation - the oracle (plain PyTorch) run in fp32 ON THE GPU
(ATen / rocBLAS kernels) - beside the CPU fp32 and CPU fp64 oracle runs: is the HIP path's distance to fp64 within what two fp32
evaluations of the reference's op sequence on different back ends differ by?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import oracle.deform as odeform
from helpers import decision_tap, l2_err, params_for, rel_err, smml, synth
from oracle.losses import batch_loss, orthogonal_loss
from oracle.mil import deform_pathomic_net
from test_oracle_golden import pathomic_args
cuda = torch.device("cuda:0")
B, S = 1, 100
args = pathomic_args(input_path_dim=512, batch_size=B)
net = smml.DeformPathomicNet(args)
params = params_for(net, 17, "cfg4")
net.load_state_dict(params); net = net.to(cuda).eval()
x_path = synth.bag(B, S * S, 512, 17, "cfg4:bag")
x_t = synth.normal((B, 59), 17, "cfg4:tumor"); x_i = synth.normal((B, 361), 17, "cfg4:immune")
label = torch.tensor([2, 0])[:B]
def total(feats, vt, vi, lg, bl, ol):
    l_t, l_i = bl(lg[3], lg[4]), bl(lg[5], lg[6])
    return (torch.nn.functional.cross_entropy(lg[2], label.to(lg[2].device)) + 0.5 * l_t.sum() + 0.5 * l_i.sum() + 0.1 * ol(vt, vi, vi, vt).sum())
with decision_tap() as tap:
    feats, vt, vi, lg, _, _, _ = net(x_path=x_path.to(cuda), x_omic=None, x_omic_tumor=x_t.to(cuda), x_omic_immune=x_i.to(cuda))
total(feats, vt, vi, lg, smml.BatchLoss(B, 1), smml.OrthogonalLoss()).backward()
runs = {}
for name, dt, dev in (("cpu32", torch.float32, "cpu"), ("gpu32", torch.float32, cuda), ("cpu64", torch.float64, "cpu")):
    p = {k: (v.clone().to(dev, dt).requires_grad_() if v.dtype.is_floating_point else v.to(dev)) for k, v in params.items()}
    odeform.DECISIONS = tap.decisions()
    f, a, b, l = deform_pathomic_net(x_path.to(dev, dt), x_t.to(dev, dt), x_i.to(dev, dt), p, grid_hw=(S, S), q_chunk=1024)
    total(f, a, b, l, lambda o, v: batch_loss(o, v, B), orthogonal_loss).backward()
    runs[name] = {k: v.grad.detach().cpu() for k, v in p.items() if getattr(v, "grad", None) is not None}
r64 = runs["cpu64"]
print(f"{'tensor':62s} {'HIP':>9s} {'cpu fp32':>9s} {'gpu fp32':>9s}   (max-norm distance to fp64)")
for k, pp in net.named_parameters():
    if pp.grad is None or k not in r64 or float(r64[k].abs().max()) < 1e-12:
        continue
    e = rel_err(pp.grad, r64[k]); c = rel_err(runs["cpu32"][k], r64[k]); g = rel_err(runs["gpu32"][k], r64[k])
    if e > 4e-5 or g > 4e-5:
        print(f"{k:62s} {e:9.2e} {c:9.2e} {g:9.2e}")

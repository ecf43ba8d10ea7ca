"""Does the approximate table mode train like the exact modes?  The headline model (DeformCrossTransMIL, 8 bags of 10 000 x 512, CE + BatchLoss, Adam
lr 2e-4, dropout 0.1 with the same seeds in every run) is trained for STEPS steps from the same initialisation on one fixed synthetic batch in four
arithmetic modes - fp32-grade, 16-bit MLP, cpb_table='forward', cpb_table=True - and the loss trajectories are compared with the fp32-grade one.
Usage: python tests/tools/diag_table_training.py [steps]"""
import argparse, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
pkg = importlib.import_module(bench.PKG)
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 60
dev = torch.device("cuda:0")
B, S, in_dim = 8, 100, 512
N = S * S
path = pkg.synth.bag(B, N, in_dim, 42, "bench:bag").to(dev)
omic = torch.relu(pkg.synth.normal((B, 128), 42, "bench:omicvec")).to(dev)
label = torch.randint(0, 4, (B,), generator=torch.Generator().manual_seed(0)).to(dev)
runs = {}
for name, dt, tab in (("fp32-grade", None, False), ("bf16 MLP", "bf16", False), ("bf16 table forward", "bf16", "forward"), ("bf16 full table", "bf16", True)):
    torch.manual_seed(1234)
    mil = pkg.DeformCrossTransMIL(bench.mil_args(in_dim, dt, tab))
    mil.load_state_dict(pkg.synth.fill_params({k: tuple(v.shape) for k, v in mil.state_dict().items()}, 42, "bench"))
    mil = mil.to(dev).train()
    opt = torch.optim.Adam(mil.parameters(), lr=2e-4)
    bloss = pkg.BatchLoss(B, 1)
    torch.manual_seed(99)                      # dropout seeds: the same sequence in every run
    losses, cpb_norm = [], []
    for it in range(STEPS):
        enc, logits, _, omic_t, vgrid = mil(path, omic)
        loss = torch.nn.functional.cross_entropy(logits, label) + torch.sum(bloss(omic_t, vgrid))
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    w = torch.cat([p.detach().flatten() for n, p in mil.named_parameters() if "attn2d.rel_pos_bias" in n])
    runs[name] = (losses, w.clone())
    print(f"{name:20s} loss: step 0 {losses[0]:.5f}  10 {losses[min(10, STEPS - 1)]:.5f}  30 {losses[min(30, STEPS - 1)]:.5f}  last {losses[-1]:.5f}", flush=True)
ref_l, ref_w = runs["fp32-grade"]
w0 = None
print("\nrelative to the fp32-grade run: max |loss - loss_fp32| / loss_fp32 over the steps; position-bias MLP weights after training (relative l2 distance, and the")
print("distance the fp32-grade run itself moved them from the initialisation, for scale)")
mil0 = pkg.DeformCrossTransMIL(bench.mil_args(in_dim, None, False))
mil0.load_state_dict(pkg.synth.fill_params({k: tuple(v.shape) for k, v in mil0.state_dict().items()}, 42, "bench"))
w_init = torch.cat([p.detach().flatten() for n, p in mil0.named_parameters() if "attn2d.rel_pos_bias" in n]).to(dev)
moved = float((ref_w - w_init).norm() / w_init.norm())
for name, (l, w) in runs.items():
    dl = max(abs(a - b) / abs(b) for a, b in zip(l, ref_l))
    print(f"  {name:20s} max rel. loss deviation {dl:.3e}   CPB weights vs fp32-grade run {float((w - ref_w).norm() / ref_w.norm()):.3e}   (fp32-grade run moved them by {moved:.3e})")

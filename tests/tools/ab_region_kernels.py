"""Interleaved A/B of builds of the library on the region forward / backward entry points at the headline shape (8 bags x 8 heads x
10 000 queries x 625 keys, dropout 0.1): every build named on the command line ("default" = lib/libsmml_hip.so, otherwise
lib/variants/<name>.so from tests/tools/build_variants.py) is loaded into THIS process and called in turn, round after round, on the same
tensors, so that clock / thermal drift of the box hits all of them alike (single runs of bench.py differ by +-5 % on one box).
Prints the median and the minimum per build: whole forward call, whole backward call (HIP events around the C call).
usage: python tests/tools/ab_region_kernels.py default v_a v_b [--rounds 12]"""
import ctypes as C
import importlib
import os
import statistics
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
smml = importlib.import_module("subspace-multimodal-learning_amd")
capi = importlib.import_module("subspace-multimodal-learning_amd._capi")
Fh = smml.functional

argv = sys.argv[1:]
rounds = 12
if "--rounds" in argv:
    i = argv.index("--rounds")
    rounds = int(argv[i + 1])
    del argv[i:i + 2]
names = argv or ["default"]
PKG = os.path.join(ROOT, "subspace-multimodal-learning_amd")
paths = {n: os.path.join(PKG, "lib", "libsmml_hip.so") if n == "default" else os.path.join(PKG, "lib", "variants", n + ".so") for n in names}

dev = torch.device("cuda", 0)
B, S, T, H = 8, 100, 25, 8
N, J = S * S, T * T
gen = torch.Generator().manual_seed(5)
rn = lambda *s: torch.randn(*s, generator=gen)
shapes = {"mlp.0.0.weight": (32, 2), "mlp.0.0.bias": (32,), "mlp.1.0.weight": (32, 32), "mlp.1.0.bias": (32,), "mlp.2.weight": (1, 32), "mlp.2.bias": (1,)}
p = smml.synth.fill_params({"layer3.attn2d.rel_pos_bias." + k: v for k, v in shapes.items()}, 42, "bench")
w = [p["layer3.attn2d.rel_pos_bias." + k].to(dev).contiguous() for k in shapes]
ax = 2.0 * torch.arange(S, dtype=torch.float32) / (S - 1) - 1.0
gq = torch.stack((ax.view(1, S).expand(S, S), ax.view(S, 1).expand(S, S)), dim=-1).reshape(N, 2).contiguous().to(dev)
off = torch.tanh(rn(B * H, 2, T, T) * 0.7) * 4.0
gx = torch.arange(T, dtype=torch.float32).view(1, T).expand(T, T)
vs = (2.0 * (torch.stack((gx, gx.t()), 0)[None] + off) / (T - 1) - 1.0).permute(0, 2, 3, 1).reshape(B * H, J, 2).contiguous().to(dev)
q, k, v = (rn(B, N, 512) * 0.4).to(dev), (rn(B, J, 512) * 0.4).to(dev), rn(B, J, 512).to(dev)
dout = rn(B, N, 512).to(dev)
pmax = Fh.table_pmax(1.0, float(vs.abs().max()))
tables = Fh.cpb_regions_build(*w, pmax)                   # built by the default library: the same tables for every build
tv = Fh.region_tables_view(tables)
print("tables:", {k2: tv[k2] for k2 in ("n_regions", "n_sub", "n_edge", "overflow")}, flush=True)
L0 = capi.lib()
nst = L0.smml_deform_attn_nst(N)
out = torch.empty(B, N, 512, device=dev)
lse = torch.empty(B, H, N, device=dev)
logits = torch.empty(B, H, nst // 32, J, 32, device=dev)
rid = torch.empty(B, H, nst // 32, J, 32, device=dev, dtype=torch.int16)
dlogits = torch.empty_like(logits)
dq, dk, dv, dvs = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v), torch.empty_like(vs)
dws = [torch.empty_like(t) for t in w]
wsb = L0.smml_deform_attn_region_bwd_workspace_bytes(B, N, J, H)
ws = torch.empty(wsb, device=dev, dtype=torch.uint8)


def load(path):
    h = C.CDLL(path)
    for fn in ("smml_deform_attn_region_fwd_f32", "smml_deform_attn_region_bwd_f32"):
        res, args = capi.SIGNATURES[fn]
        getattr(h, fn).restype, getattr(h, fn).argtypes = res, args
    return h


libs = {n: load(pth) for n, pth in paths.items()}
st = capi.stream()
opts = None


def fwd(h, seed):
    return h.smml_deform_attn_region_fwd_f32(capi.fptr(q), capi.fptr(k), capi.fptr(v), capi.fptr(vs), capi.fptr(gq), *[capi.fptr(t) for t in w],
                                             capi.ptr(tables), capi.fptr(out), capi.fptr(lse), capi.fptr(logits), capi.ptr(rid), B, N, J, H, 0.125, 0.1, seed,
                                             None, None, st, opts)


def bwd(h, seed):
    return h.smml_deform_attn_region_bwd_f32(capi.fptr(q), capi.fptr(k), capi.fptr(v), capi.fptr(vs), capi.fptr(gq), *[capi.fptr(t) for t in w],
                                             capi.ptr(tables), capi.fptr(out), capi.fptr(dout), capi.fptr(lse), capi.fptr(logits), capi.ptr(rid),
                                             capi.fptr(dlogits), capi.fptr(dq), capi.fptr(dk), capi.fptr(dv), capi.fptr(dvs), *[capi.fptr(t) for t in dws],
                                             capi.ptr(ws), wsb, B, N, J, H, 0.125, 0.1, seed, None, None, st, opts)


def timed(fn, h, seed):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    rc = fn(h, seed)
    e1.record()
    assert rc == 0, (rc, L0.smml_last_error())
    return e0, e1


tf, tb = {n: [] for n in names}, {n: [] for n in names}
for r in range(rounds + 2):
    evs = []
    for n in names:                                        # forward of build n, then its backward on what that forward saved
        evs.append((n, timed(fwd, libs[n], r), timed(bwd, libs[n], r)))
    torch.cuda.synchronize()
    if r >= 2:
        for n, (a0, a1), (b0, b1) in evs:
            tf[n].append(a0.elapsed_time(a1)); tb[n].append(b0.elapsed_time(b1))
ids = rid.view(torch.int32).to(torch.int64)            # the last forward's region ids (two per word)
ids = torch.stack((ids & 0xFFFF, (ids >> 16) & 0xFFFF))
print("pairs without a region:", float((ids == 0xFFFF).float().mean()), "beyond the LDS-resident regions:", float(((ids >= 2048) & (ids != 0xFFFF)).float().mean()), flush=True)
for n in names:
    print(f"{n:12s} forward median {statistics.median(tf[n]):.3f} min {min(tf[n]):.3f} ms | backward median {statistics.median(tb[n]):.3f} min {min(tb[n]):.3f} ms", flush=True)

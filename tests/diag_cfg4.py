"""Diagnostic: cfg4 test body at B = 2, S = 100 with per-tensor errors printed (not a test)."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import helpers
import test_gpu_configs as t
try:
    t.test_cfg4_full_fusion_10000x512.__wrapped__ if hasattr(t.test_cfg4_full_fusion_10000x512, "__wrapped__") else None
    t.test_cfg4_full_fusion_10000x512(torch.device("cuda:0"), int(sys.argv[1]), int(sys.argv[2]))
    print("PASSED")
except AssertionError as e:
    print("FAILED:", str(e)[:3000])
rows = [r for r in helpers.REPORT if r[5] in ("l2",) and r[3] == r[3]]
rows.sort(key=lambda r: -(r[2] / max(r[3], 1e-30)))
for r in rows[:40]:
    print(f"{r[1]:<75s} l2 err {r[2]:.2e} noise {r[3]:.2e} ratio {r[2] / max(r[3], 1e-30):.1f}")

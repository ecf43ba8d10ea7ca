"""Per-kernel summary of the rocprofv3 --pmc passes of tests/tools/gpu_r5_pmc.sh (gpurun_out/r5/pmc_{a,b,c,f,w}) -> stdout and
gpurun_out/r5/<tag>_pmc_summary.txt.  SQ_* cycle counters are per-SIMD quad-cycles summed over the chip (MI355X_MICROARCH.md)."""
import csv, glob, os, sys
from collections import defaultdict
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
KEYS = ("deform_region_fwd_kernel", "cpb_region_bwd_kernel", "deform_attn_bwd_dq_kernel", "deform_attn_bwd_dkv_kernel", "deform_attn_fwd_kernel", "cpb_bwd_kernel",
        "region_corners_kernel", "region_classify0_kernel", "region_classify1_kernel", "region_subcorners_kernel", "region_cand_kernel", "region_rank_kernel")
vals = defaultdict(lambda: defaultdict(list))
for p in "abcfw":
    for f in glob.glob(os.path.join(ROOT, "gpurun_out", "r5", f"pmc_{p}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = next((x for x in KEYS if x in r["Kernel_Name"]), None)
            if k:
                vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if r["Counter_Name"] in ("GRBM_GUI_ACTIVE",):
                    vals[k]["_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
m = lambda v: sum(v) / len(v) if v else float("nan")
lines = []
for k in KEYS:
    c = vals.get(k)
    if not c:
        continue
    g = lambda n: m(c.get(n, []))
    ns = g("_ns")
    wc = g("SQ_WAVE_CYCLES")
    clk = g("GRBM_GUI_ACTIVE") / 8 / ns if ns == ns else float("nan")
    lines.append(f"{k}: {ns / 1e3:.0f} us under the counters, clock {clk:.2f} GHz; per launch VALU {g('SQ_INSTS_VALU'):.3e} SALU {g('SQ_INSTS_SALU'):.3e} "
                 f"MFMA {g('SQ_INSTS_MFMA'):.3e} LDS {g('SQ_INSTS_LDS'):.3e} VMEM rd {g('SQ_INSTS_VMEM_RD'):.3e} wr {g('SQ_INSTS_VMEM_WR'):.3e} SMEM {g('SQ_INSTS_SMEM'):.3e} waves {g('SQ_WAVES'):.0f}")
    lines.append(f"    of wave cycles: ACTIVE_INST_ANY {g('SQ_ACTIVE_INST_ANY') / wc:.3f} WAIT_INST_ANY {g('SQ_WAIT_INST_ANY') / wc:.3f} WAIT_ANY {g('SQ_WAIT_ANY') / wc:.3f} "
                 f"ACTIVE_INST_VALU {g('SQ_ACTIVE_INST_VALU') / wc:.3f}; MFMA busy / SIMD busy {g('SQ_VALU_MFMA_BUSY_CYCLES') / (4 * g('SQ_BUSY_CYCLES')) if g('SQ_BUSY_CYCLES') else float('nan'):.3f}; "
                 f"LDS idx active {g('SQ_LDS_IDX_ACTIVE'):.3e} bank conflict {g('SQ_LDS_BANK_CONFLICT'):.3e}; L2 hit {g('TCC_HIT_sum') / max(g('TCC_HIT_sum') + g('TCC_MISS_sum'), 1):.3f} "
                 f"(hits {g('TCC_HIT_sum'):.3e}); HBM fetch {g('FETCH_SIZE') * 1024 * 2 / 1e9:.3f} GB (x2 gfx950) write {g('WRITE_SIZE') * 1024 / 1e9:.3f} GB")
out = "\n".join(lines)
print(out)
os.makedirs(os.path.join(ROOT, "gpurun_out", "r5"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "r5", f"{tag}_pmc_summary.txt"), "w").write(out + "\n")
# HBM traffic per launch of the four score-shaped kernels -> <tag>_hbm_traffic.json (the file bench.py replays when it cannot collect live)
import json
B, H, N, J = 8, 8, 10000, 625
pairs = B * H * N * J
ALG = {"deform_region_fwd_kernel": pairs * 6 + B * H * N * 516 + B * H * J * 520, "deform_attn_bwd_dq_kernel": pairs * 8 + 3 * B * N * 512 * 4 + B * H * N * 8,
       "deform_attn_bwd_dkv_kernel": pairs * 8 + 2 * B * N * 512 * 4, "cpb_region_bwd_kernel": pairs * 6}
tr = {"_comment": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, tests/tools/gpu_r5_pmc.sh) of `python bench.py --steps 3 --warmup 2 "
                  "--no-cpu-baseline --no-nystrom --no-traffic --no-deform16 --no-dp-overhead` at B = 8 bags of 10000 x 512 per launch; counters are KB per "
                  "dispatch, mean over the dispatches.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-byte requests at 64 bytes -> x 2, "
                  "calibrated for wide coalesced streaming reads; the 2-byte table gathers of the region forward are narrow accesses for which the guide gives "
                  "no calibration - its corrected figure is an upper bound.  WRITE_SIZE is exact for streaming stores.",
      "bags_per_launch": B, "kernels": {}}
for k, alg in ALG.items():
    c = vals.get(k)
    if c and c.get("FETCH_SIZE") and c.get("WRITE_SIZE"):
        fe, wr = m(c["FETCH_SIZE"]), m(c["WRITE_SIZE"])
        tr["kernels"][k] = {"fetch_kb_raw": fe, "write_kb": wr, "hbm_bytes_per_launch": (2 * fe + wr) * 1024, "algorithmic_bytes_per_launch": float(alg)}
if tr["kernels"]:
    json.dump(tr, open(os.path.join(ROOT, "gpurun_out", "r5", f"{tag}_hbm_traffic.json"), "w"), indent=0)
    print("step traffic, four kernels: %.2f GB by the counters, %.2f GB algorithmic" % (sum(v["hbm_bytes_per_launch"] for v in tr["kernels"].values()) / 1e9,
                                                                                         sum(v["algorithmic_bytes_per_launch"] for v in tr["kernels"].values()) / 1e9))

"""Turns the rocprofv3 passes of tests/tools/gpu_bench_prof.sh (gpurun_out/{prof,pmc1..4}) into the committed summaries under profiles/:
  profiles/<tag>_kernel_stats.csv     the --stats table of `python bench.py --steps 3 --warmup 1`
  profiles/<tag>_hbm_traffic.json     FETCH_SIZE / WRITE_SIZE per launch of the dominant kernels (separate passes), gfx950-corrected
  profiles/<tag>_pmc_notes.md         instruction mix and wait shares per kernel (SQ counters)
Usage: python tests/tools/summarize_pmc.py r02 [bags_per_launch] [pmc dir prefix = pmc] [kernel-trace dir = prof]
(the 16-bit compute mode's passes: python tests/tools/summarize_pmc.py r04_deform16 8 pmc16_ prof16)"""
import csv, glob, json, os, shutil, sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
bags = int(sys.argv[2]) if len(sys.argv) > 2 else 8
PMC = sys.argv[3] if len(sys.argv) > 3 else "pmc"
PROF = sys.argv[4] if len(sys.argv) > 4 else "prof"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
G = os.path.join(ROOT, "gpurun_out")


def newest(pattern):
    fs = glob.glob(os.path.join(G, pattern))
    return max(fs, key=os.path.getmtime) if fs else None


def short(name):
    if "deform16_fwd_kernel" in name and ("Li96E" in name or ", 96>" in name):
        return "deform16_fwd_kernel (table)"
    for key in ("cpb_table_grid_bwd_kernel", "cpb_table_bwd_kernel", "cpb16_bwd_kernel", "deform16_fwd_kernel", "deform16_bwd_dq_kernel", "deform16_bwd_dkv_kernel", "cpb_bwd_kernel", "deform_attn_fwd_kernel", "deform_attn_bwd_dq_kernel", "deform_attn_bwd_dkv_kernel", "gemm_f32_fast_kernel",
                "gemm_bf3_kernel", "offsets_bwd", "offsets_fwd", "layernorm", "colsum", "attn16_fwd", "attn16_bwd_dq", "attn16_bwd_dkv"):
        if key in name:
            return key
    return None


def counters(path):
    """kernel -> counter -> [values per dispatch]"""
    out = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            if k:
                out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    out[k]["_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return out


KERNELS = ("deform16_fwd_kernel (table)", "cpb_table_grid_bwd_kernel", "cpb_table_bwd_kernel", "cpb16_bwd_kernel", "deform16_fwd_kernel", "deform16_bwd_dq_kernel", "deform16_bwd_dkv_kernel", "cpb_bwd_kernel", "deform_attn_fwd_kernel",
           "deform_attn_bwd_dq_kernel", "deform_attn_bwd_dkv_kernel")
ks = newest(PROF + "/runc/*_kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(ROOT, "profiles", f"{tag}_bench_b{bags}_kernel_stats.csv"))
    print("copied", ks)
mean = lambda v: sum(v) / len(v) if v else float("nan")
f3, f4 = newest(PMC + "3/runc/*_counter_collection.csv"), newest(PMC + "4/runc/*_counter_collection.csv")
traffic = {"_comment": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, tests/tools/gpu_bench_prof.sh) of `python bench.py --steps 3 --warmup 1 "
                       f"--no-cpu-baseline --no-nystrom` at B = {bags} bags of 10000 x 512 per launch; counter values are KB per dispatch, mean over the "
                       "dispatches of the run.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE tallies 128-byte requests at 64 bytes -> x 2 for "
                       "wide coalesced streaming reads (all kernels below read 16 B per lane); WRITE_SIZE is exact for 16-byte streaming stores and float atomics.",
           "bags_per_launch": bags, "kernels": {}}
if f3 and f4:
    c3, c4 = counters(f3), counters(f4)
    for k in [k for k in KERNELS if c3[k]["FETCH_SIZE"]]:
        fe, wr = mean(c3[k]["FETCH_SIZE"]), mean(c4[k]["WRITE_SIZE"])
        name = {"cpb_bwd_kernel": "cpb_bwd_kernel<2>", "deform_attn_fwd_kernel": "deform_attn_fwd_kernel<2>"}.get(k, k)
        traffic["kernels"][name] = {"FETCH_SIZE_KB": fe, "WRITE_SIZE_KB": wr, "hbm_bytes_per_launch": (2 * fe + wr) * 1024, "dispatches": len(c3[k]["FETCH_SIZE"])}
    with open(os.path.join(ROOT, "profiles", f"{tag}_hbm_traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
    print(json.dumps(traffic["kernels"], indent=1))
f1, f2 = newest(PMC + "1/runc/*_counter_collection.csv"), newest(PMC + "2/runc/*_counter_collection.csv")
if f1 and f2:
    c1, c2 = counters(f1), counters(f2)
    lines = [f"# PMC notes ({tag}): `python bench.py --steps 3 --warmup 1`, B = {bags} bags of 10 000 x 512, means per dispatch", "",
             "SQ_* cycle counters are in quad-cycles summed over waves (MI355X_MICROARCH.md); instruction counters are wave-instructions.", "",
             "| kernel | VALU insts | MFMA insts | VALU / MFMA | LDS insts | LDS bank-conflict / LDS active | WAIT_ANY | WAIT_INST_ANY | ACTIVE_INST_ANY (shares of WAVE_CYCLES) | MFMA busy cycles | MFMA busy / SIMD cycles | co-exec / MFMA busy | clock GHz |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for k in [k for k in KERNELS if c1[k]["SQ_WAVE_CYCLES"]]:
        a, b = c1[k], c2[k]
        wc = mean(a["SQ_WAVE_CYCLES"])
        valu, mf = mean(a["SQ_INSTS_VALU"]), mean(b["SQ_INSTS_MFMA"])
        lines.append(f"| {k} | {valu:.3e} | {mf:.3e} | {(valu / mf) if mf else float('inf'):.1f} | {mean(b['SQ_INSTS_LDS']):.3e} | {mean(b['SQ_LDS_BANK_CONFLICT']) / max(mean(b['SQ_LDS_IDX_ACTIVE']), 1):.3f} | "
                     f"{mean(a['SQ_WAIT_ANY']) / wc:.2f} | {mean(a['SQ_WAIT_INST_ANY']) / wc:.2f} | {mean(a['SQ_ACTIVE_INST_ANY']) / wc:.2f} | {mean(a['SQ_VALU_MFMA_BUSY_CYCLES']):.3e} | "
                     f"{mean(a['SQ_VALU_MFMA_BUSY_CYCLES']) / 1024 / (mean(a['GRBM_GUI_ACTIVE']) / 8):.3f} | "
                     f"{(mean(a['SQ_VALU_MFMA_COEXEC_CYCLES']) / mean(a['SQ_VALU_MFMA_BUSY_CYCLES'])) if (a['SQ_VALU_MFMA_COEXEC_CYCLES'] and mean(a['SQ_VALU_MFMA_BUSY_CYCLES'])) else float('nan'):.3f} | "
                     f"{mean(a['GRBM_GUI_ACTIVE']) / 8 / (mean(a['_ns']) if a['_ns'] else float('nan')):.2f} |")
    open(os.path.join(ROOT, "profiles", f"{tag}_pmc_notes.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))

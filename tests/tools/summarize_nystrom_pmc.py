"""Turns the two rocprofv3 --pmc passes of tests/tools/gpu_nystrom_pmc.sh (gpurun_out/npmc1, npmc2) into profiles/<tag>_nystrom16_pmc.txt:
per kernel of the Nystrom block's bf16 step (means per dispatch) the duration, instruction counts and the wait shares.
Usage: python tests/tools/summarize_nystrom_pmc.py r03"""
import csv, glob, os, re, subprocess, sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
G = os.path.join(ROOT, "gpurun_out")


def demangle(n):
    if n.startswith("_Z"):
        n = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip() or n
    m = re.match(r"_ZN12_GLOBAL__N_1\d+(\w+?_kernel)I(DF16b|DF16_)(f|DF16b|DF16_)(f|DF16b|DF16_)E", n)
    if m:                                   # c++filt does not know the bf16 / _Float16 manglings
        t = {"DF16b": "bf16", "DF16_": "fp16", "f": "float"}
        return f"{m.group(1)}<{t[m.group(2)]}, {t[m.group(3)]}, {t[m.group(4)]}>"
    m = re.match(r"_ZN12_GLOBAL__N_1\d+(\w+?_kernel)E", n)
    if m:
        return m.group(1)
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*", "", n)


def counters(path):
    out = defaultdict(lambda: defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            k = demangle(r["Kernel_Name"])
            if not any(s in k for s in ("attn16", "gemm_b16", "chain_", "resconv_b16", "segment_mean")):
                continue
            out[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] in ("GRBM_GUI_ACTIVE", "SQ_INSTS_MFMA"):
                out[k]["_ns:" + r["Counter_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return out


def newest(p):
    fs = glob.glob(os.path.join(G, p))
    return max(fs, key=os.path.getmtime) if fs else None


c1, c2 = counters(newest("npmc1/**/*_counter_collection.csv") or newest("npmc1/*/*_counter_collection.csv")), counters(newest("npmc2/*/*_counter_collection.csv"))
mean = lambda v: sum(v) / len(v) if v else float("nan")
def maybe(pat):
    f = newest(pat)
    return counters(f) if f else {}


c3, c4 = maybe("npmc3/*/*_counter_collection.csv"), maybe("npmc4/*/*_counter_collection.csv")


def hbm(k, ns):
    fe, wr = c3.get(k, {}).get("FETCH_SIZE"), c4.get(k, {}).get("WRITE_SIZE")
    if not fe or not wr:
        return "n/a\tn/a"
    mb = (2 * mean(fe) + mean(wr)) * 1024 / 1e6
    return f"{mb:.0f}\t{mb * 1e6 / (ns * 1e-9) / 1e12:.2f}"



lines = ["# rocprofv3 --pmc (two passes, tests/tools/gpu_nystrom_pmc.sh) of `python tests/tools/bench_nystrom.py --n 10000 --bags 4 --dtype bfloat16 --steps 3`:",
         "# kernels of the Nystrom block's bf16 step, means per dispatch; wait / active counters as shares of SQ_WAVE_CYCLES; MFMA busy as a share of",
         "# the SIMD cycles of the launch (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs, as in tests/summarize_pmc.py).  attn16 template arguments: <pipe type, query-side storage, key-side storage>:",
         "# <bf16, bf16, float> is the [n', m] side (10 240 queries x 256 landmark keys), <bf16, float, bf16> the [m, n'] side (key-split).",
         "# HBM MB per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB (gfx950: FETCH_SIZE counts 128-byte requests at 64 bytes), separate passes; TB/s against the us column.",
         "kernel\tdispatches\tus\tVALU insts\tMFMA insts\tVALU/MFMA\tLDS insts\tACTIVE_INST_ANY\tWAIT_INST_ANY\tWAIT_ANY\tMFMA busy / SIMD cycles\tHBM MB / launch\tTB/s"]
for k in sorted(c1, key=lambda k: -mean(c1[k]["_ns:GRBM_GUI_ACTIVE"]) * len(c1[k]["_ns:GRBM_GUI_ACTIVE"])):
    a, b = c1[k], c2.get(k, {})
    wc = mean(a["SQ_WAVE_CYCLES"])
    valu, mf = mean(a["SQ_INSTS_VALU"]), mean(b.get("SQ_INSTS_MFMA", []))
    lines.append(f"{k}\t{len(a['_ns:GRBM_GUI_ACTIVE'])}\t{mean(a['_ns:GRBM_GUI_ACTIVE']) / 1e3:.1f}\t{valu:.3e}\t{mf:.3e}\t{valu / mf if mf else float('nan'):.1f}\t"
                 f"{mean(b.get('SQ_INSTS_LDS', [])):.3e}\t{mean(a['SQ_ACTIVE_INST_ANY']) / wc:.2f}\t{mean(a['SQ_WAIT_INST_ANY']) / wc:.2f}\t"
                 f"{mean(a['SQ_WAIT_ANY']) / wc:.2f}\t{mean(a['SQ_VALU_MFMA_BUSY_CYCLES']) / 1024 / (mean(a['GRBM_GUI_ACTIVE']) / 8):.3f}\t" + hbm(k, mean(a['_ns:GRBM_GUI_ACTIVE'])))
out = os.path.join(ROOT, "profiles", f"{tag}_nystrom16_pmc.txt")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

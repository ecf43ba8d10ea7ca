#!/bin/bash
# Matrix / vector co-execution counters of the two position-bias kernels: separate --pmc passes of the bench step.
set -u
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
grep -i -E "MFMA|COEXEC|BUSY_CY|INST_CYCLES|ACTIVE_INST" gpurun_out/counters_list.txt | cut -c1-160 | sort -u | head -60
A="--steps 3 --warmup 1 --no-cpu-baseline --no-nystrom --no-traffic"
run() { local name=$1 t=$2; shift 2; echo "=== $name"; timeout -k 10 "$t" "$@" > "gpurun_out/$name.log" 2>&1; local rc=$?; echo "rc=$rc"; tail -n 2 "gpurun_out/$name.log" | cut -c1-600
  if [ $rc -eq 124 ] || [ $rc -ge 128 ]; then echo "!!! $name died (rc $rc): stopping"; exit $rc; fi; return $rc; }
run coexec1 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/coexec1 -- python bench.py $A || exit 1
run coexec2 500 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d gpurun_out/coexec2 -- python bench.py $A
python - <<'PY'
import csv, glob, collections
for d in ("coexec1", "coexec2"):
    fs = glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(max(fs, key=lambda p: __import__('os').path.getmtime(p)))):
        for k in ("cpb_bwd_kernel", "deform_attn_fwd_kernel", "bwd_dq_kernel", "bwd_dkv_kernel"):
            if k in r["Kernel_Name"]:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                acc[k]["_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, v in acc.items():
        print(d, k, {c: f"{sum(x)/len(x):.4e}" for c, x in v.items()})
PY
echo done

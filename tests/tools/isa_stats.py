"""Instruction statistics of the kernels of one HIP source: compiles csrc/<file> to gfx950 assembly with the flags of _build.py
and prints, per kernel, VGPR count, scratch, and the instruction mix of the whole body and of its hottest loop (the innermost
loop with the most MFMAs).  Usage: python tests/tools/isa_stats.py deform_attn.hip [-DSMML_SPLIT_TERMS=3 ...]"""
import importlib, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
b = importlib.import_module("subspace-multimodal-learning_amd._build")
src = sys.argv[1]
extra = sys.argv[2:]
out = os.path.join(tempfile.mkdtemp(), src[:-4] + ".s")
cmd = ["/opt/rocm/bin/hipcc", *b.FLAGS, *b.EXTRA_FLAGS.get(src, []), *extra, "-I", b.CSRC, "-S", "--cuda-device-only", os.path.join(b.CSRC, src), "-o", out]
subprocess.run(cmd, check=True)
txt = open(out).read()
print("asm:", out)
for m in re.finditer(r"^(_Z\w+):\n(.*?)\n\s*\.end_amdhsa_kernel", txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    dem = subprocess.run(["/usr/bin/c++filt", name], capture_output=True, text=True).stdout.strip()
    lines = [l.strip() for l in body.split("\n")]
    ins = [l for l in lines if l and not l.startswith((";", ".", "/")) and not l.endswith(":")]
    def mix(seq):
        c = {"mfma": 0, "valu": 0, "salu": 0, "ds": 0, "vmem": 0, "trans": 0, "pk": 0, "cvt": 0, "cmp": 0, "cndmask": 0, "waitcnt": 0, "nop": 0}
        for l in seq:
            op = l.split()[0]
            if op.startswith("v_mfma"): c["mfma"] += 1
            elif op.startswith("v_"):
                c["valu"] += 1
                if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")): c["trans"] += 1
                if op.startswith("v_pk_"): c["pk"] += 1
                if op.startswith("v_cvt") or op.startswith("v_fma_mix"): c["cvt"] += 1
                if op.startswith("v_cmp"): c["cmp"] += 1
                if op.startswith("v_cndmask"): c["cndmask"] += 1
            elif op.startswith("s_waitcnt"): c["waitcnt"] += 1
            elif op.startswith("s_nop"): c["nop"] += 1
            elif op.startswith("s_"): c["salu"] += 1
            elif op.startswith("ds_"): c["ds"] += 1
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")): c["vmem"] += 1
        return c
    # loops: a backward branch to label L between label position and branch position
    labels = {l[:-1]: i for i, l in enumerate(lines) if l.endswith(":")}
    best = None
    for i, l in enumerate(lines):
        mm = re.match(r"s_cbranch_\w+\s+(\S+)", l) or re.match(r"s_branch\s+(\S+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            seq = [x for x in lines[labels[mm.group(1)]:i + 1] if x and not x.startswith((";", ".", "/")) and not x.endswith(":")]
            c = mix(seq)
            key = (c["mfma"], -len(seq))
            if c["mfma"] and (best is None or len(seq) < best[2] or False):
                # innermost = shortest loop that still contains MFMAs
                best = (mm.group(1), c, len(seq))
    vg = re.search(r"\.amdhsa_next_free_vgpr (\d+)", txt[m.end() - 4000:m.end()] if False else body + txt[m.end():m.end() + 10])
    print(f"\n{dem[:110]}")
    print("  body :", mix(ins), "instrs", len(ins))
    if best:
        print(f"  loop {best[0]} ({best[2]} instrs):", best[1])
for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n){0,30}?\s+\.vgpr_count:\s+(\d+)", txt):
    pass
for blk in re.finditer(r"- \.agpr_count:\s+(\d+)(.*?)\.wavefront_size", txt, re.S):
    t = blk.group(0)
    g = lambda k: (re.search(k + r":\s+(\S+)", t) or [None, "?"])[1]
    print(g(r"\.name")[:60], "vgpr", g(r"\.vgpr_count"), "agpr", g(r"\.agpr_count"), "sgpr", g(r"\.sgpr_count"), "spill", g(r"\.vgpr_spill_count"), "scratch", g(r"\.private_segment_fixed_size"), "lds", g(r"\.group_segment_fixed_size"))

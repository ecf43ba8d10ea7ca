"""Instruction mix of the loops that contain MFMAs, per kernel of one HIP source (gfx950 assembly, the flags of _build.py).
Usage: python tests/tools/isa_loops.py deform_attn.hip [kernel-name-substring] [-DSMML_... flags]"""
import collections, importlib, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
b = importlib.import_module("subspace-multimodal-learning_amd._build")
src = sys.argv[1]
rest = sys.argv[2:]
want = [a for a in rest if not a.startswith("-")]
extra = [a for a in rest if a.startswith("-")]
out = os.path.join(tempfile.mkdtemp(), src[:-4] + ".s")
subprocess.run(["/opt/rocm/bin/hipcc", *b.FLAGS, *b.EXTRA_FLAGS.get(src, []), *extra, "-I", b.CSRC, "-S", "--cuda-device-only",
                os.path.join(b.CSRC, src), "-o", out], check=True, stderr=subprocess.DEVNULL)
lines = open(out).read().split("\n")
starts = [(i, l.split(":")[0]) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
print("asm:", out)
for n, (i0, name) in enumerate(starts):
    i1 = starts[n + 1][0] if n + 1 < len(starts) else len(lines)
    dem = subprocess.run(["/usr/bin/c++filt", name], capture_output=True, text=True).stdout.strip()
    if want and not any(w in dem for w in want):
        continue
    body = [l.split(";")[0].strip() for l in lines[i0:i1]]
    labels = {l[:-1]: i for i, l in enumerate(body) if re.match(r"^\.?\w+:$", l)}
    loops = []
    for i, l in enumerate(body):
        m = re.match(r"s_cbranch_\w+\s+(\S+)", l) or re.match(r"s_branch\s+(\S+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            seq = [x for x in body[labels[m.group(1)]:i + 1] if x and not x.startswith((";", ".", "/")) and not x.endswith(":")]
            nm = sum(1 for x in seq if x.startswith("v_mfma"))
            if nm:
                loops.append((len(seq), nm, m.group(1), seq))
    print("\n" + dem[:120])
    shown = 0
    for n_, nm, lab, seq in sorted(loops):
        # nested loops contain their inner loops' instructions: list the innermost ones only (loops whose label range holds no other
        # listed loop), at most four per kernel
        if any(o[0] < n_ and all(x in seq for x in o[3][:3]) and labels[o[2]] > labels[lab] for o in loops if o[2] != lab):
            continue
        shown += 1
        if shown > 4:
            break
        c = collections.Counter(x.split()[0] for x in seq)
        valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
        print(f"  loop {lab}: {n_} instrs, mfma {nm}, valu {valu}, salu {sum(v for k, v in c.items() if k.startswith('s_'))}, "
              f"ds {sum(v for k, v in c.items() if k.startswith('ds_'))}, vmem {sum(v for k, v in c.items() if k.startswith(('global', 'buffer', 'scratch')))}")
        print("    ", ", ".join(f"{k} {v}" for k, v in sorted(c.items(), key=lambda kv: -kv[1])[:45]))

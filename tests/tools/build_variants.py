"""Builds variants of the HIP library with different -D knobs into lib/variants/<name>.so (measurement only;
select one at run time with SMML_LIB=<path>).  Usage: python tests/tools/build_variants.py name=-DFOO=1,-DBAR=2 ..."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, "subspace-multimodal-learning_amd")
sys.path.insert(0, ROOT)
import importlib
bld = importlib.import_module("subspace-multimodal-learning_amd._build")


def build_variant(name, defs, files=("deform_attn.hip", "deform_attn16.hip")):
    out_dir = os.path.join(PKG, "lib", "variants")
    obj_dir = os.path.join(PKG, "build", "variants", name)
    os.makedirs(out_dir, exist_ok=True); os.makedirs(obj_dir, exist_ok=True)
    objs = []
    for s in bld.sources():
        src = os.path.join(bld.CSRC, s)
        if s in files:
            obj = os.path.join(obj_dir, s[:-4] + ".o")
            cmd = ["/opt/rocm/bin/hipcc", *bld.FLAGS, *bld.EXTRA_FLAGS.get(s, []), *defs, "-I", bld.CSRC, "-c", src, "-o", obj]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode:
                raise RuntimeError(r.stderr)
        else:
            obj = os.path.join(PKG, "build", s[:-4] + ".o")
        objs.append(obj)
    lib = os.path.join(out_dir, name + ".so")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", f"--offload-arch={bld.ARCH}", "-fno-gpu-rdc", *objs, "-o", lib],
                       capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError(r.stderr)
    print("built", lib)


if __name__ == "__main__":
    bld.build()
    jobs = []
    for a in sys.argv[1:]:
        name, _, d = a.partition("=")
        jobs.append((name, [x for x in d.split(",") if x]))
    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(lambda j: build_variant(*j), jobs))

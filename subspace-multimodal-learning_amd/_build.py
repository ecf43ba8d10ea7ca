"""Builds lib/libsmml_hip.so (the C-ABI of include/smml.h) from csrc/*.hip with hipcc for gfx950.

hipcc cross-compiles without a GPU; the .so stays in-tree (git-ignored) so it travels with the
repository snapshot to the GPU box."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libsmml_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
# per-file additions.  deform_attn.hip: the ReLUs on matrix-core results (fmaxf(acc, 0)) otherwise cost a second
# v_max per element that only quiets signalling NaNs (IEEE mode); no NaN is ever produced or consumed there.
# -fno-slp-vectorize: packed fp32 forms are no faster than two plain fp32 instructions on gfx950 except for FMA, and the
# SLP vectoriser turns modifier forms (x + |x|) into packed ones that need extra instructions; the kernels write float2
# arithmetic explicitly where it pays (tests/microbench/valu_mix_probe.hip)
_DEFORM_FLAGS = ["-fno-honor-nans", "-mno-amdgpu-ieee", "-fno-slp-vectorize"]
EXTRA_FLAGS = {"deform_attn.hip": _DEFORM_FLAGS, "deform_attn16.hip": _DEFORM_FLAGS}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(PKG, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    jobs = []
    for s in sources():
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s[:-4] + ".o")
        if force or _stale(obj, [src] + headers):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(os.path.basename(src), []), *os.environ.get("SMML_HIPCC_FLAGS", "").split(),
               "-I", CSRC, "-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr}")
        if verbose:
            print(f"[smml build] compiled {os.path.basename(src)}", file=sys.stderr)
        return obj

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(objdir, s[:-4] + ".o") for s in sources()]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        if verbose:
            print(f"[smml build] linked {LIB}", file=sys.stderr)
    return LIB


HOST_SAN_LIB = os.path.join(LIBDIR, "libsmml_host_san.so")
SAN_RUNTIME = "/opt/rocm/lib/llvm/lib/clang/22/lib/linux/libclang_rt.asan-x86_64.so"


def build_host_sanitized(force: bool = False, verbose: bool = True) -> str:
    """AddressSanitizer + UndefinedBehaviorSanitizer build of the HOST side of every csrc/*.hip - argument validation, workspace sizing,
    launch-geometry arithmetic of the C-ABI entry points - into lib/libsmml_host_san.so: the sanitizer flags go to the host pass only
    (-Xarch_host; the device code is compiled as usual so that the library links and loads).  CPU only: GPU sanitizers are not available
    on this pool.  tests/test_host_sanitizers.py loads it in a child process with the ASan runtime preloaded and
    walks every entry point with null / boundary arguments."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(PKG, "build", "host_san")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    san = ["-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g"]
    jobs = []
    for s in sources():
        src, obj = os.path.join(CSRC, s), os.path.join(objdir, s[:-4] + ".o")
        if force or _stale(obj, [src] + headers):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        r = subprocess.run([hipcc, "-O1", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-fno-gpu-rdc", *san, "-I", CSRC, "-c", src, "-o", obj],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"sanitized host build failed on {src}:\n{r.stderr}")

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(objdir, s[:-4] + ".o") for s in sources()]
    if force or jobs or _stale(HOST_SAN_LIB, objs):
        r = subprocess.run([hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-fno-gpu-rdc", "-Xarch_host", "-fsanitize=address,undefined", *objs,
                            "-o", HOST_SAN_LIB], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"sanitized host link failed:\n{r.stderr}")
        if verbose:
            print(f"[smml build] linked {HOST_SAN_LIB}", file=sys.stderr)
    return HOST_SAN_LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    if "--host-san" in sys.argv:
        build_host_sanitized(force="--force" in sys.argv)

"""CPU: AddressSanitizer + UndefinedBehaviorSanitizer over the HOST side of the C-ABI.

`_build.build_host_sanitized()` compiles every csrc/*.hip with the sanitizers on the HOST pass only (hipcc -Xarch_host -fsanitize=address,undefined) into
lib/libsmml_host_san.so.  A child process with the ASan runtime preloaded loads it through ctypes (no torch, no GPU) and walks EVERY entry
point include/smml.h declares:
  * all-null pointers and zero sizes: a negative return code and a message, never a crash or a sanitizer report;
  * the workspace / size functions on boundary shapes (one query, one key, the 50 176-row bag of config 5, products of the dimensions
    near and beyond 2^31): UBSan aborts on any signed overflow in the launch-geometry or sizing arithmetic;
  * valid-looking dimensions with null pointers (validation must come before any dereference or launch).
GPU sanitizers are not available on this pool; device code is covered by the parity tests."""
import os
import re
import subprocess
import sys
import textwrap

import pytest

from helpers import smml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
bld = sys.modules[smml.__name__ + "._build"] if smml.__name__ + "._build" in sys.modules else __import__("importlib").import_module(smml.__name__ + "._build")

CHILD = r'''
import ctypes as C, re, sys
lib_path, header = sys.argv[1], sys.argv[2]
txt = re.sub(r"/\*.*?\*/", "", open(header).read(), flags=re.S)
txt = re.sub(r"typedef struct SmmlDeformOpts \{.*?\} SmmlDeformOpts;", "", txt, flags=re.S)
L = C.CDLL(lib_path)
L.smml_last_error.restype = C.c_char_p
decls = re.findall(r"\b(int|size_t|void|const char\*)\s+(smml_[a-z0-9_]+)\s*\(([^;]*?)\)\s*;", txt, flags=re.S)
assert len(decls) >= 60, len(decls)

def ctype(p):
    p = " ".join(p.split())
    if "*" in p: return C.c_void_p
    if p.startswith("unsigned long long"): return C.c_ulonglong
    if p.startswith("long long"): return C.c_longlong
    if p.startswith("size_t"): return C.c_size_t
    if p.startswith("float"): return C.c_float
    if p.startswith(("int", "unsigned")): return C.c_int
    raise SystemExit("unparsed parameter: " + p)

funcs = {}
for ret, name, body in decls:
    body = body.strip()
    args = [] if body in ("", "void") else [ctype(a) for a in body.split(",")]
    f = getattr(L, name)
    f.argtypes = args
    f.restype = {"int": C.c_int, "size_t": C.c_size_t, "void": None, "const char*": C.c_char_p}[ret]
    funcs[name] = (ret, args, f)

def zero(t):
    return None if t is C.c_void_p else t(0)

n_err = 0
# 1. everything null / zero
for name, (ret, args, f) in sorted(funcs.items()):
    if name in ("smml_event_destroy",):
        continue
    r = f(*[zero(t) for t in args])
    if ret == "int" and args and any(t is C.c_void_p for t in args):
        assert r <= 0, (name, r)
        n_err += r < 0
# 2. sizing functions on boundary shapes (UBSan: -fno-sanitize-recover aborts the child on signed overflow)
shapes4 = [(1, 1, 1, 1), (8, 10000, 625, 8), (1, 50176, 3136, 8), (64, 50176, 768, 8), (65535, 1, 1, 1), (1, 2147483647, 1, 1), (8, 1 << 20, 768, 8),
           (0, 0, 0, 0), (-1, 5, 5, 8), (1, 1, 2147483647, 8)]
for name in ("smml_deform_attn_bwd_workspace_bytes", "smml_deform_attn_region_bwd_workspace_bytes"):
    for s in shapes4:
        funcs[name][2](*s)
for s in [(1, 1, 1), (64, 50176, 256), (64, 256, 50176), (1 << 16, 1 << 15, 256), (0, 0, 0)]:
    for name in ("smml_attn16_fwd_workspace_bytes", "smml_attn16_bwd_workspace_bytes"):
        funcs[name][2](*s)
for s in shapes4:
    funcs["smml_deform_attn_table_bwd_workspace_bytes"][2](*s, 2)
    funcs["smml_deform_attn_table_bwd_workspace_bytes"][2](*s, 1)
for s in [(8, 100, 100, 8, 64, 6, 4, 2), (1, 1, 1, 1, 4, 6, 4, 1), (65535, 4096, 4096, 8, 64, 6, 4, 2), (8, 224, 224, 8, 64, 6, 4, 2)]:
    funcs["smml_offsets_bwd_workspace_bytes"][2](*s)
for n in (1, 31, 32, 10000, 50176, 2147483520):
    funcs["smml_deform_attn_nst"][2](n)
for s in [(50, 6, 4), (2501, 6, 4), (1, 6, 4), (0, 6, 4), (2147483647, 6, 4)]:
    funcs["smml_offsets_out_len"][2](*s)
assert funcs["smml_cpb_regions_bytes"][2]() > (1 << 20)
# 3. plausible dimensions, null pointers: validation precedes every dereference and launch
ONE = {C.c_int: 8, C.c_longlong: 8, C.c_size_t: 1 << 20, C.c_float: 0.5, C.c_ulonglong: 3}
for name, (ret, args, f) in sorted(funcs.items()):
    if ret != "int" or not any(t is C.c_void_p for t in args) or name.startswith("smml_event_"):
        continue                                     # (destroying a null event handle is a no-op by contract)
    r = f(*[None if t is C.c_void_p else t(ONE[t]) for t in args])
    assert r < 0, (name, r)
    assert L.smml_last_error(), name
assert L.smml_abi_version() == 2
print("OK", len(funcs), "entry points,", n_err, "null-argument errors reported")
'''


@pytest.mark.timeout(600)
def test_host_layer_under_asan_and_ubsan(tmp_path):
    if not os.path.exists(bld.SAN_RUNTIME):
        pytest.skip("no ASan runtime in this image")
    lib = bld.build_host_sanitized(verbose=False)
    script = tmp_path / "walk.py"
    script.write_text(textwrap.dedent(CHILD))
    env = dict(os.environ, LD_PRELOAD=bld.SAN_RUNTIME, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, str(script), lib, os.path.join(ROOT, "include", "smml.h")], capture_output=True, text=True, env=env, timeout=300)
    report = (r.stdout + r.stderr)
    assert r.returncode == 0 and "OK" in r.stdout, report[-4000:]
    assert not re.search(r"runtime error|AddressSanitizer|UndefinedBehaviorSanitizer", report), report[-4000:]

"""Builds lib4dflow_hip.so (gfx950 only) in-tree with hipcc.  No GPU is needed to build."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib4dflow_hip.so")
# Test build: the same sources with -DFDN_TEST_HOOKS, which adds the fdn_debug_* variant-forcing / ablation entry points
# (process-global switches).  Loaded only by tests/ and tools/ through _lib.test_build(); the product library has none.
LIB_TEST = os.path.join(HERE, "lib4dflow_hip_test.so")
SOURCES = ["api.hip", "conv64_mfma.hip", "conv64_wino.hip", "conv64_bf16.hip", "wgrad64_mfma.hip", "wgrad64_wino.hip", "wgrad64_bf16.hip", "small_convs.hip", "cin3_mfma.hip", "heads_mfma.hip", "elementwise.hip", "patch_gather.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-I", os.path.join(ROOT, "include"), "-I", CSRC]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stamp():
    h = hashlib.sha256()
    for name in sorted(n for n in os.listdir(CSRC) if n.endswith((".hip", ".h"))) + ["../../include/fdn.h"]:
        with open(os.path.join(CSRC, name), "rb") as f:
            h.update(name.encode()); h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build_library(force=False, verbose=False, test_hooks=False):
    """Compile every HIP translation unit for gfx950 and link the shared library.  Returns its path.
    test_hooks=True builds lib4dflow_hip_test.so (adds the fdn_debug_* entry points)."""
    lib = LIB_TEST if test_hooks else LIB
    extra = ["-DFDN_TEST_HOOKS"] if test_hooks else []
    stamp_file = lib + ".stamp"
    stamp = _stamp() + "".join(extra)
    if not force and os.path.exists(lib) and os.path.exists(stamp_file) and open(stamp_file).read() == stamp:
        return lib
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build", "test" if test_hooks else "product")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc, "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj] + FLAGS + extra
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(4, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    r = subprocess.run([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    with open(stamp_file, "w") as f:
        f.write(stamp)
    return lib


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
    print(build_library(force="--force" in sys.argv, verbose=True, test_hooks=True))

"""Builds lib4dflow_hip.so (gfx950 only) in-tree with hipcc.  No GPU is needed to build.

Every built library carries a stamp (<lib>.stamp = sha256 over csrc/*.hip, csrc/*.h, include/fdn.h and the compiler flags).
`ensure_built` -- called by `_lib.load()` before every dlopen -- compares it with the sources in the tree and rebuilds (or, with
FDN_NO_REBUILD=1, raises StaleLibraryError) when they differ, so a binary older than its sources is never executed.  Builds are
serialised by an exclusive file lock (csrc/../build/.lock): with several ranks per node (torch.distributed.run) the first rank to
take the lock compiles, the others block on it and then find the fresh stamp."""
import contextlib
import fcntl
import hashlib
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib4dflow_hip.so")
# Test build: the same sources with -DFDN_TEST_HOOKS, which adds the fdn_debug_* variant-forcing / ablation entry points
# (process-global switches).  Loaded only by tests/ and tools/ through _lib.test_build(); the product library has none.
LIB_TEST = os.path.join(HERE, "lib4dflow_hip_test.so")
SOURCES = ["api.hip", "conv64_mfma.hip", "conv64_wino.hip", "conv64_wino2d.hip", "conv64_bf16.hip", "wgrad64_mfma.hip", "wgrad64_wino.hip", "wgrad64_bf16.hip", "small_convs.hip", "cin3_mfma.hip", "conv1x1_mfma.hip", "heads_mfma.hip", "elementwise.hip", "patch_gather.hip"]
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
    # the flags enter with the checkout's own path replaced: the same sources must stamp alike wherever the tree is copied
    # (the GPU box runs a snapshot under another root; a path-dependent stamp would rebuild there on every call)
    h.update(" ".join(f.replace(ROOT, "<root>") for f in FLAGS).encode())
    return h.hexdigest()


class StaleLibraryError(RuntimeError):
    pass


def lib_path(test_hooks=False):
    return LIB_TEST if test_hooks else LIB


def source_stamp(test_hooks=False):
    """The stamp a library built from the sources currently in the tree would carry."""
    return _stamp() + ("-DFDN_TEST_HOOKS" if test_hooks else "")


def suite_stamp():
    """sha256 over everything a `pytest -m gpu` + smoke() run executes besides the kernels: the library stamp, the package's
    python, the GPU tests (+ conftest, the golden fixtures' loaders), the oracle and the entry points.  A tracked GPU test log
    (profiles/r*_gputest.txt, tools/gputest_round.sh) carries it, so a log older than the code it vouches for is detectable."""
    import glob
    h = hashlib.sha256(_stamp().encode())
    files = (glob.glob(os.path.join(HERE, "*.py")) + glob.glob(os.path.join(ROOT, "tests", "test_gpu_*.py")) +
             glob.glob(os.path.join(ROOT, "oracle", "*.py")) +
             [os.path.join(ROOT, "tests", n) for n in ("conftest.py", "test_tf_golden.py", "test_fft_downsampling.py")] +
             [os.path.join(ROOT, "__graft_entry__.py")])
    for path in sorted(files):
        if os.path.exists(path):
            with open(path, "rb") as f:
                h.update(os.path.relpath(path, ROOT).encode()); h.update(f.read())
    return h.hexdigest()


def built_stamp(test_hooks=False):
    """The stamp of the library in the tree, or None if the library or its stamp file is missing."""
    lib = lib_path(test_hooks)
    try:
        if os.path.exists(lib):
            with open(lib + ".stamp") as f:
                return f.read()
    except OSError:
        pass
    return None


def is_current(test_hooks=False):
    return built_stamp(test_hooks) == source_stamp(test_hooks)


@contextlib.contextmanager
def _build_lock():
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    with open(os.path.join(HERE, "build", ".lock"), "a+") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)


def ensure_built(test_hooks=False):
    """Return the path of a library whose stamp matches the sources: as is when current, else rebuilt under the build lock.
    FDN_NO_REBUILD=1 turns a stale or missing library into StaleLibraryError instead."""
    if is_current(test_hooks):
        return lib_path(test_hooks)
    if os.environ.get("FDN_NO_REBUILD", "0") not in ("", "0"):
        have = built_stamp(test_hooks)
        raise StaleLibraryError("%s is %s (FDN_NO_REBUILD is set): run `python 4dflownet_amd/build.py`"
                                % (os.path.basename(lib_path(test_hooks)),
                                   "not built" if have is None else "older than its sources (stamp %s..., sources %s...)"
                                   % (have[:12], source_stamp(test_hooks)[:12])))
    return build_library(test_hooks=test_hooks)


def build_library(force=False, verbose=False, test_hooks=False):
    """Compile every HIP translation unit for gfx950 and link the shared library.  Returns its path.
    test_hooks=True builds lib4dflow_hip_test.so (adds the fdn_debug_* entry points).  Safe to call from several processes."""
    if not force and is_current(test_hooks):
        return lib_path(test_hooks)
    with _build_lock():
        if not force and is_current(test_hooks):           # another process built it while we waited for the lock
            return lib_path(test_hooks)
        return _build_locked(verbose, test_hooks)


def _build_locked(verbose, test_hooks):
    lib = lib_path(test_hooks)
    extra = ["-DFDN_TEST_HOOKS"] if test_hooks else []
    stamp_file = lib + ".stamp"
    stamp = source_stamp(test_hooks)
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build", "test" if test_hooks else "product")
    os.makedirs(objdir, exist_ok=True)
    if os.path.exists(stamp_file):
        os.remove(stamp_file)                              # an interrupted build must not leave a matching stamp behind

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
    tmp = lib + ".tmp%d" % os.getpid()                     # link beside the target, then rename: a reader never maps a half-written file
    r = subprocess.run([hipcc, "-shared", "-fPIC", "--offload-arch=gfx950", "-o", tmp] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    if source_stamp(test_hooks) != stamp:
        os.remove(tmp)
        raise RuntimeError("sources changed while %s was being built; run the build again" % os.path.basename(lib))
    os.replace(tmp, lib)
    with open(stamp_file + ".tmp", "w") as f:
        f.write(stamp)
    os.replace(stamp_file + ".tmp", stamp_file)
    with open(os.path.join(HERE, "build", "build.log"), "a") as f:
        f.write("%s pid %d built %s stamp %s\n" % (time.strftime("%Y-%m-%d %H:%M:%S"), os.getpid(), os.path.basename(lib), stamp[:16]))
    return lib


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
    print(build_library(force="--force" in sys.argv, verbose=True, test_hooks=True))

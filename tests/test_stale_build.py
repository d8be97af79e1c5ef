"""A binary older than its sources is never executed: _lib.load() compares the build stamp with the sources
and rebuilds under a file lock, or raises with FDN_NO_REBUILD=1.  Runs on a scratch copy of the build machinery with a
one-function translation unit, so the real tree is never touched.  CPU only (hipcc cross-compiles)."""
import os
import shutil
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "4dflownet_amd")


def _scratch_tree(tmp_path):
    root = tmp_path / "root"
    pkg = root / "pkgcopy"
    (pkg / "csrc").mkdir(parents=True)
    (root / "include").mkdir()
    shutil.copy2(os.path.join(ROOT, "include", "fdn.h"), root / "include" / "fdn.h")
    for n in ("build.py", "_lib.py"):
        shutil.copy2(os.path.join(PKG, n), pkg / n)
    (pkg / "__init__.py").write_text("")
    # the stamp covers every csrc file: two real headers ride along, the only translation unit is a tiny one
    for n in ("fdn_common.h", "conv64_args.h"):
        shutil.copy2(os.path.join(PKG, "csrc", n), pkg / "csrc" / n)
    (pkg / "csrc" / "tiny.hip").write_text('extern "C" int fdn_version(void) { return 7; }\n')
    return root, pkg


DRIVER = textwrap.dedent('''
    import ctypes, importlib, os, sys
    sys.path.insert(0, sys.argv[1])
    build = importlib.import_module("pkgcopy.build")
    build.SOURCES = ["tiny.hip"]
    lib = importlib.import_module("pkgcopy._lib")
    sigs = {"fdn_version": (ctypes.c_int, [])}
    if os.environ.get("WORLD_SIZE", "1") != "1":          # two ranks race for the build lock
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
        dist.barrier()
    try:
        h = lib._open(lib.LIB_PATH, sigs)
        print("LOADED", h.fdn_version(), build.is_current())
    except lib.FdnError as e:
        print("FDNERROR", str(e).replace("\\n", " "))
''')


def _run(root, env_extra=None, **kw):
    script = os.path.join(str(root), "driver.py")
    if not os.path.exists(script):
        open(script, "w").write(DRIVER)
    env = dict(os.environ)
    env.pop("FDN_NO_REBUILD", None)
    env.update(env_extra or {})
    return subprocess.Popen([sys.executable, script, str(root)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, **kw)


def _out(p, timeout=300):
    out, _ = p.communicate(timeout=timeout)
    return out


def _builds(pkg):
    log = pkg / "build" / "build.log"
    return [l for l in log.read_text().splitlines() if "built lib4dflow_hip.so" in l] if log.exists() else []


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_stale_library_is_rebuilt_or_refused(tmp_path):
    root, pkg = _scratch_tree(tmp_path)
    # missing library + FDN_NO_REBUILD: loud error, nothing built
    out = _out(_run(root, {"FDN_NO_REBUILD": "1"}))
    assert "FDNERROR" in out and "not built" in out and not (pkg / "lib4dflow_hip.so").exists(), out
    # first load builds
    out = _out(_run(root))
    assert "LOADED 7 True" in out, out
    assert len(_builds(pkg)) == 1
    stamp1 = (pkg / "lib4dflow_hip.so.stamp").read_text()
    # current library: no rebuild
    out = _out(_run(root))
    assert "LOADED 7 True" in out and len(_builds(pkg)) == 1, out
    # edit a COMMENT in a source the stamp covers -> stale
    with open(pkg / "csrc" / "conv64_args.h", "a") as f:
        f.write("// touched by tests/test_stale_build.py\n")
    out = _out(_run(root, {"FDN_NO_REBUILD": "1"}))
    assert "FDNERROR" in out and "older than its sources" in out and len(_builds(pkg)) == 1, out
    # a functional edit: the rebuilt library is the one that gets loaded
    (pkg / "csrc" / "tiny.hip").write_text('extern "C" int fdn_version(void) { return 8; }\n')
    out = _out(_run(root))
    assert "LOADED 8 True" in out and len(_builds(pkg)) == 2, out
    assert (pkg / "lib4dflow_hip.so.stamp").read_text() != stamp1
    # a compile error is an FdnError, and leaves no stamp that could match later
    (pkg / "csrc" / "tiny.hip").write_text('extern "C" int fdn_version(void) { return nine; }\n')
    out = _out(_run(root))
    assert "FDNERROR" in out and "hipcc failed" in out, out
    assert not (pkg / "lib4dflow_hip.so.stamp").exists()


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="no hipcc")
def test_two_ranks_racing_for_a_stale_library_build_it_once(tmp_path):
    import socket
    root, pkg = _scratch_tree(tmp_path)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = [_run(root, {"WORLD_SIZE": "2", "RANK": str(r), "LOCAL_RANK": str(r), "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
             for r in range(2)]
    outs = [_out(p) for p in procs]
    assert all("LOADED 7 True" in o for o in outs), outs
    assert len(_builds(pkg)) == 1, (pkg / "build" / "build.log").read_text()


def test_stamp_does_not_depend_on_where_the_tree_lives(tmp_path):
    """The GPU box runs a snapshot of the tree under another root: identical sources must give the identical stamp there,
    or every call would rebuild (and no committed profile could be matched to the sources it was measured on)."""
    stamps = []
    for sub in ("a", "some/deeper/b"):
        d = tmp_path / sub
        d.mkdir(parents=True)
        root, pkg = _scratch_tree(d)
        out = subprocess.run([sys.executable, "-c", "import importlib,sys; sys.path.insert(0, %r); "
                              "print(importlib.import_module('pkgcopy.build').source_stamp())" % str(root)],
                             capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        stamps.append(out.stdout.strip())
    assert stamps[0] == stamps[1] and len(stamps[0]) == 64

"""CPU-side checks of the boundary: the library builds, loads and exports every symbol include/fdn.h declares."""
import ctypes
import os
import re
import subprocess

import pytest


def test_library_builds_and_exports_every_declared_symbol(fdn):
    from importlib import import_module
    build = import_module("4dflownet_amd.build")
    path = build.build_library()
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "fdn.h")).read()
    declared = set(re.findall(r"\b(fdn_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 15
    for name in sorted(declared):
        assert hasattr(lib, name), "missing export %s" % name
    # the ctypes table covers exactly the declared API
    assert declared == set(fdn._lib.SIGNATURES), declared ^ set(fdn._lib.SIGNATURES)
    assert fdn._lib.load().fdn_version() == 161 == fdn._lib.FDN_VERSION          # == FDN_VERSION of include/fdn.h (the C link test below prints the header's)
    # the product library carries no process-global switches (include/fdn.h: "no global mutable state"); the variant-forcing
    # hooks live in the test build only, which exports the full API as well
    exported = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True).stdout
    assert "fdn_debug" not in exported and "fdn_conv3d_fwd" in exported
    tpath = build.build_library(test_hooks=True)
    tlib = ctypes.CDLL(tpath)
    for name in sorted(declared) + sorted(fdn._lib.DEBUG_SIGNATURES):
        assert hasattr(tlib, name), "test build misses %s" % name


def test_no_cpu_fallback(fdn):
    import torch
    with pytest.raises(fdn.FdnError):
        fdn.ops.input_features(*[torch.zeros(8) for _ in range(6)])


def test_header_is_plain_c_and_links_from_c(fdn, tmp_path):
    """include/fdn.h must be consumable by a C compiler (the boundary is a C-ABI, not a C++ or Python one): compile a C99
    translation unit against it with gcc, link it to the shared library and run the two entry points that need no GPU."""
    import shutil
    from importlib import import_module
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    build = import_module("4dflownet_amd.build")
    lib = build.build_library()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi_smoke.c"
    src.write_text('#include <stdio.h>\n#include "fdn.h"\n'
                   'int main(void) {\n'
                   '    /* every prototype is visible to C: take the address of a few */\n'
                   '    int (*f)(const float*, float*, float*, void*) = fdn_pack_conv64_weights;\n'
                   '    size_t (*g)(int, int, int, int, int, int, int) = fdn_conv3d_wgrad_workspace_bytes;\n'
                   '    printf("%d %d %d %s\\n", fdn_version() == FDN_VERSION ? FDN_VERSION : -1, f != 0, (int)(g(8, 24, 24, 24, 64, 64, 3) > 0), FDN_CONV64_PACK_FLOATS == 423 * 4096 ? "ok" : "bad");\n'
                   '    /* an argument error comes back as a code + message, not as an exception */\n'
                   '    int rc = fdn_l2_sumsq(0, 0, 0, 0, 0);\n'
                   '    printf("%d %s\\n", rc, fdn_last_error());\n'
                   '    return 0;\n}\n')
    exe = tmp_path / "abi_smoke"
    r = subprocess.run([gcc, "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe), lib,
                        "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    assert lines[0].split()[1:] == ["1", "1", "ok"] and int(lines[0].split()[0]) == 161
    assert lines[1].startswith("-1 ") and "fdn_l2_sumsq" in lines[1]


def test_oracle_is_test_infrastructure_only():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/: nothing in the package, scripts/ or tools/
    imports it, and bench.py / __graft_entry__.py reach it only inside those two functions."""
    import ast
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for pat in ("4dflownet_amd/**/*.py", "scripts/*.py", "tools/*.py"):
        for f in glob.glob(os.path.join(root, pat), recursive=True):
            src = open(f).read()
            for node in ast.walk(ast.parse(src)):
                if isinstance(node, ast.Import):
                    assert not any(a.name.split(".")[0] == "oracle" for a in node.names), f
                if isinstance(node, ast.ImportFrom):
                    assert (node.module or "").split(".")[0] != "oracle", f
                if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "import_module":
                    assert not any(isinstance(a, ast.Constant) and str(a.value).startswith("oracle") for a in node.args), f
    allowed = {"bench.py": {"cpu_baseline"}, "__graft_entry__.py": {"smoke", "build"}}      # build() compiles the checker, never runs it
    for fname, funcs in allowed.items():
        tree = ast.parse(open(os.path.join(root, fname)).read())
        for node in tree.body:                         # module level: no oracle import
            if isinstance(node, (ast.Import, ast.ImportFrom)):
                names = [a.name for a in node.names] + [getattr(node, "module", "") or ""]
                assert not any(n.split(".")[0] == "oracle" for n in names), fname
        for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
            uses = any((isinstance(n, ast.ImportFrom) and (n.module or "").split(".")[0] == "oracle") or
                       (isinstance(n, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in n.names)) or
                       (isinstance(n, ast.Constant) and isinstance(n.value, str) and n.value.startswith("oracle."))
                       for n in ast.walk(fn))
            if uses:
                assert fn.name in funcs or fn.name.startswith("_cpu"), (fname, fn.name)


def test_round3_profiles_carry_the_sources_they_were_measured_on():
    """profiles/r3_*: every file names the commit and the library source stamp it was measured on.  A stamp
    that differs from the tree's is reported as a warning, not a failure: it means kernels changed after the last profile run."""
    import glob
    import json
    import warnings
    from importlib import import_module
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    # (files produced by tools/profile_round.sh + collect_profiles.py; the hand-collected ablation / A-B / soak logs of a round name their tree in words)
    stamped = ("_bench_kernel_stats.csv", "_bench_line.json", "_bench_line_nosecondary.json", "_pmc_sq.txt", "_pmc_traffic.json", "_pmc_traffic.txt",
               "_cfg4_bench_line.json", "_cfg4_kernel_stats.csv", "_cfg4_pmc_sq.txt", "_cfg4_pmc_traffic.json", "_cfg4_pmc_traffic.txt")
    files = sorted(f for f in glob.glob(os.path.join(root, "profiles", "r[3-9]_*")) if f.endswith(stamped))
    if not files:
        pytest.skip("no round-3/4 profiles collected yet")
    stamp = import_module("4dflownet_amd.build").source_stamp()
    latest = max(int(re.match(r"r(\d+)_", os.path.basename(f)).group(1)) for f in files)
    stale = []
    for f in files:
        if f.endswith(".json"):
            meta = json.load(open(f)).get("_meta")
            assert meta and re.fullmatch(r"[0-9a-f]{40}(\+dirty)?", meta["commit"]) and re.fullmatch(r"[0-9a-f]{64}", meta["lib_source_stamp"]), f
            got = meta["lib_source_stamp"]
        else:
            first = open(f).readline()
            m = re.match(r"# commit ([0-9a-f]{40}(?:\+dirty)?) lib_source_stamp ([0-9a-f]{64})", first)
            assert m, (f, first)
            got = m.group(2)
        if got != stamp and os.path.basename(f).startswith("r%d_" % latest):     # older rounds' files are history: header check only
            stale.append(os.path.basename(f))
    if stale:
        warnings.warn("profiles measured on other kernel sources than the tree's (re-run tools/profile_round.sh): %s" % ", ".join(stale))


def test_tracked_gpu_test_log_matches_the_tree():
    """profiles/r*_gputest.txt (tools/gputest_round.sh -> tools/collect_profiles.py) is the round's parity record: the whole
    `-m gpu` suite + smoke() on an MI355X, naming the kernel sources (lib_source_stamp) and the python / tests / oracle
    (suite_stamp) it ran on.  The log must be a green run; a missing log or one whose stamps differ from the
    tree's is reported as a WARNING, not a failure -- it means product code changed after the last recorded GPU run, and the
    round is not finished until `bash tools/gputest_round.sh` has been re-run and collected."""
    import glob
    import warnings
    from importlib import import_module
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    logs = sorted(glob.glob(os.path.join(root, "profiles", "r*_gputest.txt")), key=lambda p: int(re.search(r"r(\d+)_", p).group(1)))
    if not logs:
        warnings.warn("no tracked GPU test log under profiles/ (run tools/gputest_round.sh through gpurun, then tools/collect_profiles.py)")
        return
    body = open(logs[-1]).read()
    head = body.splitlines()[0]
    m = re.match(r"# commit ([0-9a-f]{40}(?:\+dirty)?) lib_source_stamp ([0-9a-f]{64}) suite_stamp ([0-9a-f]{64})", head)
    assert m, head
    assert "== pytest exit code 0" in body and "== smoke exit code 0" in body, "the tracked GPU test log is not a green run"
    assert re.search(r"^\d+ passed", body, re.M) and not re.search(r"^\d+ failed|\d+ error", body, re.M)
    assert "[wino-parity" in body                    # the per-layer Winograd parity tables travel with the log
    build = import_module("4dflownet_amd.build")
    stale = [k for k, got, want in (("kernels", m.group(2), build.source_stamp()), ("python/tests", m.group(3), build.suite_stamp()))
             if got != want]
    if stale:
        warnings.warn("%s: %s changed since the recorded GPU run -- re-run tools/gputest_round.sh and collect it" %
                      (os.path.basename(logs[-1]), " and ".join(stale)))


def test_pack_stream_query_follows_the_launchers_selection(fdn):
    """fdn_conv64_pack_streams runs the launcher's own selection in probe mode (no GPU needed): the streams named in include/fdn.h's
    FDN_ALGO_* table.  Bits: 1 direct, 2 1-D Winograd, 4 F(2,3)xF(4,3), 8 F(4,3)xF(4,3), 16 the same as three bf16 pieces."""
    q = fdn._lib.load().fdn_conv64_pack_streams
    FWD, DG, FUSED = 0, 1, 2
    AUTO, DIRECT, WINO_W, WINO_H2 = 0, 1, 2, 3
    for shp in ((8, 24, 24, 24), (8, 48, 48, 48), (1, 12, 12, 12)):          # H, W multiples of 4: the cfg2 grids
        assert q(*shp, AUTO, FWD) == 8 and q(*shp, AUTO, FUSED) == 8 | 2     # dgrad: inner box + the 1-D shell faces
        assert q(*shp, WINO_H2, FWD) == 4 and q(*shp, WINO_H2, FUSED) == 4 | 2
        assert q(*shp, WINO_W, FWD) == 2 and q(*shp, WINO_W, FUSED) == 2
        assert q(*shp, DIRECT, FWD) == 1 and q(*shp, DIRECT, FUSED) == 1 and q(*shp, DIRECT, DG) == 1
    assert q(2, 8, 6, 8, AUTO, FWD) == 4                                      # H only even
    assert q(2, 9, 9, 12, AUTO, FWD) == 2 and q(2, 9, 9, 12, AUTO, FUSED) == 2   # odd H: 1-D Winograd
    assert q(8, 18, 18, 18, AUTO, FWD) == 8 | 1 and q(8, 18, 18, 18, AUTO, FUSED) == 8 | 1   # W % 4 != 0: the aligned 16 x 16 box on F(4,3) x F(4,3), the strips direct
    assert q(2, 10, 10, 10, AUTO, FWD) == 1 and q(64, 10, 6, 6, AUTO, FUSED) == 1      # ... too few voxels / too small an aligned box for the split: direct
    assert q(8, 18, 18, 18, WINO_H2, FWD) == 1 and q(8, 18, 18, 18, 4, FWD) == 16 | 1   # (FDN_ALGO_WINO_BF16X3 reads the bf16 x 3 stream)
    assert q(8, 24, 24, 24, 4, FWD) == 16 and q(8, 24, 24, 24, 4, FUSED) == 16 | 2
    assert q(2, 10, 10, 10, AUTO, DG) == 8                                    # the padded grid is 12^3
    assert q(2, 10, 10, 10, 7, FWD) < 0 and q(0, 10, 10, 10, AUTO, FWD) < 0 and q(2, 10, 10, 10, AUTO, 3) < 0


def test_mask_query_and_multi_source_argument_checks_need_no_gpu(fdn):
    """fdn_conv64_mask_ok answers from the launcher's selection (no GPU needed); fdn_conv64_dgrad_fused_multi refuses bad source counts and
    packs that do not lie in one buffer before it touches the device (the pointers here are never dereferenced)."""
    import ctypes
    lib = fdn._lib.load()
    ok = lib.fdn_conv64_mask_ok
    assert ok(8, 24, 24, 24, 0) == 1 and ok(8, 48, 48, 48, 0) == 1 and ok(1, 5, 8, 12, 0) == 1
    assert ok(8, 18, 18, 18, 0) == 0 and ok(2, 9, 9, 12, 0) == 0 and ok(8, 24, 24, 24, 1) == 0 and ok(8, 24, 24, 24, 4) == 0
    assert ok(0, 24, 24, 24, 0) < 0
    fake = lambda *a: (ctypes.c_void_p * len(a))(*a)
    multi = lib.fdn_conv64_dgrad_fused_multi
    err = lambda: lib.fdn_last_error().decode()
    args = (0x1000, None, None, None, 2, 0.2, 0x2000, 2, 8, 8, 8, 0, None)
    assert multi(fake(0x10000, 0x20000), fake(0x30000, 0x40000), 4, *args) != 0 and "1..3" in err()
    assert multi(fake(0x10000, 0x20000), fake(0x30000, 0x30000 + (1 << 31)), 2, *args) != 0 and "1 GiB" in err()
    assert multi(fake(0x10000, None), fake(0x30000, 0x40000), 2, *args) != 0 and "source 1" in err()

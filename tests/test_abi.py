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
    assert fdn._lib.load().fdn_version() >= 100
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

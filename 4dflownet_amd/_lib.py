"""ctypes binding of lib4dflow_hip.so (include/fdn.h).  There is NO fallback: if the library is missing (and cannot be
built) or a call fails, an exception is raised.  A library older than the sources in the tree is never loaded: `load()` goes
through build.ensure_built(), which compares the build stamp and rebuilds under a file lock (or raises with FDN_NO_REBUILD=1)."""
import contextlib
import ctypes
import os

from . import build as _build

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib4dflow_hip.so")
TEST_LIB_PATH = os.path.join(_HERE, "lib4dflow_hip_test.so")     # same sources + fdn_debug_* hooks (tests / tools only)

c_fp = ctypes.c_void_p      # device pointer to float
c_i = ctypes.c_int
c_i64 = ctypes.c_int64
c_f = ctypes.c_float
c_sz = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/fdn.h one to one
SIGNATURES = {
    "fdn_version": (c_i, []),
    "fdn_last_error": (ctypes.c_char_p, []),
    "fdn_input_features": (c_i, [c_fp] * 8 + [c_i64, c_fp]),
    "fdn_pack_conv64_weights": (c_i, [c_fp, c_fp, c_fp, c_fp]),
    "fdn_conv3d_fwd": (c_i, [c_fp] * 7 + [c_i] * 10 + [c_f, c_i, c_fp]),
    "fdn_conv3d_dgrad": (c_i, [c_fp] * 4 + [c_i] * 10 + [c_fp]),
    "fdn_conv_cout1_dgrad_folded": (c_i, [c_fp, c_fp, c_fp, c_i, c_f, c_fp, c_fp, c_fp, c_sz, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "fdn_conv_cout1_dgrad_folded_mask": (c_i, [c_fp, c_fp, c_fp, c_i, c_f, c_fp, c_fp, c_fp, c_sz, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "fdn_fold_halo": (c_i, [c_fp, c_fp, c_fp, c_i, c_fp, c_fp, c_i, c_f, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "fdn_conv3d_dgrad_fused": (c_i, [c_fp] * 5 + [c_i, c_f, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "fdn_conv64_mask_ok": (c_i, [c_i] * 5),
    "fdn_conv64_fwd_mask": (c_i, [c_fp] * 6 + [c_i] * 5 + [c_f, c_i, c_fp]),
    "fdn_conv64_dgrad_fused_mask": (c_i, [c_fp] * 5 + [c_i, c_f, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "fdn_conv64_dgrad_fused_multi": (c_i, [c_fp, c_fp, c_i] + [c_fp] * 4 + [c_i, c_f, c_fp, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "fdn_conv3d_dgrad_fused_part": (c_i, [c_fp] * 5 + [c_i, c_f, c_fp, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "fdn_fold_halo_border": (c_i, [c_fp, c_fp, c_fp, c_i, c_fp, c_fp, c_i, c_f, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "fdn_conv1x1_dgrad": (c_i, [c_fp] * 6 + [c_i64, c_fp]),
    "fdn_conv3d_wgrad_workspace_bytes": (c_sz, [c_i] * 7),
    "fdn_conv3d_wgrad_batch_workspace_bytes": (c_sz, [c_i] * 5),
    "fdn_conv3d_wgrad_batch": (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_fp, c_sz] + [c_i] * 5 + [c_fp]),     # (the first four: HOST arrays of device pointers)
    "fdn_conv3d_wgrad": (c_i, [c_fp] * 6 + [c_sz] + [c_i] * 10 + [c_fp]),
    "fdn_upsample_trilinear_fwd": (c_i, [c_fp, c_fp] + [c_i] * 6 + [c_fp]),
    "fdn_upsample_trilinear_bwd": (c_i, [c_fp, c_fp, c_i, c_f, c_fp] + [c_i] * 6 + [c_fp]),
    "fdn_loss_metrics": (c_i, [c_fp] * 8 + [c_i, c_i64, c_fp]),
    "fdn_gather_patches": (c_i, [c_fp, c_fp, c_i, c_i, c_fp]),
    "fdn_l2_sumsq": (c_i, [c_fp, c_fp, c_i64, c_fp, c_fp]),
    "fdn_adam_step": (c_i, [c_fp] * 5 + [c_i64] + [c_f] * 5 + [c_fp, c_fp, c_fp]),
    "fdn_sum_partials": (c_i, [c_fp, c_i, c_fp, c_fp]),
    "fdn_l2_sumsq_partials": (c_i, [c_fp, c_fp, c_i64, c_fp, c_fp]),
    "fdn_pack_conv64_weights_batch": (c_i, [c_fp, c_fp, c_i, c_fp, c_fp]),
    "fdn_conv64_pack_streams": (c_i, [c_i] * 6),
    "fdn_pack_conv64_weights_batch_streams": (c_i, [c_fp, c_fp, c_i, c_fp, c_i, c_i, c_fp]),
    # bf16 activation path
    "fdn_pack_conv64_weights_bf16": (c_i, [c_fp, c_fp, c_fp, c_fp]),
    "fdn_pack_conv64_weights_bf16_batch": (c_i, [c_fp, c_fp, c_i, c_fp, c_fp]),
    "fdn_conv64_fwd_bf16": (c_i, [c_fp] * 5 + [c_i] * 5 + [c_f, c_fp]),
    "fdn_conv64_dgrad_fused_bf16": (c_i, [c_fp] * 5 + [c_i, c_f, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "fdn_conv64_fwd_bf16_mask": (c_i, [c_fp] * 6 + [c_i] * 5 + [c_f, c_fp]),
    "fdn_conv64_dgrad_fused_bf16_mask": (c_i, [c_fp] * 6 + [c_i, c_f, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "fdn_conv64_dgrad_fused_bf16_multi": (c_i, [c_fp, c_fp, c_i] + [c_fp] * 4 + [c_i, c_f, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "fdn_fold_halo_border_bf16": (c_i, [c_fp, c_fp, c_fp, c_i, c_fp, c_fp, c_i, c_f, c_fp, c_i, c_i, c_i, c_i, c_fp]),
    "fdn_input_features_bf16": (c_i, [c_fp] * 8 + [c_i64, c_fp]),
    "fdn_conv3d_fwd_bf16": (c_i, [c_fp] * 7 + [c_i] * 10 + [c_f, c_fp]),
    "fdn_conv3d_wgrad_bf16_workspace_bytes": (c_sz, [c_i] * 7),
    "fdn_conv3d_wgrad_bf16": (c_i, [c_fp] * 6 + [c_sz] + [c_i] * 9 + [c_fp]),
    "fdn_conv3d_wgrad_bf16_batch_workspace_bytes": (c_sz, [c_i] * 5),
    "fdn_conv3d_wgrad_bf16_batch": (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_fp, c_sz] + [c_i] * 4 + [c_fp]),     # (the first four: HOST arrays of device pointers)
    "fdn_conv_cout1_dgrad_folded_bf16": (c_i, [c_fp, c_fp, c_fp, c_i, c_f, c_fp, c_fp, c_fp, c_sz, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "fdn_conv_cout1_dgrad_folded_bf16_mask": (c_i, [c_fp, c_fp, c_fp, c_fp, c_i, c_f, c_fp, c_fp, c_fp, c_sz, c_i, c_i, c_i, c_i, c_i, c_i, c_fp]),
    "fdn_conv1x1_dgrad_bf16": (c_i, [c_fp] * 6 + [c_i64, c_fp]),
    "fdn_upsample_trilinear_fwd_bf16": (c_i, [c_fp, c_fp] + [c_i] * 6 + [c_fp]),
    "fdn_upsample_trilinear_bwd_bf16": (c_i, [c_fp, c_fp, c_i, c_f, c_fp] + [c_i] * 6 + [c_fp]),
}


# test-build-only entry points (csrc: #ifdef FDN_TEST_HOOKS); NOT part of include/fdn.h
DEBUG_SIGNATURES = {
    "fdn_debug_set_conv64_mt": (c_i, [c_i]),
    "fdn_debug_set_conv64_dbg": (c_i, [c_i]),
    "fdn_debug_set_conv64_shell_slabs": (c_i, [c_i]),
    "fdn_debug_set_conv64_wface_direct": (c_i, [c_i]),
    "fdn_debug_set_conv64_split_dgrad": (c_i, [c_i]),
    "fdn_debug_set_conv64_bf16_mt": (c_i, [c_i]),
    "fdn_debug_set_conv64_bf16_dbg": (c_i, [c_i]),
    "fdn_debug_set_conv64_bf16_mode2": (c_i, [c_i]),
    "fdn_debug_set_heads_mfma": (c_i, [c_i]),
    "fdn_debug_set_upsample_bwd_hb": (c_i, [c_i]),
    "fdn_debug_set_wgrad64_direct": (c_i, [c_i]),
    "fdn_debug_set_conv64_wino_dbg": (c_i, [c_i]),
    "fdn_debug_set_conv64_wino_tile": (c_i, [c_i]),
    "fdn_debug_set_conv64_wino2d_dbg": (c_i, [c_i]),
    "fdn_debug_set_conv64_wino2d_tile": (c_i, [c_i]),
    "fdn_debug_set_conv64_wino2d_variant": (c_i, [c_i]),
    "fdn_debug_set_conv64_wino2d_mb": (c_i, [c_i]),
    "fdn_debug_set_cin3_mfma": (c_i, [c_i]),
    "fdn_debug_set_conv1x1_mfma": (c_i, [c_i]),
    "fdn_debug_set_wgrad64_wino_dbg": (c_i, [c_i]),
    "fdn_debug_set_wgrad64_wino_nodep": (c_i, [c_i]),
    "fdn_debug_set_wgrad64_bf16_dbg": (c_i, [c_i]),
    "fdn_debug_set_wgrad64_bf16_variant": (c_i, [c_i]),
}


class FdnError(RuntimeError):
    pass


FDN_VERSION = 161    # include/fdn.h


_product = None
_test = None
_lib = None          # the library load() hands out: the product build, except inside `with test_build():`


def _open(path, sigs, test_hooks=False):
    try:
        built = _build.ensure_built(test_hooks=test_hooks)
    except Exception as e:                        # hipcc missing / compile error / FDN_NO_REBUILD with a stale binary
        raise FdnError("%s is not usable (%s): %s.  Run `python __graft_entry__.py` or `python 4dflownet_amd/build.py`; "
                       "there is no CPU fallback." % (os.path.basename(path), path, e)) from e
    assert os.path.samefile(built, path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)          # AttributeError if an export is missing
        fn.restype = res
        fn.argtypes = args
    return lib


def _check_version(lib, path):
    """The python mirror of include/fdn.h hard-codes sizes the library trusts (ops.CONV64_PACK_FLOATS: a pack buffer sized for an older
    header is too small for this library's pack kernel): a library of another version -- FDN_NO_REBUILD with an external build -- is refused."""
    have = lib.fdn_version()
    if have != FDN_VERSION:
        raise FdnError("%s reports fdn_version() = %d, this package mirrors include/fdn.h at FDN_VERSION %d: rebuild (python 4dflownet_amd/build.py)"
                       % (os.path.basename(path), have, FDN_VERSION))
    return lib


def load():
    """Load the shared library (once) and attach the prototypes.  Raises FdnError if it is not built."""
    global _lib, _product
    if _lib is not None:
        return _lib
    if _product is None:
        _product = _check_version(_open(LIB_PATH, SIGNATURES), LIB_PATH)
    _lib = _product
    return _lib


@contextlib.contextmanager
def test_build():
    """Tests / tools only: route every call of the operator layer through lib4dflow_hip_test.so -- the same sources
    compiled with -DFDN_TEST_HOOKS -- so kernel variants the planner would not pick at a given size can be forced
    (fdn_debug_*).  Yields the test library; restores the product library on exit."""
    global _lib, _test
    if _test is None:
        sigs = dict(SIGNATURES)
        sigs.update(DEBUG_SIGNATURES)
        _test = _check_version(_open(TEST_LIB_PATH, sigs, test_hooks=True), TEST_LIB_PATH)
    prev = _lib
    _lib = _test
    try:
        yield _test
    finally:
        _lib = prev


def check(rc, what):
    if rc != 0:
        msg = load().fdn_last_error()
        raise FdnError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))

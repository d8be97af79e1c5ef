"""Operator layer of the bf16 activation path (BASELINE.json configs[3]): torch.bfloat16 activation tensors,
fp32 parameters -> raw pointers -> lib4dflow_hip.so.  Same rules as ops.py: GPU only, no fallback."""
import torch

from . import _lib
from ._lib import FdnError, check
from .ops import ACT_NONE, ACT_RELU, ACT_LEAKY, LEAKY_ALPHA, _stream

BF16 = torch.bfloat16


def _pb(t, name="tensor", allow_none=False, dtype=BF16):
    if t is None:
        if allow_none:
            return None
        raise FdnError("%s is None" % name)
    if not t.is_cuda:
        raise FdnError("%s must live on the GPU; the HIP path has no CPU fallback" % name)
    if t.dtype != dtype:
        raise FdnError("%s must be %s (got %s)" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise FdnError("%s must be contiguous" % name)
    return t.data_ptr()


def _pf(t, name="tensor", allow_none=False):
    return _pb(t, name, allow_none, torch.float32)


def pack_conv64_weights(w, wp_fwd=None, wp_dgrad=None, want_dgrad=True):
    """fp32 (3,3,3,64,64) -> bf16 operand streams (27*64*64 each)."""
    if wp_fwd is None:
        wp_fwd = torch.empty(27 * 64 * 64, device=w.device, dtype=BF16)
    if wp_dgrad is None and want_dgrad:
        wp_dgrad = torch.empty(27 * 64 * 64, device=w.device, dtype=BF16)
    check(_lib.load().fdn_pack_conv64_weights_bf16(_pf(w, "w"), _pb(wp_fwd), _pb(wp_dgrad, allow_none=True), _stream()),
          "fdn_pack_conv64_weights_bf16")
    return wp_fwd, wp_dgrad


def conv64_fwd(x, wpack, bias=None, act=ACT_NONE, alpha=LEAKY_ALPHA, residual=None, out=None):
    N, D, H, W, C = x.shape
    if C != 64:
        raise FdnError("conv64_fwd: 64 input channels expected, got %d" % C)
    if out is None:
        out = torch.empty_like(x)
    check(_lib.load().fdn_conv64_fwd_bf16(_pb(x, "x"), _pb(wpack, "wpack"), _pf(bias, allow_none=True),
                                          _pb(residual, allow_none=True), _pb(out, "out"), N, D, H, W, act, float(alpha),
                                          _stream()), "fdn_conv64_fwd_bf16")
    return out


def conv64_dgrad_fused(dz, wpack_dgrad, dxpad, out, skip=None, y_prev=None, act=ACT_NONE, alpha=LEAKY_ALPHA):
    N, D, H, W = dz.shape[:4]
    check(_lib.load().fdn_conv64_dgrad_fused_bf16(_pb(dz, "dz"), _pb(wpack_dgrad, "wpack"), _pf(dxpad, "dxpad"),
                                                  _pb(skip, allow_none=True), _pb(y_prev, allow_none=True), act,
                                                  float(alpha), _pb(out, "out"), N, D, H, W, _stream()),
          "fdn_conv64_dgrad_fused_bf16")
    return out


def fold_halo_border(dxpads, out, skip=None, y_prev=None, act=ACT_NONE, alpha=LEAKY_ALPHA):
    N, D, H, W = out.shape[:4]
    ptrs = [_pf(t) for t in dxpads] + [None] * (3 - len(dxpads))
    check(_lib.load().fdn_fold_halo_border_bf16(ptrs[0], ptrs[1], ptrs[2], len(dxpads), _pb(skip, allow_none=True),
                                                _pb(y_prev, allow_none=True), act, float(alpha), _pb(out), N, D, H, W,
                                                _stream()), "fdn_fold_halo_border_bf16")
    return out

"""Operator layer of the bf16 activation path (BASELINE.json configs[3]): torch.bfloat16 activation tensors,
fp32 parameters / parameter gradients / prediction -> raw pointers -> lib4dflow_hip.so.

Same function names and argument meaning as ops.py, so network.py selects one of the two modules by dtype.
Same rules: GPU only, no fallback."""
import torch

from . import _lib
from ._lib import FdnError, check
from .ops import (ACT_NONE, ACT_RELU, ACT_LEAKY, LEAKY_ALPHA, _stream, loss_metrics, l2_sumsq, adam_step)  # noqa: F401

BF16 = torch.bfloat16
ACT_DTYPE = BF16


def _pt(t, dtype, name="tensor", allow_none=False):
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


def _pb(t, name="tensor", allow_none=False):
    return _pt(t, BF16, name, allow_none)


def _pf(t, name="tensor", allow_none=False):
    return _pt(t, torch.float32, name, allow_none)


def input_features(u, v, w, mu, mv, mw, phase=None, pc=None):
    shp = u.shape[:-1] if u.shape[-1] == 1 else u.shape
    if phase is None:
        phase = torch.empty(tuple(shp) + (3,), device=u.device, dtype=BF16)
    if pc is None:
        pc = torch.empty(tuple(shp) + (3,), device=u.device, dtype=BF16)
    check(_lib.load().fdn_input_features_bf16(_pf(u), _pf(v), _pf(w), _pf(mu), _pf(mv), _pf(mw), _pb(phase), _pb(pc),
                                              u.numel(), _stream()), "fdn_input_features_bf16")
    return phase, pc


def pack_conv64_weights(w, wp_fwd=None, wp_dgrad=None, want_dgrad=True):
    """fp32 (3,3,3,64,64) -> bf16 operand streams (27*64*64 each)."""
    if tuple(w.shape) != (3, 3, 3, 64, 64):
        raise FdnError("pack_conv64_weights: expected (3,3,3,64,64), got %s" % (tuple(w.shape),))
    if wp_fwd is None:
        wp_fwd = torch.empty(27 * 64 * 64, device=w.device, dtype=BF16)
    if wp_dgrad is None and want_dgrad:
        wp_dgrad = torch.empty(27 * 64 * 64, device=w.device, dtype=BF16)
    check(_lib.load().fdn_pack_conv64_weights_bf16(_pf(w, "w"), _pb(wp_fwd), _pb(wp_dgrad, allow_none=True), _stream()),
          "fdn_pack_conv64_weights_bf16")
    return wp_fwd, wp_dgrad


def pack_conv64_weights_batch(w_flat, w_offsets, packs):
    """Every 64->64 kernel of the flat fp32 parameter buffer in one launch.  w_offsets: int64 DEVICE tensor of float offsets;
    packs: bf16 (n_layers, 2, 27*64*64)."""
    n = w_offsets.numel()
    if w_offsets.dtype != torch.int64 or not w_offsets.is_cuda or packs.dtype != BF16 or packs.numel() != n * 2 * 27 * 64 * 64:
        raise FdnError("pack_conv64_weights_batch (bf16): bad offsets / packs")
    check(_lib.load().fdn_pack_conv64_weights_bf16_batch(_pf(w_flat, "w"), w_offsets.data_ptr(), n, _pb(packs), _stream()),
          "fdn_pack_conv64_weights_bf16_batch")
    return packs


def conv3d_fwd(x, w, bias=None, act=ACT_NONE, alpha=LEAKY_ALPHA, residual=None, x2=None, wpack=None, out=None,
               ldy=None, y_coff=0, algo=0, mask=None):
    """x bf16 (N,D,H,W,Cin[/2 if x2]); w fp32 Keras layout; output bf16, except Cout == 1 (prediction) -> fp32.
    algo is accepted for signature parity with ops.conv3d_fwd and ignored: the bf16 kernels are direct convolutions.
    mask (64->64 only; new_sign_mask(out)): also receives the sign mask of the output, see conv64_fwd."""
    N, D, H, W = x.shape[:4]
    K, Cin, Cout = w.shape[0], w.shape[3], w.shape[4]
    if mask is not None:
        if (K, Cin, Cout) != (3, 64, 64) or x2 is not None or ldy not in (None, 64) or y_coff:
            raise FdnError("conv3d_fwd (bf16): a sign mask belongs to a dense 64->64 3x3x3 layer")
        if wpack is None:
            wpack, _ = pack_conv64_weights(w, want_dgrad=False)
        return conv64_fwd(x, wpack, bias, act, alpha, residual, out, mask=mask)
    odt = torch.float32 if Cout == 1 else BF16
    if out is None:
        out = torch.empty((N, D, H, W, Cout), device=x.device, dtype=odt)
        ldy = Cout
    elif ldy is None:
        ldy = out.shape[-1]
    if Cin == 64 and Cout == 64 and K == 3 and wpack is None:
        wpack, _ = pack_conv64_weights(w, want_dgrad=False)
    check(_lib.load().fdn_conv3d_fwd_bf16(_pb(x, "x"), _pb(x2, allow_none=True), _pf(w, "w"), _pb(wpack, allow_none=True),
                                          _pf(bias, allow_none=True), _pb(residual, allow_none=True), _pt(out, odt, "out"),
                                          N, D, H, W, Cin, Cout, K, ldy, y_coff, act, float(alpha), _stream()),
          "fdn_conv3d_fwd_bf16")
    return out


def new_sign_mask(y):
    """Sign-mask buffer for a bf16 (N,D,H,W,64) tensor: 64 bits per voxel as four int16 words (include/fdn.h)."""
    return torch.empty(tuple(y.shape[:4]) + (4,), device=y.device, dtype=torch.int16)


def _pm(t, name="mask", allow_none=False):
    if t is None and allow_none:
        return None
    if t.dtype != torch.int16 or not t.is_cuda or not t.is_contiguous():
        raise FdnError("%s: contiguous int16 CUDA tensor (N,D,H,W,4) expected" % name)
    return t.data_ptr()


def conv64_fwd(x, wpack, bias=None, act=ACT_NONE, alpha=LEAKY_ALPHA, residual=None, out=None, mask=None):
    """mask (new_sign_mask(out)), if given, receives bit c of voxel v = (out[v][c] > 0): what the fused dgrad needs of `out` for act'."""
    N, D, H, W, C = x.shape
    if C != 64:
        raise FdnError("conv64_fwd: 64 input channels expected, got %d" % C)
    if out is None:
        out = torch.empty_like(x)
    if mask is not None and mask.numel() != N * D * H * W * 4:
        raise FdnError("conv64_fwd: mask needs %d int16 words" % (N * D * H * W * 4))
    check(_lib.load().fdn_conv64_fwd_bf16_mask(_pb(x, "x"), _pb(wpack, "wpack"), _pf(bias, allow_none=True),
                                               _pb(residual, allow_none=True), _pb(out, "out"), _pm(mask, allow_none=True), N, D, H, W, act,
                                               float(alpha), _stream()), "fdn_conv64_fwd_bf16_mask")
    return out


def conv3d_dgrad_fused(dz, wpack_dgrad, dxpad, out, skip=None, y_prev=None, act=ACT_NONE, alpha=LEAKY_ALPHA, algo=0, mask=None):
    """mask: the sign mask the forward wrote beside y_prev (conv64_fwd(mask=...)); read instead of y_prev for act'."""
    N, D, H, W = dz.shape[:4]
    if mask is not None and mask.numel() != N * D * H * W * 4:
        raise FdnError("conv3d_dgrad_fused: mask needs %d int16 words" % (N * D * H * W * 4))
    check(_lib.load().fdn_conv64_dgrad_fused_bf16_mask(_pb(dz, "dz"), _pb(wpack_dgrad, "wpack"), _pf(dxpad, "dxpad"),
                                                       _pb(skip, allow_none=True), _pb(y_prev, allow_none=True),
                                                       _pm(mask, allow_none=True), act, float(alpha), _pb(out, "out"), N, D, H, W,
                                                       _stream()), "fdn_conv64_dgrad_fused_bf16_mask")
    return out


conv64_dgrad_fused = conv3d_dgrad_fused


def conv3d_dgrad_fused_multi(dzs, wpacks_dgrad, dxpad, out, skip=None, y_prev=None, act=ACT_NONE, alpha=LEAKY_ALPHA, algo=0, mask=None):
    """The fused dgrad of 1..3 64->64 layers that share their input, as ONE launch (ops.conv3d_dgrad_fused_multi for bf16 activations): the
    sum over the sources stays in the fp32 accumulators (the chained launches round the running sum to bf16 after every source)."""
    import ctypes
    n = len(dzs)
    N, D, H, W = dzs[0].shape[:4]
    if not 1 <= n <= 3 or len(wpacks_dgrad) != n or any(tuple(t.shape) != tuple(dzs[0].shape) for t in dzs):
        raise FdnError("conv3d_dgrad_fused_multi: 1..3 sources of one shape, one pack each")
    if mask is not None and mask.numel() != N * D * H * W * 4:
        raise FdnError("conv3d_dgrad_fused_multi: mask needs %d int16 words" % (N * D * H * W * 4))
    tz = (ctypes.c_void_p * n)(*[_pb(t, "dz") for t in dzs])
    tw = (ctypes.c_void_p * n)(*[_pb(t, "wpack") for t in wpacks_dgrad])
    check(_lib.load().fdn_conv64_dgrad_fused_bf16_multi(tz, tw, n, _pf(dxpad, "dxpad"), _pb(skip, allow_none=True), _pb(y_prev, allow_none=True),
                                                        _pm(mask, allow_none=True), act, float(alpha), _pb(out, "out"), N, D, H, W, _stream()),
          "fdn_conv64_dgrad_fused_bf16_multi")
    return out


def fold_halo_border(dxpads, out, skip=None, y_prev=None, act=ACT_NONE, alpha=LEAKY_ALPHA):
    N, D, H, W = out.shape[:4]
    ptrs = [_pf(t) for t in dxpads] + [None] * (3 - len(dxpads))
    check(_lib.load().fdn_fold_halo_border_bf16(ptrs[0], ptrs[1], ptrs[2], len(dxpads), _pb(skip, allow_none=True),
                                                _pb(y_prev, allow_none=True), act, float(alpha), _pb(out), N, D, H, W,
                                                _stream()), "fdn_fold_halo_border_bf16")
    return out


def conv_cout1_dgrad_folded(dz, w, spatial, y_prev=None, act=ACT_NONE, alpha=LEAKY_ALPHA, lddz=1, dz_coff=0, out=None,
                            dbias_prev=None, workspace=None, mask=None):
    """64->1 head dgrad (dz = fp32 prediction gradient) + halo fold + act'(y_prev): bf16 (N,D,H,W,64).
    mask: the sign mask conv64_fwd(mask=...) wrote beside y_prev; read instead of y_prev for act'."""
    N, D, H, W = spatial
    if out is None:
        out = torch.empty((N, D, H, W, 64), device=dz.device, dtype=BF16)
    if dbias_prev is not None and workspace is None:
        workspace = torch.empty(2048 * 64, device=dz.device, dtype=torch.float32)
    if mask is not None and mask.numel() != N * D * H * W * 4:
        raise FdnError("conv_cout1_dgrad_folded: mask needs %d int16 words" % (N * D * H * W * 4))
    wsb = 0 if workspace is None else workspace.numel() * workspace.element_size()
    check(_lib.load().fdn_conv_cout1_dgrad_folded_bf16_mask(_pf(dz, "dz"), _pf(w, "w"), _pb(y_prev, allow_none=True), _pm(mask, allow_none=True),
                                                            act, float(alpha), _pb(out), _pf(dbias_prev, allow_none=True),
                                                            _pf(workspace, allow_none=True), wsb, N, D, H, W, lddz, dz_coff,
                                                            _stream()), "fdn_conv_cout1_dgrad_folded_bf16_mask")
    return out


def conv1x1_dgrad(dz, w, ya, yb, dxa=None, dxb=None):
    nvox = dz.numel() // 64
    if dxa is None:
        dxa = torch.empty_like(ya)
    if dxb is None:
        dxb = torch.empty_like(yb)
    check(_lib.load().fdn_conv1x1_dgrad_bf16(_pb(dz), _pf(w), _pb(ya), _pb(yb), _pb(dxa), _pb(dxb), nvox, _stream()),
          "fdn_conv1x1_dgrad_bf16")
    return dxa, dxb


def wgrad_workspace_bytes(N, D, H, W, Cin, Cout, K):
    return int(_lib.load().fdn_conv3d_wgrad_bf16_workspace_bytes(N, D, H, W, Cin, Cout, K))


def conv3d_wgrad(x, dz, K, Cin, Cout, x2=None, want_bias=False, dw=None, dbias=None, workspace=None, lddz=None,
                 dz_coff=0, algo=0):
    """x bf16; dz bf16, except Cout == 1 where dz is the fp32 prediction gradient; dw / dbias fp32."""
    N, D, H, W = x.shape[:4]
    if lddz is None:
        lddz = dz.shape[-1]
    if dw is None:
        dw = torch.empty((K, K, K, Cin, Cout), device=x.device, dtype=torch.float32)
    if want_bias and dbias is None:
        dbias = torch.empty((Cout,), device=x.device, dtype=torch.float32)
    need = wgrad_workspace_bytes(N, D, H, W, Cin, Cout, K)
    if workspace is None:
        workspace = torch.empty((need + 3) // 4, device=x.device, dtype=torch.float32)
    check(_lib.load().fdn_conv3d_wgrad_bf16(_pb(x, "x"), _pb(x2, allow_none=True),
                                            _pt(dz, torch.float32 if Cout == 1 else BF16, "dz"), _pf(dw, "dw"),
                                            _pf(dbias, allow_none=True), _pf(workspace, "workspace"),
                                            workspace.numel() * workspace.element_size(), N, D, H, W, Cin, Cout, K, lddz,
                                            dz_coff, _stream()), "fdn_conv3d_wgrad_bf16")
    return dw, dbias


def wgrad_batch_workspace_bytes(n_layers, N, D, H, W):
    return int(_lib.load().fdn_conv3d_wgrad_bf16_batch_workspace_bytes(n_layers, N, D, H, W))


def conv3d_wgrad_batch(xs, dzs, dws, dbiases=None, workspace=None, algo=0):
    """Weight gradients of several 64->64 3x3x3 layers that share one grid, up to seven per launch (fdn_conv3d_wgrad_bf16_batch): xs / dzs
    bf16 (N,D,H,W,64), dws fp32 (3,3,3,64,64); dbiases: None or a list with None / fp32 (64,) entries.  algo: signature parity, ignored."""
    import ctypes
    n = len(xs)
    if not (n and len(dzs) == n and len(dws) == n and (dbiases is None or len(dbiases) == n)):
        raise ValueError("conv3d_wgrad_batch: xs, dzs, dws (and dbiases) must be lists of one length")
    N, D, H, W = xs[0].shape[:4]
    for x, dz in zip(xs, dzs):
        if tuple(x.shape) != (N, D, H, W, 64) or tuple(dz.shape) != (N, D, H, W, 64):
            raise ValueError("conv3d_wgrad_batch: every layer must have the grid %s with 64 channels" % ((N, D, H, W),))
    need = wgrad_batch_workspace_bytes(n, N, D, H, W)
    if workspace is None:
        workspace = torch.empty((need + 3) // 4, device=xs[0].device, dtype=torch.float32)
    tx = (ctypes.c_void_p * n)(*[_pb(t, "x") for t in xs])
    tz = (ctypes.c_void_p * n)(*[_pb(t, "dz") for t in dzs])
    tw = (ctypes.c_void_p * n)(*[_pf(t, "dw") for t in dws])
    tb = (ctypes.c_void_p * n)(*[_pf(t, "dbias", allow_none=True) for t in dbiases]) if dbiases is not None and any(b is not None for b in dbiases) else None
    check(_lib.load().fdn_conv3d_wgrad_bf16_batch(tx, tz, tw, tb, n, _pf(workspace, "workspace"), workspace.numel() * workspace.element_size(),
                                                  N, D, H, W, _stream()), "fdn_conv3d_wgrad_bf16_batch")
    return dws


def upsample_trilinear_fwd(x, R, out=None):
    N, D, H, W, C = x.shape
    if out is None:
        out = torch.empty((N, D * R, H * R, W * R, C), device=x.device, dtype=BF16)
    check(_lib.load().fdn_upsample_trilinear_fwd_bf16(_pb(x), _pb(out), N, D, H, W, C, R, _stream()),
          "fdn_upsample_trilinear_fwd_bf16")
    return out


def upsample_trilinear_bwd(dy, R, y_prev=None, act=ACT_NONE, alpha=LEAKY_ALPHA, out=None):
    N, OD, OH, OW, C = dy.shape
    D, H, W = OD // R, OH // R, OW // R
    if out is None:
        out = torch.empty((N, D, H, W, C), device=dy.device, dtype=BF16)
    check(_lib.load().fdn_upsample_trilinear_bwd_bf16(_pb(dy), _pb(y_prev, allow_none=True), act, float(alpha), _pb(out), N,
                                                      D, H, W, C, R, _stream()), "fdn_upsample_trilinear_bwd_bf16")
    return out

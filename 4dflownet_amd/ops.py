"""Operator layer: torch tensors (containers only) -> raw pointers -> lib4dflow_hip.so.

Every function launches asynchronously on torch's current HIP stream and returns its output tensor(s).
Tensors must be fp32, contiguous and on a ROCm device; anything else raises (no CPU path)."""
import ctypes

import torch

from . import _lib
from ._lib import FdnError, check

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
# FDN_ALGO_*: per-call algorithm of the 64->64 3x3x3 entry points (AUTO = Winograd along W when W % 4 == 0, DIRECT = never)
# ALGO_WINO_BF16X3: like AUTO, the F(4,3)xF(4,3) products on the bf16 matrix pipe (operands split exactly into 3 bf16 pieces, 6 terms, fp32 accumulation)
ALGO_AUTO, ALGO_DIRECT, ALGO_WINO_W, ALGO_WINO_H2, ALGO_WINO_BF16X3 = 0, 1, 2, 3, 4
LEAKY_ALPHA = 0.2
# FDN_CONV64_PACK_FLOATS (mirrored; _lib.load() checks fdn_version() against FDN_VERSION below): direct stream (27 taps) + Winograd F(4,3)
# stream (54) + 2-D F(2,3)xF(4,3) stream (72) + 2-D F(4,3)xF(4,3) stream (108) + the same as three bf16 pieces per value (162 float-sized slots)
CONV64_PACK_FLOATS = 423 * 64 * 64


def _p(t, name="tensor", allow_none=False):
    if t is None:
        if allow_none:
            return None
        raise FdnError("%s is None" % name)
    if not t.is_cuda:
        raise FdnError("%s must live on the GPU; the HIP path has no CPU fallback" % name)
    if t.dtype != torch.float32 and t.dtype != torch.uint8:
        raise FdnError("%s must be float32 (got %s)" % (name, t.dtype))
    if not t.is_contiguous():
        raise FdnError("%s must be contiguous" % name)
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def input_features(u, v, w, mu, mv, mw, phase=None, pc=None):
    shp = u.shape[:-1] if u.shape[-1] == 1 else u.shape
    nvox = u.numel()
    if phase is None:
        phase = torch.empty(tuple(shp) + (3,), device=u.device, dtype=torch.float32)
    if pc is None:
        pc = torch.empty(tuple(shp) + (3,), device=u.device, dtype=torch.float32)
    check(_lib.load().fdn_input_features(_p(u), _p(v), _p(w), _p(mu), _p(mv), _p(mw), _p(phase), _p(pc), nvox, _stream()),
          "fdn_input_features")
    return phase, pc


def pack_conv64_weights(w, wp_fwd=None, wp_dgrad=None, want_dgrad=True):
    if tuple(w.shape) != (3, 3, 3, 64, 64):
        raise FdnError("pack_conv64_weights: expected (3,3,3,64,64), got %s" % (tuple(w.shape),))
    if wp_fwd is None:
        wp_fwd = torch.empty(CONV64_PACK_FLOATS, device=w.device, dtype=torch.float32)
    if wp_dgrad is None and want_dgrad:
        wp_dgrad = torch.empty(CONV64_PACK_FLOATS, device=w.device, dtype=torch.float32)
    check(_lib.load().fdn_pack_conv64_weights(_p(w), _p(wp_fwd), _p(wp_dgrad, allow_none=True), _stream()),
          "fdn_pack_conv64_weights")
    return wp_fwd, wp_dgrad


def conv64_mask_ok(N, D, H, W, algo=ALGO_AUTO):
    """True when the 64->64 forward and fused dgrad of an (N,D,H,W) grid can write / read sign masks (fdn_conv64_mask_ok)."""
    r = _lib.load().fdn_conv64_mask_ok(int(N), int(D), int(H), int(W), int(algo))
    check(min(r, 0), "fdn_conv64_mask_ok")
    return r == 1


def new_sign_mask(y):
    """Sign-mask buffer of an fp32 (N,D,H,W,64) tensor: four planes [cout / 16][voxel] of int16 words (include/fdn.h)."""
    return torch.empty((4, y.shape[0] * y.shape[1] * y.shape[2] * y.shape[3]), device=y.device, dtype=torch.int16)


def _pm(t, name="mask"):
    if not t.is_cuda or t.dtype != torch.int16 or not t.is_contiguous():
        raise FdnError("%s must be a contiguous int16 tensor on the GPU" % name)
    return t.data_ptr()


def conv3d_fwd(x, w, bias=None, act=ACT_NONE, alpha=LEAKY_ALPHA, residual=None, x2=None, wpack=None, out=None,
               ldy=None, y_coff=0, algo=ALGO_AUTO, mask=None):
    """x (N,D,H,W,Cin[/2 if x2]); w Keras layout (K,K,K,Cin,Cout).
    mask (64->64 only, new_sign_mask(out), where conv64_mask_ok): also receives the sign mask of the output."""
    N, D, H, W = x.shape[:4]
    K, Cin, Cout = w.shape[0], w.shape[3], w.shape[4]
    if mask is not None:
        if (K, Cin, Cout) != (3, 64, 64) or x2 is not None or (ldy not in (None, 64)) or y_coff != 0 or mask.numel() != 4 * N * D * H * W:
            raise FdnError("conv3d_fwd: a sign mask (4 x %d int16 words) belongs to a dense 64->64 3x3x3 layer" % (N * D * H * W))
        if out is None:
            out = torch.empty((N, D, H, W, 64), device=x.device, dtype=torch.float32)
        if wpack is None:
            wpack, _ = pack_conv64_weights(w, want_dgrad=False)
        check(_lib.load().fdn_conv64_fwd_mask(_p(x, "x"), _p(wpack, "wpack"), _p(bias, allow_none=True), _p(residual, allow_none=True),
                                              _p(out, "out"), _pm(mask), N, D, H, W, act, float(alpha), int(algo), _stream()),
              "fdn_conv64_fwd_mask")
        return out
    if out is None:
        out = torch.empty((N, D, H, W, Cout), device=x.device, dtype=torch.float32)
        ldy = Cout
    elif ldy is None:
        ldy = out.shape[-1]
    if Cin == 64 and Cout == 64 and K == 3 and wpack is None:
        wpack, _ = pack_conv64_weights(w, want_dgrad=False)
    check(_lib.load().fdn_conv3d_fwd(_p(x, "x"), _p(x2, allow_none=True), _p(w, "w"), _p(wpack, allow_none=True),
                                     _p(bias, allow_none=True), _p(residual, allow_none=True), _p(out, "out"),
                                     N, D, H, W, Cin, Cout, K, ldy, y_coff, act, float(alpha), int(algo), _stream()),
          "fdn_conv3d_fwd")
    return out


def conv3d_dgrad(dz, w, wpack_dgrad=None, out=None, lddz=None, dz_coff=0, spatial=None, algo=ALGO_AUTO):
    """Returns the gradient on the PADDED input grid (N,D+2,H+2,W+2,Cin) for K=3."""
    K, Cin, Cout = w.shape[0], w.shape[3], w.shape[4]
    N, D, H, W = dz.shape[:4] if spatial is None else spatial
    if lddz is None:
        lddz = dz.shape[-1]
    if out is None:
        out = torch.empty((N, D + 2, H + 2, W + 2, Cin), device=dz.device, dtype=torch.float32)
    if Cin == 64 and Cout == 64 and K == 3 and wpack_dgrad is None:
        _, wpack_dgrad = pack_conv64_weights(w)
    check(_lib.load().fdn_conv3d_dgrad(_p(dz, "dz"), _p(w, "w"), _p(wpack_dgrad, allow_none=True), _p(out, "dxpad"),
                                       N, D, H, W, Cin, Cout, K, lddz, dz_coff, int(algo), _stream()), "fdn_conv3d_dgrad")
    return out


def conv_cout1_dgrad_folded(dz, w, spatial, y_prev=None, act=ACT_NONE, alpha=LEAKY_ALPHA, lddz=1, dz_coff=0, out=None,
                            dbias_prev=None, workspace=None, mask=None):
    """64->1 head dgrad + halo fold + act'(y_prev) in one kernel.  Returns (N,D,H,W,64).  With dbias_prev (64 floats) it
    also emits the producing layer's bias gradient (sum of the result over voxels); needs a >= 512 KB workspace.
    mask: the sign mask conv3d_fwd(mask=...) wrote beside y_prev (W % 4 == 0), read instead of y_prev's rows."""
    N, D, H, W = spatial
    if out is None:
        out = torch.empty((N, D, H, W, 64), device=dz.device, dtype=torch.float32)
    if dbias_prev is not None and workspace is None:
        workspace = torch.empty(2048 * 64, device=dz.device, dtype=torch.float32)
    wsb = 0 if workspace is None else workspace.numel() * workspace.element_size()
    if mask is not None:
        if mask.numel() != 4 * N * D * H * W:
            raise FdnError("conv_cout1_dgrad_folded: a sign mask of 4 x %d int16 words expected" % (N * D * H * W))
        check(_lib.load().fdn_conv_cout1_dgrad_folded_mask(_p(dz, "dz"), _p(w, "w"), _pm(mask), act, float(alpha), _p(out),
                                                           _p(dbias_prev, allow_none=True), _p(workspace, allow_none=True), wsb,
                                                           N, D, H, W, lddz, dz_coff, _stream()), "fdn_conv_cout1_dgrad_folded_mask")
        return out
    check(_lib.load().fdn_conv_cout1_dgrad_folded(_p(dz, "dz"), _p(w, "w"), _p(y_prev, allow_none=True), act, float(alpha),
                                                  _p(out), _p(dbias_prev, allow_none=True), _p(workspace, allow_none=True), wsb,
                                                  N, D, H, W, lddz, dz_coff, _stream()), "fdn_conv_cout1_dgrad_folded")
    return out


def fold_halo(dxpads, skip=None, y_prev=None, act=ACT_NONE, alpha=LEAKY_ALPHA, out=None):
    """dxpads: 1..3 tensors (N,D+2,H+2,W+2,C).  Returns (N,D,H,W,C)."""
    p0 = dxpads[0]
    N, D, H, W, C = p0.shape[0], p0.shape[1] - 2, p0.shape[2] - 2, p0.shape[3] - 2, p0.shape[4]
    if out is None:
        out = torch.empty((N, D, H, W, C), device=p0.device, dtype=torch.float32)
    ptrs = [_p(t) for t in dxpads] + [None] * (3 - len(dxpads))
    check(_lib.load().fdn_fold_halo(ptrs[0], ptrs[1], ptrs[2], len(dxpads), _p(skip, allow_none=True),
                                    _p(y_prev, allow_none=True), act, float(alpha), _p(out), N, D, H, W, C, _stream()),
          "fdn_fold_halo")
    return out


DGRAD_INNER, DGRAD_SHELL = 1, 2


def conv3d_dgrad_fused(dz, wpack_dgrad, dxpad, out, skip=None, y_prev=None, act=ACT_NONE, alpha=LEAKY_ALPHA, parts=3,
                       algo=ALGO_AUTO, mask=None):
    """64->64 dgrad; interior voxels of `out` are finished in the conv epilogue (skip may alias out), the rest lands
    in the padded scratch `dxpad` for fold_halo_border.  parts: DGRAD_INNER | DGRAD_SHELL -- the two pieces write disjoint
    positions and may run on different streams.  mask: the sign mask the forward wrote beside y_prev (conv3d_fwd(mask=...)), read
    instead of y_prev for act' (the one-launch form only)."""
    N, D, H, W = dz.shape[:4]
    if mask is not None:
        if parts != 3 or mask.numel() != 4 * N * D * H * W:
            raise FdnError("conv3d_dgrad_fused: a sign mask needs the one-launch form and 4 x %d int16 words" % (N * D * H * W))
        check(_lib.load().fdn_conv64_dgrad_fused_mask(_p(dz, "dz"), _p(wpack_dgrad, "wpack"), _p(dxpad, "dxpad"), _p(skip, allow_none=True),
                                                      _pm(mask), act, float(alpha), _p(out, "out"), N, D, H, W, int(algo), _stream()),
              "fdn_conv64_dgrad_fused_mask")
        return out
    if parts == 3:
        check(_lib.load().fdn_conv3d_dgrad_fused(_p(dz, "dz"), _p(wpack_dgrad, "wpack"), _p(dxpad, "dxpad"),
                                                 _p(skip, allow_none=True), _p(y_prev, allow_none=True), act, float(alpha),
                                                 _p(out, "out"), N, D, H, W, int(algo), _stream()), "fdn_conv3d_dgrad_fused")
    else:
        check(_lib.load().fdn_conv3d_dgrad_fused_part(_p(dz, "dz"), _p(wpack_dgrad, "wpack"), _p(dxpad, "dxpad"),
                                                      _p(skip, allow_none=True), _p(y_prev, allow_none=True), act, float(alpha),
                                                      _p(out, "out"), N, D, H, W, int(parts), int(algo), _stream()),
              "fdn_conv3d_dgrad_fused_part")
    return out


def conv3d_dgrad_fused_multi(dzs, wpacks_dgrad, dxpad, out, skip=None, y_prev=None, act=ACT_NONE, alpha=LEAKY_ALPHA, algo=ALGO_AUTO, mask=None):
    """The fused dgrad of 1..3 64->64 layers that share their input, as ONE launch: out / dxpad receive what chained conv3d_dgrad_fused
    calls (skip = the running sum) would leave, to fp32 rounding -- the sum over the sources is formed in the kernel's registers.  Only
    where conv64_mask_ok(N, D, H, W, algo); the packs must be views of one pack buffer.  y_prev or mask (its sign mask) or neither."""
    n = len(dzs)
    N, D, H, W = dzs[0].shape[:4]
    if not 1 <= n <= 3 or len(wpacks_dgrad) != n or any(tuple(t.shape) != tuple(dzs[0].shape) for t in dzs):
        raise FdnError("conv3d_dgrad_fused_multi: 1..3 sources of one shape, one pack each")
    if mask is not None and (y_prev is not None or mask.numel() != 4 * N * D * H * W):
        raise FdnError("conv3d_dgrad_fused_multi: y_prev OR its sign mask of 4 x %d int16 words" % (N * D * H * W))
    tz = (ctypes.c_void_p * n)(*[_p(t, "dz") for t in dzs])
    tw = (ctypes.c_void_p * n)(*[_p(t, "wpack") for t in wpacks_dgrad])
    check(_lib.load().fdn_conv64_dgrad_fused_multi(tz, tw, n, _p(dxpad, "dxpad"), _p(skip, allow_none=True), _p(y_prev, allow_none=True),
                                                   None if mask is None else _pm(mask), act, float(alpha), _p(out, "out"), N, D, H, W, int(algo),
                                                   _stream()), "fdn_conv64_dgrad_fused_multi")
    return out


def fold_halo_border(dxpads, out, skip=None, y_prev=None, act=ACT_NONE, alpha=LEAKY_ALPHA):
    N, D, H, W = out.shape[:4]
    ptrs = [_p(t) for t in dxpads] + [None] * (3 - len(dxpads))
    check(_lib.load().fdn_fold_halo_border(ptrs[0], ptrs[1], ptrs[2], len(dxpads), _p(skip, allow_none=True),
                                           _p(y_prev, allow_none=True), act, float(alpha), _p(out), N, D, H, W, _stream()),
          "fdn_fold_halo_border")
    return out


def conv1x1_dgrad(dz, w, ya, yb, dxa=None, dxb=None):
    nvox = dz.numel() // 64
    if dxa is None:
        dxa = torch.empty_like(ya)
    if dxb is None:
        dxb = torch.empty_like(yb)
    check(_lib.load().fdn_conv1x1_dgrad(_p(dz), _p(w), _p(ya), _p(yb), _p(dxa), _p(dxb), nvox, _stream()),
          "fdn_conv1x1_dgrad")
    return dxa, dxb


def wgrad_workspace_bytes(N, D, H, W, Cin, Cout, K):
    return int(_lib.load().fdn_conv3d_wgrad_workspace_bytes(N, D, H, W, Cin, Cout, K))


def conv3d_wgrad(x, dz, K, Cin, Cout, x2=None, want_bias=False, dw=None, dbias=None, workspace=None, lddz=None,
                 dz_coff=0, algo=ALGO_AUTO):
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
    check(_lib.load().fdn_conv3d_wgrad(_p(x, "x"), _p(x2, allow_none=True), _p(dz, "dz"), _p(dw, "dw"),
                                       _p(dbias, allow_none=True), _p(workspace, "workspace"),
                                       workspace.numel() * workspace.element_size(), N, D, H, W, Cin, Cout, K, lddz,
                                       dz_coff, int(algo), _stream()), "fdn_conv3d_wgrad")
    return dw, dbias


def wgrad_batch_workspace_bytes(n_layers, N, D, H, W):
    return int(_lib.load().fdn_conv3d_wgrad_batch_workspace_bytes(n_layers, N, D, H, W))


def conv3d_wgrad_batch(xs, dzs, dws, dbiases=None, workspace=None, algo=ALGO_AUTO):
    """Weight gradients of several 64->64 3x3x3 layers that share one grid in ONE launch (fdn_conv3d_wgrad_batch): xs / dzs / dws are
    lists of tensors (N,D,H,W,64) / (N,D,H,W,64) / (3,3,3,64,64); dbiases: None or a list with None / (64,) entries."""
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
    table = lambda ts, name: (ctypes.c_void_p * n)(*[None if t is None else _p(t, name) for t in ts])
    tx, tz, tw = table(xs, "x"), table(dzs, "dz"), table(dws, "dw")
    tb = table(dbiases, "dbias") if dbiases is not None and any(b is not None for b in dbiases) else None
    check(_lib.load().fdn_conv3d_wgrad_batch(tx, tz, tw, tb, n, _p(workspace, "workspace"), workspace.numel() * workspace.element_size(),
                                            N, D, H, W, int(algo), _stream()), "fdn_conv3d_wgrad_batch")
    return dws


def upsample_trilinear_fwd(x, R, out=None):
    N, D, H, W, C = x.shape
    if out is None:
        out = torch.empty((N, D * R, H * R, W * R, C), device=x.device, dtype=torch.float32)
    check(_lib.load().fdn_upsample_trilinear_fwd(_p(x), _p(out), N, D, H, W, C, R, _stream()), "fdn_upsample_trilinear_fwd")
    return out


def upsample_trilinear_bwd(dy, R, y_prev=None, act=ACT_NONE, alpha=LEAKY_ALPHA, out=None):
    N, OD, OH, OW, C = dy.shape
    D, H, W = OD // R, OH // R, OW // R
    if out is None:
        out = torch.empty((N, D, H, W, C), device=dy.device, dtype=torch.float32)
    check(_lib.load().fdn_upsample_trilinear_bwd(_p(dy), _p(y_prev, allow_none=True), act, float(alpha), _p(out), N, D, H, W,
                                                 C, R, _stream()), "fdn_upsample_trilinear_bwd")
    return out


def loss_metrics(pred, uh, vh, wh, mask, want_grad=True, out=None, dpred=None, scratch=None):
    """Returns out (N,4) = [mse-loss, rel-err %, sum mask, sum nonfluid] and dpred (N,...,3) or None."""
    N = pred.shape[0]
    V = pred.numel() // (3 * N)
    if out is None:
        out = torch.empty((N, 4), device=pred.device, dtype=torch.float32)
    if scratch is None:
        scratch = torch.empty((N * (8 + 3 * 256),), device=pred.device, dtype=torch.float32)     # FDN_LOSS_SCRATCH_FLOATS(N)
    if want_grad and dpred is None:
        dpred = torch.empty_like(pred)
    check(_lib.load().fdn_loss_metrics(_p(pred), _p(uh), _p(vh), _p(wh), _p(mask), _p(out),
                                       _p(dpred, allow_none=True) if want_grad else None, _p(scratch), N, V, _stream()),
          "fdn_loss_metrics")
    return out, (dpred if want_grad else None)


def l2_sumsq(w_flat, is_kernel, out=None):
    if out is None:
        out = torch.empty((1,), device=w_flat.device, dtype=torch.float32)
    check(_lib.load().fdn_l2_sumsq(_p(w_flat), _p(is_kernel), w_flat.numel(), _p(out), _stream()), "fdn_l2_sumsq")
    return out


ADAM_PARTIALS = 2048                  # FDN_ADAM_PARTIALS


def adam_step(w, g, m, v, is_kernel, lr_t, b1, b2, eps, l2_grad_scale, l2_scale_dev=None, sumsq_partials=None):
    """sumsq_partials (ADAM_PARTIALS floats): also receive per-block sums of the updated kernel parameters' squares."""
    if sumsq_partials is not None and sumsq_partials.numel() < ADAM_PARTIALS:
        raise FdnError("adam_step: sumsq_partials needs %d floats" % ADAM_PARTIALS)
    check(_lib.load().fdn_adam_step(_p(w), _p(g), _p(m), _p(v), _p(is_kernel), w.numel(), float(lr_t), float(b1), float(b2),
                                    float(eps), float(l2_grad_scale), _p(l2_scale_dev, allow_none=True),
                                    _p(sumsq_partials, allow_none=True), _stream()),
          "fdn_adam_step")


def l2_sumsq_partials(w_flat, is_kernel, partials):
    """ADAM_PARTIALS per-block sums of the kernel parameters' squares (what adam_step leaves behind), for parameters no Adam step has touched."""
    if partials.numel() < ADAM_PARTIALS:
        raise FdnError("l2_sumsq_partials: partials needs %d floats" % ADAM_PARTIALS)
    check(_lib.load().fdn_l2_sumsq_partials(_p(w_flat), _p(is_kernel), w_flat.numel(), _p(partials), _stream()), "fdn_l2_sumsq_partials")
    return partials


def sum_partials(partials, out=None):
    if out is None:
        out = torch.empty((1,), device=partials.device, dtype=torch.float32)
    check(_lib.load().fdn_sum_partials(_p(partials), partials.numel(), _p(out), _stream()), "fdn_sum_partials")
    return out


def pack_conv64_weights_batch(w_flat, w_offsets, packs, streams=None):
    """Every 64->64 kernel of the flat parameter buffer in one launch.  w_offsets: int64 DEVICE tensor of float offsets;
    packs: (n_layers, 2, CONV64_PACK_FLOATS).  streams = (forward mask, dgrad mask) of PACK_STREAM_* bits restricts the launch to
    those streams of the packs (conv64_pack_streams names the ones a grid reads); None = all."""
    n = w_offsets.numel()
    if w_offsets.dtype != torch.int64 or not w_offsets.is_cuda or packs.numel() != n * 2 * CONV64_PACK_FLOATS:
        raise FdnError("pack_conv64_weights_batch: bad offsets / packs")
    sf, sd = (PACK_STREAM_ALL, PACK_STREAM_ALL) if streams is None else streams
    check(_lib.load().fdn_pack_conv64_weights_batch_streams(_p(w_flat), w_offsets.data_ptr(), n, _p(packs), int(sf), int(sd), _stream()),
          "fdn_pack_conv64_weights_batch_streams")
    return packs


PACK_STREAM_DIRECT, PACK_STREAM_WINO_W, PACK_STREAM_WINO_H2, PACK_STREAM_WINO_H4, PACK_STREAM_WINO_H4S = 1, 2, 4, 8, 16
PACK_STREAM_ALL = 31
ROLE_FWD, ROLE_DGRAD, ROLE_DGRAD_FUSED = 0, 1, 2


def conv64_pack_streams(N, D, H, W, algo=ALGO_AUTO, role=ROLE_FWD):
    """PACK_STREAM_* bits of the pack streams the 64->64 entry point of `role` reads on an (N,D,H,W) grid (fdn_conv64_pack_streams)."""
    m = _lib.load().fdn_conv64_pack_streams(int(N), int(D), int(H), int(W), int(algo), int(role))
    check(min(m, 0), "fdn_conv64_pack_streams")
    return m

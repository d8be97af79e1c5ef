"""CPU oracle for the 4DFlowNet hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product path (4dflownet_amd/) never does; it fails loudly when
the HIP library is missing.

PARITY UNPINNED: the arithmetic of the reference lives in TensorFlow 2.2 / Keras
(un-vendored third-party dependency named only in /root/reference/README.md:4).
TensorFlow is not installable here, and the reference ships no tests, golden
vectors or saved activations for this path.  This file
restates the published semantics of the TF ops the reference calls, anchored on
the reference's own call sites (cited per function), and is cross-checked
against torch-CPU autograd in tests/test_oracle.py as an independent second
opinion on the math.  The loader / tiler parts of the path ARE pinned against
the reference itself (tests/golden/make_golden.py imports the reference's
TF-free modules).

All tensors are NDHWC numpy arrays.  `dtype` selects float64 (checking) or
float32 (CPU-baseline timing).
"""
import numpy as np

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2
L2_LAMBDA = 5e-7        # SR4DFlowNet.py:99  tf.keras.regularizers.l2(5e-7)
LEAKY_ALPHA = 0.2       # SR4DFlowNet.py:113,118
ADAM_B1, ADAM_B2, ADAM_EPS = 0.9, 0.999, 1e-7   # tf.keras.optimizers.Adam defaults (TrainerController.py:73)


# ----------------------------------------------------------------------------
# elementwise helpers
# ----------------------------------------------------------------------------
def bf16_round(a):
    """Round to the nearest bfloat16 (ties to even), returned in the input's dtype.  Models the store of an
    activation tensor in the bf16 variant of the path (BASELINE.json configs[3]; no reference counterpart)."""
    a = np.asarray(a)
    f = np.ascontiguousarray(a, dtype=np.float32)
    u = f.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16 << 16).astype(np.uint32)
    out = r.view(np.float32).reshape(f.shape)
    out = np.where(np.isfinite(f), out, f)
    return out.astype(a.dtype if a.dtype.kind == "f" else np.float32)


def act_fwd(z, act, alpha=LEAKY_ALPHA):
    if act == ACT_NONE:
        return z
    if act == ACT_RELU:
        return np.maximum(z, 0)
    if act == ACT_LEAKY:
        return np.where(z > 0, z, alpha * z)
    raise ValueError(act)


def act_bwd_from_output(dy, y, act, alpha=LEAKY_ALPHA):
    """dz = dy * act'(z), with act'(z) recovered from y = act(z) (y>0 <=> z>0)."""
    if act == ACT_NONE:
        return dy
    if act == ACT_RELU:
        return dy * (y > 0)
    if act == ACT_LEAKY:
        return dy * np.where(y > 0, 1.0, alpha).astype(dy.dtype)
    raise ValueError(act)


def input_features(u, v, w, mu, mv, mw):
    """SR4DFlowNet.py:10-15.  Inputs (N,D,H,W,1) -> phase (N,D,H,W,3), pc (N,D,H,W,3)."""
    speed = (u ** 2 + v ** 2 + w ** 2) ** 0.5
    mag = (mu ** 2 + mv ** 2 + mw ** 2) ** 0.5
    pcmr = mag * speed
    phase = np.concatenate([u, v, w], axis=-1)
    pc = np.concatenate([pcmr, mag, speed], axis=-1)
    return phase, pc


# ----------------------------------------------------------------------------
# conv3d: tf.pad(SYMMETRIC, p=(k-1)//2) + Conv3D(valid)   (SR4DFlowNet.py:93-108)
# SYMMETRIC with p=1 repeats the edge voxel == numpy 'edge' == clamp indexing.
# Kernel layout (kd,kh,kw,Cin,Cout), cross-correlation (Keras Conv3D).
# ----------------------------------------------------------------------------
def _pad_edge(x, p):
    if p == 0:
        return x
    return np.pad(x, ((0, 0), (p, p), (p, p), (p, p), (0, 0)), mode="symmetric")


def conv3d_linear(x, w):
    k = w.shape[0]
    p = (k - 1) // 2
    xp = _pad_edge(x, p)
    N, D, H, W, Cin = x.shape
    Cout = w.shape[-1]
    y = np.zeros((N * D * H * W, Cout), dtype=x.dtype)
    for a in range(k):
        for b in range(k):
            for c in range(k):
                patch = xp[:, a:a + D, b:b + H, c:c + W, :].reshape(-1, Cin)
                y += patch @ w[a, b, c]
    return y.reshape(N, D, H, W, Cout)


def conv3d_fwd(x, w, bias=None, act=ACT_NONE, alpha=LEAKY_ALPHA, residual=None):
    """y = act(conv(x) + bias + residual).  residual models resnet_block's `x + tmp` (SR4DFlowNet.py:117)."""
    z = conv3d_linear(x, w)
    if bias is not None:
        z = z + bias
    if residual is not None:
        z = z + residual
    return act_fwd(z, act, alpha)


def conv3d_dgrad(dz, w, in_shape):
    """Adjoint of conv3d_linear w.r.t. x.  = Conv3DBackpropInput + MirrorPadGrad:
    full correlation onto the padded grid, then fold the replicated halo back
    onto the edge voxels (corners receive 8 contributions)."""
    k = w.shape[0]
    p = (k - 1) // 2
    N, D, H, W, Cin = in_shape
    Cout = w.shape[-1]
    dxp = np.zeros((N, D + 2 * p, H + 2 * p, W + 2 * p, Cin), dtype=dz.dtype)
    dz2 = dz.reshape(-1, Cout)
    for a in range(k):
        for b in range(k):
            for c in range(k):
                dxp[:, a:a + D, b:b + H, c:c + W, :] += (dz2 @ w[a, b, c].T).reshape(N, D, H, W, Cin)
    return fold_halo(dxp, p)


def fold_halo(dxp, p):
    """Adjoint of edge-replicate padding by p (MirrorPadGrad, SYMMETRIC, p<=1)."""
    if p == 0:
        return dxp
    assert p == 1
    for ax in (1, 2, 3):
        n = dxp.shape[ax]
        core = np.take(dxp, np.arange(1, n - 1), axis=ax).copy()
        lo = np.take(dxp, [0], axis=ax)
        hi = np.take(dxp, [n - 1], axis=ax)
        idx_lo = [slice(None)] * 5
        idx_hi = [slice(None)] * 5
        idx_lo[ax] = slice(0, 1)
        idx_hi[ax] = slice(core.shape[ax] - 1, core.shape[ax])
        core[tuple(idx_lo)] += lo
        core[tuple(idx_hi)] += hi
        dxp = core
    return dxp


def conv3d_wgrad(x, dz, k):
    """dW[a,b,c,ci,co] = sum_{n,o} xpad[n,o+(a,b,c),ci] * dz[n,o,co]   (Conv3DBackpropFilter)."""
    p = (k - 1) // 2
    xp = _pad_edge(x, p)
    N, D, H, W, Cin = x.shape
    Cout = dz.shape[-1]
    dz2 = dz.reshape(-1, Cout)
    dw = np.zeros((k, k, k, Cin, Cout), dtype=x.dtype)
    for a in range(k):
        for b in range(k):
            for c in range(k):
                patch = xp[:, a:a + D, b:b + H, c:c + W, :].reshape(-1, Cin)
                dw[a, b, c] = patch.T @ dz2
    return dw


def bias_grad(dz):
    return dz.reshape(-1, dz.shape[-1]).sum(axis=0)


# ----------------------------------------------------------------------------
# upsample3d (SR4DFlowNet.py:53-90): two tf.compat.v1.image.resize_bilinear
# (align_corners=True) passes + transposes == separable trilinear with
# src = i * (n-1)/(n*R-1), lo = floor(src), hi = min(lo+1, n-1).
# ----------------------------------------------------------------------------
def _lin_weights(n, R, dtype):
    m = n * R
    scale = (n - 1) / (m - 1) if m > 1 else 0.0
    src = np.arange(m, dtype=np.float64) * scale
    lo = np.floor(src).astype(np.int64)
    hi = np.minimum(lo + 1, n - 1)
    frac = (src - lo).astype(dtype)
    return lo, hi, frac


def lerp_coeffs_f32(n, R):
    """The float32 coefficient convention the HIP kernel uses (TF resize_bilinear
    computes `in = out * scale` in float32: scale = (n-1)/(nR-1) as float)."""
    m = n * R
    scale = np.float32(n - 1) / np.float32(m - 1) if m > 1 else np.float32(0)
    src = np.arange(m, dtype=np.float32) * scale
    lo = np.floor(src).astype(np.int64)
    hi = np.minimum(lo + 1, n - 1)
    frac = (src - lo.astype(np.float32)).astype(np.float32)
    return lo, hi, frac


def _resize_axis(x, axis, R, f32_coeffs=False):
    n = x.shape[axis]
    lo, hi, frac = lerp_coeffs_f32(n, R) if f32_coeffs else _lin_weights(n, R, x.dtype)
    frac = frac.astype(x.dtype)
    shp = [1] * x.ndim
    shp[axis] = -1
    f = frac.reshape(shp)
    a = np.take(x, lo, axis=axis)
    b = np.take(x, hi, axis=axis)
    return a + (b - a) * f           # TF: top + (bottom - top) * lerp


def upsample_trilinear_fwd(x, R, f32_coeffs=False):
    if R == 1:
        return x
    # reference order: (y,z) resize first, then x.  Separable, so order only affects rounding.
    y = _resize_axis(x, 2, R, f32_coeffs)
    y = _resize_axis(y, 3, R, f32_coeffs)
    y = _resize_axis(y, 1, R, f32_coeffs)
    return y


def _resize_axis_adjoint(dy, axis, n, R, f32_coeffs=False):
    lo, hi, frac = lerp_coeffs_f32(n, R) if f32_coeffs else _lin_weights(n, R, dy.dtype)
    frac = frac.astype(dy.dtype)
    shp = [1] * dy.ndim
    shp[axis] = -1
    f = frac.reshape(shp)
    out_shape = list(dy.shape)
    out_shape[axis] = n
    dx = np.zeros(out_shape, dtype=dy.dtype)
    dyt = np.moveaxis(dy, axis, 0)
    ft = np.moveaxis(f, axis, 0)
    dxt = np.moveaxis(dx, axis, 0)
    np.add.at(dxt, lo, dyt * (1 - ft))
    np.add.at(dxt, hi, dyt * ft)
    return dx


def upsample_trilinear_bwd(dy, in_spatial, R, f32_coeffs=False):
    if R == 1:
        return dy
    D, H, W = in_spatial
    d = _resize_axis_adjoint(dy, 1, D, R, f32_coeffs)
    d = _resize_axis_adjoint(d, 3, W, R, f32_coeffs)
    d = _resize_axis_adjoint(d, 2, H, R, f32_coeffs)
    return d


# ----------------------------------------------------------------------------
# loss / metric   (TrainerController.py:84-127,152-156; loss_utils.py:64-103)
# ----------------------------------------------------------------------------
def masked_mse_loss_fwd_bwd(pred, target, mask):
    """pred/target (N,D,H,W,3), mask (N,D,H,W) -> loss (N,), dpred (N,D,H,W,3).
    loss_b = sum(mse*mask)/(sum(mask)+1) + sum(mse*nf)/(sum(nf)+1), mse = sum_c (p-t)^2, nf = mask<0.5.
    dpred is the gradient of sum_b loss_b (tape.gradient of a vector sums it, TrainerController.py:223)."""
    diff = pred - target
    mse = (diff ** 2).sum(axis=-1)
    nf = (mask < 0.5).astype(pred.dtype)
    sm = mask.sum(axis=(1, 2, 3))
    snf = nf.sum(axis=(1, 2, 3))
    fluid = (mse * mask).sum(axis=(1, 2, 3)) / (sm + 1)
    nonfluid = (mse * nf).sum(axis=(1, 2, 3)) / (snf + 1)
    loss = fluid + nonfluid
    wgt = mask / (sm + 1)[:, None, None, None] + nf / (snf + 1)[:, None, None, None]
    dpred = 2 * diff * wgt[..., None]
    return loss, dpred


def relative_error(pred, target, mask):
    """loss_utils.calculate_relative_error (loss_utils.py:64-103).  -> (N,) percent.
    np.round is round-half-to-even, like tf.round."""
    eps = 1e-5
    diff = np.sqrt(((pred - target) ** 2).sum(axis=-1))
    actual = np.sqrt((target ** 2).sum(axis=-1))
    rel = np.clip(diff / (actual + eps), 0.0, 1.0)
    corr = np.where(actual != 0, rel, diff)
    corr = np.round(corr * 1e4) / 1e4
    corr = np.where(mask == 1.0, corr, 0.0)
    return corr.sum(axis=(1, 2, 3)) / (mask.sum(axis=(1, 2, 3)) + 1) * 100


def l2_regularizer(params):
    """TrainerController.py:129-141: sum over conv kernels of 5e-7 * sum(w^2); biases are not regularised."""
    return sum(L2_LAMBDA * float((p["w"].astype(np.float64) ** 2).sum()) for p in params)


# ----------------------------------------------------------------------------
# parameters: Keras creation order conv3d, conv3d_1 ... (SR4DFlowNet.py:17-46)
# ----------------------------------------------------------------------------
def layer_specs(low_resblock=8, hi_resblock=4):
    """[(name, k, cin, cout, use_bias)] in Keras creation order."""
    specs = []

    def add(k, cin, cout, bias):
        i = len(specs)
        specs.append(("conv3d" if i == 0 else "conv3d_%d" % i, k, cin, cout, bias))

    add(3, 3, 64, True); add(3, 64, 64, True)          # pc path      :17-18
    add(3, 3, 64, True); add(3, 64, 64, True)          # phase path   :20-21
    add(1, 128, 64, True); add(3, 64, 64, True)        # fuse         :24-25
    for _ in range(low_resblock):                      # :29-30
        add(3, 64, 64, False); add(3, 64, 64, False)
    for _ in range(hi_resblock):                       # :35-36
        add(3, 64, 64, False); add(3, 64, 64, False)
    for _ in range(3):                                 # heads u,v,w :39-46
        add(3, 64, 64, True); add(3, 64, 1, True)
    return specs


def glorot_uniform(rng, k, cin, cout, dtype=np.float32):
    """Keras GlorotUniform: limit = sqrt(6/(fan_in+fan_out)), fan_in = k^3*cin, fan_out = k^3*cout."""
    fan_in, fan_out = k ** 3 * cin, k ** 3 * cout
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=(k, k, k, cin, cout)).astype(dtype)


def init_params(seed=0, low_resblock=8, hi_resblock=4, dtype=np.float32):
    rng = np.random.default_rng(seed)
    params = []
    for name, k, cin, cout, use_bias in layer_specs(low_resblock, hi_resblock):
        p = {"name": name, "k": k, "w": glorot_uniform(rng, k, cin, cout, dtype)}
        p["b"] = np.zeros(cout, dtype=dtype) if use_bias else None
        params.append(p)
    return params


def count_params(params):
    return sum(p["w"].size + (p["b"].size if p["b"] is not None else 0) for p in params)


# ----------------------------------------------------------------------------
# network forward / backward  (SR4DFlowNet.build_network, SR4DFlowNet.py:7-51)
# ----------------------------------------------------------------------------
def _bf16_hooks(bf16):
    """(r, wq): activation-store rounding and the weight view of a layer under the bf16 variant of the path
    (BASELINE.json configs[3]).  Rounding points = where 4dflownet_amd stores a bf16 tensor: every 64-channel
    activation / activation gradient, the 3-channel input features, the upsample output; the MFMA operand copy of the
    64->64 kernels.  Everything else (thin-layer weights, prediction, parameter gradients) stays full precision."""
    if not bf16:
        return (lambda a: a), (lambda w: w)
    return bf16_round, (lambda w: bf16_round(w) if w.shape == (3, 3, 3, 64, 64) else w)


def network_forward(params, inputs, res_increase, low_resblock=8, hi_resblock=4, f32_coeffs=False, bf16=False):
    """inputs: 6 arrays (N,P,P,P,1).  Returns pred (N,PR,PR,PR,3) and a cache for backward."""
    r, wq = _bf16_hooks(bf16)
    u, v, w, mu, mv, mw = inputs
    phase, pc = input_features(u, v, w, mu, mv, mw)
    phase, pc = r(phase), r(pc)
    P = params
    c = {"phase": phase, "pc": pc}
    c["a0"] = r(conv3d_fwd(pc, P[0]["w"], P[0]["b"], ACT_RELU))
    c["a1"] = r(conv3d_fwd(c["a0"], wq(P[1]["w"]), P[1]["b"], ACT_RELU))
    c["p0"] = r(conv3d_fwd(phase, P[2]["w"], P[2]["b"], ACT_RELU))
    c["p1"] = r(conv3d_fwd(c["p0"], wq(P[3]["w"]), P[3]["b"], ACT_RELU))
    c["cat"] = np.concatenate([c["p1"], c["a1"]], axis=-1)          # :23  [phase, pc]
    c["c0"] = r(conv3d_fwd(c["cat"], P[4]["w"], P[4]["b"], ACT_RELU))
    c["c1"] = r(conv3d_fwd(c["c0"], wq(P[5]["w"]), P[5]["b"], ACT_RELU))
    rb = c["c1"]
    li = 6
    c["blocks"] = []
    for i in range(low_resblock + hi_resblock):
        if i == low_resblock:
            c["up_in"] = rb
            rb = r(upsample_trilinear_fwd(rb, res_increase, f32_coeffs))
            c["up_out"] = rb
        h = r(conv3d_fwd(rb, wq(P[li]["w"]), None, ACT_LEAKY))
        out = r(conv3d_fwd(h, wq(P[li + 1]["w"]), None, ACT_LEAKY, residual=rb))
        c["blocks"].append((rb, h, out))
        rb = out
        li += 2
    if hi_resblock == 0:
        c["up_in"] = rb
        rb = r(upsample_trilinear_fwd(rb, res_increase, f32_coeffs))
        c["up_out"] = rb
    c["rb"] = rb
    outs = []
    c["heads"] = []
    for hidx in range(3):
        g = r(conv3d_fwd(rb, wq(P[li]["w"]), P[li]["b"], ACT_RELU))
        o = conv3d_fwd(g, P[li + 1]["w"], P[li + 1]["b"], ACT_NONE)
        c["heads"].append(g)
        outs.append(o)
        li += 2
    pred = np.concatenate(outs, axis=-1)
    return pred, c


def network_backward(params, c, dpred, res_increase, low_resblock=8, hi_resblock=4, f32_coeffs=False, bf16=False, sides=None):
    """Gradients of sum_b loss_b w.r.t. every parameter, given dpred.  Returns list of {"w","b"}.
    sides (optional): {"a0","a1","p0","p1","c0","c1": bool array, "blocks": [(h, out) bool arrays], "heads": [bool arrays]} -- the side of
    the ReLU / LeakyReLU kink (True = positive) ANOTHER evaluation of the same forward landed on, unit by unit.  The network is piecewise
    linear; given `sides`, the backward pass differentiates in that evaluation's linear region instead of this forward's own (a unit
    within rounding of 0 may sit on either side, and one such unit moves single gradient elements by 1e-4..1e-3 of their scale), so
    that what separates the two gradients is arithmetic error only.  The caller checks that the units which changed side are few and tiny."""
    r, wq = _bf16_hooks(bf16)
    P = params
    if sides is not None:
        sgn = lambda m: np.where(np.asarray(m), 1.0, -1.0)           # act_bwd_from_output only looks at y > 0
        c = dict(c)
        for k in ("a0", "a1", "p0", "p1", "c0", "c1"):
            c["side_" + k] = sgn(sides[k])
        c["side_blocks"] = [(sgn(h), sgn(o)) for h, o in sides["blocks"]]
        c["side_heads"] = [sgn(g) for g in sides["heads"]]
    side = lambda key, val, idx=None, j=None: (val if sides is None else
                                               (c["side_" + key] if idx is None else (c["side_" + key][idx] if j is None else c["side_" + key][idx][j])))
    G = [{"w": None, "b": None} for _ in P]
    rb = c["rb"]
    li = len(P) - 6
    d_rb = np.zeros_like(rb)
    for hidx in range(3):
        g = c["heads"][hidx]
        dz_o = dpred[..., hidx:hidx + 1]
        G[li + 1]["w"] = conv3d_wgrad(g, dz_o, 3)
        G[li + 1]["b"] = bias_grad(dz_o)
        dg = conv3d_dgrad(dz_o, P[li + 1]["w"], g.shape)
        dz_g = act_bwd_from_output(dg, side("heads", g, hidx), ACT_RELU)
        G[li]["b"] = bias_grad(dz_g)                 # the HIP path sums this before the store rounds it
        dz_g = r(dz_g)
        G[li]["w"] = conv3d_wgrad(rb, dz_g, 3)
        d_rb = d_rb + conv3d_dgrad(dz_g, wq(P[li]["w"]), rb.shape)
        if hidx < 2:
            d_rb = r(d_rb)            # fan-in chained through a stored tensor; the third add is rounded after act'
        li += 2
    li = len(P) - 6
    d_out = d_rb                      # gradient w.r.t. block output (post activation)
    nblocks = low_resblock + hi_resblock
    if hi_resblock == 0:
        d_out = upsample_trilinear_bwd(r(d_out), c["up_in"].shape[1:4], res_increase, f32_coeffs)
    for i in reversed(range(nblocks)):
        x, h, out = c["blocks"][i]
        li -= 2
        dz_out = r(act_bwd_from_output(d_out, side("blocks", out, i, 1), ACT_LEAKY))
        G[li + 1]["w"] = conv3d_wgrad(h, dz_out, 3)
        dh = conv3d_dgrad(dz_out, wq(P[li + 1]["w"]), h.shape)
        dz_h = r(act_bwd_from_output(dh, side("blocks", h, i, 0), ACT_LEAKY))
        G[li]["w"] = conv3d_wgrad(x, dz_h, 3)
        d_out = conv3d_dgrad(dz_h, wq(P[li]["w"]), x.shape) + dz_out
        if i == low_resblock:
            d_out = r(d_out)          # stored (producer of this block input is the linear upsample) before U^T
            d_out = upsample_trilinear_bwd(d_out, c["up_in"].shape[1:4], res_increase, f32_coeffs)
    assert li == 6
    dz_c1 = r(act_bwd_from_output(d_out, side("c1", c["c1"]), ACT_RELU))
    G[5]["w"] = conv3d_wgrad(c["c0"], dz_c1, 3); G[5]["b"] = bias_grad(dz_c1)
    dz_c0 = r(act_bwd_from_output(conv3d_dgrad(dz_c1, wq(P[5]["w"]), c["c0"].shape), side("c0", c["c0"]), ACT_RELU))
    G[4]["w"] = conv3d_wgrad(c["cat"], dz_c0, 1); G[4]["b"] = bias_grad(dz_c0)
    dcat = conv3d_dgrad(dz_c0, P[4]["w"], c["cat"].shape)
    dz_p1 = r(act_bwd_from_output(dcat[..., :64], side("p1", c["p1"]), ACT_RELU))
    dz_a1 = r(act_bwd_from_output(dcat[..., 64:], side("a1", c["a1"]), ACT_RELU))
    G[3]["w"] = conv3d_wgrad(c["p0"], dz_p1, 3); G[3]["b"] = bias_grad(dz_p1)
    dz_p0 = r(act_bwd_from_output(conv3d_dgrad(dz_p1, wq(P[3]["w"]), c["p0"].shape), side("p0", c["p0"]), ACT_RELU))
    G[2]["w"] = conv3d_wgrad(c["phase"], dz_p0, 3); G[2]["b"] = bias_grad(dz_p0)
    G[1]["w"] = conv3d_wgrad(c["a0"], dz_a1, 3); G[1]["b"] = bias_grad(dz_a1)
    dz_a0 = r(act_bwd_from_output(conv3d_dgrad(dz_a1, wq(P[1]["w"]), c["a0"].shape), side("a0", c["a0"]), ACT_RELU))
    G[0]["w"] = conv3d_wgrad(c["pc"], dz_a0, 3); G[0]["b"] = bias_grad(dz_a0)
    for g, p in zip(G, P):
        if p["b"] is None:
            g["b"] = None
    return G


# ----------------------------------------------------------------------------
# train step  (TrainerController.train_step, TrainerController.py:209-225)
# ----------------------------------------------------------------------------
def flatten(plist, key_w="w", key_b="b"):
    """Flat vector in trainable_variables order: kernel, bias per layer in creation order."""
    parts = []
    for p in plist:
        parts.append(p[key_w].reshape(-1))
        if p.get(key_b) is not None:
            parts.append(p[key_b].reshape(-1))
    return np.concatenate(parts)


def adam_step_tf(w, g, m, v, t, lr, b1=ADAM_B1, b2=ADAM_B2, eps=ADAM_EPS):
    """Keras Adam (non-amsgrad) dense update: lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    m,v EMA; w -= lr_t*m/(sqrt(v)+eps)  (epsilon OUTSIDE the bias correction)."""
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    m[:] = b1 * m + (1 - b1) * g
    v[:] = b2 * v + (1 - b2) * g * g
    w[:] = w - lr_t * m / (np.sqrt(v) + eps)


def loss_and_grads(params, batch, res_increase, low_resblock=8, hi_resblock=4, f32_coeffs=False, bf16=False, sides=None):
    """batch = (u,v,w,u_mag,v_mag,w_mag,u_hr,v_hr,w_hr,venc,mask) as the loader yields them
    (PatchHandler3D.py:78-81).  Returns dict with per-sample loss (incl. L2), mse, rel-error,
    l2 scalar, pred and the gradient list of sum_b(loss_b) = sum_b mse_b + B*L2."""
    u, v, w, mu, mv, mw, uh, vh, wh, venc, mask = batch
    hires = np.concatenate([uh, vh, wh], axis=-1)
    pred, cache = network_forward(params, (u, v, w, mu, mv, mw), res_increase, low_resblock, hi_resblock, f32_coeffs, bf16)
    mse, dpred = masked_mse_loss_fwd_bwd(pred, hires, mask)
    rel = relative_error(pred, hires, mask)
    l2 = l2_regularizer(params)
    if callable(sides):                 # sides(cache) -> sides: lets the caller compare its own forward with this one's cache (one oracle forward)
        sides = sides(cache)
    grads = network_backward(params, cache, dpred, res_increase, low_resblock, hi_resblock, f32_coeffs, bf16, sides=sides)
    B = u.shape[0]
    for g, p in zip(grads, params):
        g["w"] = g["w"] + (B * 2 * L2_LAMBDA) * p["w"]
    return {"loss": mse + l2, "mse": mse, "rel_err": rel, "l2": l2, "pred": pred, "grads": grads}


def train_step(params, state, batch, lr, res_increase, low_resblock=8, hi_resblock=4, f32_coeffs=False, bf16=False):
    """One TrainerController.train_step.  state = {"t":int, "m":[...], "v":[...]} (created on first call)."""
    out = loss_and_grads(params, batch, res_increase, low_resblock, hi_resblock, f32_coeffs, bf16)
    if not state:
        state["t"] = 0
        state["m"] = [{"w": np.zeros_like(p["w"]), "b": None if p["b"] is None else np.zeros_like(p["b"])} for p in params]
        state["v"] = [{"w": np.zeros_like(p["w"]), "b": None if p["b"] is None else np.zeros_like(p["b"])} for p in params]
    state["t"] += 1
    for p, g, m, v in zip(params, out["grads"], state["m"], state["v"]):
        adam_step_tf(p["w"], g["w"], m["w"], v["w"], state["t"], lr)
        if p["b"] is not None:
            adam_step_tf(p["b"], g["b"], m["b"], v["b"], state["t"], lr)
    return out


# ----------------------------------------------------------------------------
# synthetic inputs
# ----------------------------------------------------------------------------
def synthetic_batch(B, P, R, seed=1234, dtype=np.float32):
    rng = np.random.default_rng(seed)
    lr = lambda lo, hi: rng.uniform(lo, hi, size=(B, P, P, P, 1)).astype(dtype)
    u, v, w = lr(-1, 1), lr(-1, 1), lr(-1, 1)
    mu, mv, mw = lr(0, 0.016), lr(0, 0.016), lr(0, 0.016)
    H = P * R
    hr = lambda: rng.uniform(-0.45, 0.45, size=(B, H, H, H, 1)).astype(dtype)
    uh, vh, wh = hr(), hr(), hr()
    mask = (rng.uniform(size=(B, H, H, H)) < 0.12).astype(dtype)
    venc = np.full((B,), 1.5, dtype=dtype)
    return (u, v, w, mu, mv, mw, uh, vh, wh, venc, mask)

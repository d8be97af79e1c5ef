"""TEST / BASELINE INFRASTRUCTURE -- not part of the product path (only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline may import anything under oracle/).

The 4DFlowNet train step written with stock torch-CPU operators (oneDNN conv3d + autograd): the graph of
src/Network/SR4DFlowNet.py:7-120, the loss of src/Network/TrainerController.py:84-127,245-249 and Keras-Adam
(:73,225).  Two uses:
  * float64: an independent second opinion on the numpy oracle's math (tests/test_oracle.py);
  * float32 on all host cores: the FAIR CPU baseline bench.py reports next to the GPU number -- the reference's own
    TensorFlow CPU path cannot run here (TensorFlow absent), and a vendor-tuned conv3d is the closest
    stand-in for what TF would dispatch to (MKL/oneDNN)."""
import time

import numpy as np
import torch
import torch.nn.functional as F

from . import flownet_oracle as O


def t_conv(x, w, b=None):
    """tf.pad SYMMETRIC p=(k-1)//2 (== replicate for p=1) + valid conv, NDHWC in/out.  SR4DFlowNet.py:93-108."""
    k = w.shape[0]
    p = (k - 1) // 2
    xt = x.permute(0, 4, 1, 2, 3)
    if p:
        xt = F.pad(xt, (p,) * 6, mode="replicate")
    wt = w.permute(4, 3, 0, 1, 2)
    y = F.conv3d(xt, wt, b)
    return y.permute(0, 2, 3, 4, 1)


def _up(x, R):
    return F.interpolate(x.permute(0, 4, 1, 2, 3), scale_factor=R, mode="trilinear", align_corners=True).permute(0, 2, 3, 4, 1)


class _ActWithMask(torch.autograd.Function):
    """relu / leaky-relu whose BACKWARD takes the side of the kink from a given mask (True = positive side) instead of the sign of its
    own input.  The piecewise-linear network is differentiated in the linear region another evaluation (the fp32 GPU forward) landed in:
    a unit whose pre-activation is within rounding of 0 may sit on either side, and a single such unit moves gradient elements by
    1e-4..1e-3 of their scale -- with the mask given, what is left between the two gradients is arithmetic error only."""

    @staticmethod
    def forward(ctx, z, slope, mask):
        ctx.slope = slope
        ctx.save_for_backward(mask)
        return torch.where(z > 0, z, z * slope)

    @staticmethod
    def backward(ctx, gy):
        (m,) = ctx.saved_tensors
        return torch.where(m, gy, gy * ctx.slope), None, None


def t_forward(params, inputs, R, LB, HB, masks=None, zs=None):
    """params: [(w, b_or_None)] in creation order.  SR4DFlowNet.py:7-51.
    masks (optional): one bool tensor per activation, in call order (the 6 stem convs, (h, out) of every ResBlock, the 3 head convs) --
    the side of the kink the backward pass differentiates on (_ActWithMask); zs (optional list): receives every pre-activation."""
    if masks is not None or zs is not None:
        it = iter(masks) if masks is not None else None

        def act(z, slope):
            if zs is not None:
                zs.append(z.detach())
            return _ActWithMask.apply(z, slope, next(it) if it is not None else z > 0)
        relu = lambda z: act(z, 0.0)
        leaky = lambda z, s: act(z, s)
    else:
        relu, leaky = F.relu, F.leaky_relu
    u, v, w, mu, mv, mw = inputs
    speed = (u ** 2 + v ** 2 + w ** 2) ** 0.5
    mag = (mu ** 2 + mv ** 2 + mw ** 2) ** 0.5
    pcmr = mag * speed
    phase = torch.cat([u, v, w], -1)
    pc = torch.cat([pcmr, mag, speed], -1)
    P = params
    pc = relu(t_conv(pc, *P[0])); pc = relu(t_conv(pc, *P[1]))
    ph = relu(t_conv(phase, *P[2])); ph = relu(t_conv(ph, *P[3]))
    x = relu(t_conv(torch.cat([ph, pc], -1), *P[4]))
    x = relu(t_conv(x, *P[5]))
    li = 6
    for i in range(LB + HB):
        if i == LB and R > 1:
            x = _up(x, R)
        h = leaky(t_conv(x, P[li][0]), 0.2)
        x = leaky(x + t_conv(h, P[li + 1][0]), 0.2)
        li += 2
    if HB == 0 and R > 1:
        x = _up(x, R)
    outs = []
    for _ in range(3):
        g = relu(t_conv(x, *P[li]))
        outs.append(t_conv(g, *P[li + 1]))
        li += 2
    return torch.cat(outs, -1)


def t_loss(pred, hires, mask):
    """TrainerController.py:84-107,152-156 -> (B,)."""
    mse = ((pred - hires) ** 2).sum(-1)
    nf = (mask < 0.5).to(pred.dtype)
    fluid = (mse * mask).sum((1, 2, 3)) / (mask.sum((1, 2, 3)) + 1)
    nonfluid = (mse * nf).sum((1, 2, 3)) / (nf.sum((1, 2, 3)) + 1)
    return fluid + nonfluid


def to_torch_params(params, dtype, requires_grad=True):
    tp = []
    for p in params:
        w = torch.tensor(np.asarray(p["w"]), dtype=dtype, requires_grad=requires_grad)
        b = None if p["b"] is None else torch.tensor(np.asarray(p["b"]), dtype=dtype, requires_grad=requires_grad)
        tp.append((w, b))
    return tp


def train_step(tp, state, batch, lr, R, LB, HB):
    """One full step on torch-CPU: forward, (B,) loss + L2, backward of the summed loss, Keras-Adam (epsilon outside
    the bias correction).  tp: list of (w, b) leaf tensors; state: dict with m, v, t.  Returns the (B,) loss."""
    pred = t_forward(tp, batch[:6], R, LB, HB)
    hires = torch.cat(batch[6:9], -1)
    l2 = sum(O.L2_LAMBDA * (w ** 2).sum() for w, _ in tp)
    loss = t_loss(pred, hires, batch[10]) + l2
    leaves = [t for wb in tp for t in wb if t is not None]
    grads = torch.autograd.grad(loss.sum(), leaves)
    if "t" not in state:
        state["t"] = 0
        state["m"] = [torch.zeros_like(t) for t in leaves]
        state["v"] = [torch.zeros_like(t) for t in leaves]
    state["t"] += 1
    t = state["t"]
    lr_t = lr * np.sqrt(1.0 - 0.999 ** t) / (1.0 - 0.9 ** t)
    with torch.no_grad():
        for p, g, m, v in zip(leaves, grads, state["m"], state["v"]):
            m.mul_(0.9).add_(g, alpha=0.1)
            v.mul_(0.999).addcmul_(g, g, value=0.001)
            p.sub_(lr_t * m / (v.sqrt() + 1e-7))
    return loss.detach()


def time_train_steps(P, R, LB, HB, B=1, threads=None, warm=2, reps=5):
    """Seconds of each of `reps` consecutive train steps of B patches (float32) after `warm` untimed ones, and the thread count."""
    if threads:
        torch.set_num_threads(int(threads))
    params = O.init_params(0, LB, HB, np.float32)
    tp = to_torch_params(params, torch.float32)
    batch = [torch.from_numpy(np.ascontiguousarray(a)) for a in O.synthetic_batch(B, P, R, seed=1234, dtype=np.float32)]
    state = {}
    times = []
    for i in range(warm + reps):
        t0 = time.time()
        train_step(tp, state, batch, 1e-4, R, LB, HB)
        if i >= warm:
            times.append(time.time() - t0)
    return times, torch.get_num_threads()


def time_train_step(P, R, LB, HB, B=1, threads=None, repeats=1):
    """Seconds per train step of B patches in float32 on `threads` host threads (default: torch's setting)."""
    if threads:
        torch.set_num_threads(int(threads))
    params = O.init_params(0, LB, HB, np.float32)
    tp = to_torch_params(params, torch.float32)
    batch = [torch.from_numpy(np.ascontiguousarray(a)) for a in O.synthetic_batch(B, P, R, seed=1234, dtype=np.float32)]
    state = {}
    best = None
    for _ in range(repeats):
        t0 = time.time()
        train_step(tp, state, batch, 1e-4, R, LB, HB)
        dt = time.time() - t0
        best = dt if best is None else min(best, dt)
    return best, torch.get_num_threads()


if __name__ == "__main__":
    # python -m oracle.torch_cpu P R LB HB warm reps [threads]  -> one JSON line (bench.py's cpu_baseline leg runs this in a child
    # process whose CPU affinity and OpenMP binding it set beforehand)
    import json
    import sys
    a = [int(v) for v in sys.argv[1:]]
    ts, n = time_train_steps(a[0], a[1], a[2], a[3], warm=a[4], reps=a[5], threads=a[6] if len(a) > 6 else None)
    print(json.dumps({"times": ts, "threads": n}))

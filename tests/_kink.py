"""Shared by the end-to-end gradient tests: which side of its ReLU / LeakyReLU kink every activation unit of a GPU forward landed on."""
import numpy as np


def kink_sides(cache, rc):
    """(sides, flips, worst): the side of the ReLU / LeakyReLU kink every activation unit of the fp32 GPU forward landed on (the `sides`
    argument of the oracle's backward), how many units sit on the other side than in the float64 oracle forward `rc`, and the largest
    |oracle activation| among those relative to the tensor's scale.  A unit within fp32 rounding of the kink may legitimately land on
    either side; each such flip moves gradient elements by ~1/(B*V) relative, which at these tiny test volumes is ~1e-3 -- so the oracle
    differentiates in the GPU's linear region, and the flips are counted and bounded instead of being forgiven."""
    flips, worst = 0, 0.0

    def one(t, ref):
        nonlocal flips, worst
        m = t.float().cpu().numpy() > 0
        f = m != (ref > 0)
        if f.any():
            flips += int(f.sum())
            worst = max(worst, float(np.abs(ref[f]).max() / max(np.abs(ref).max(), 1e-30)))
        return m
    sides = {k: one(cache[k], rc[k]) for k in ("a0", "a1", "p0", "p1", "c0", "c1")}
    sides["blocks"] = [(one(h, rc["blocks"][i][1]), one(out, rc["blocks"][i][2])) for i, (x, h, out) in enumerate(cache["blocks"])]
    sides["heads"] = [one(g, rc["heads"][i]) for i, g in enumerate(cache["heads"])]
    return sides, flips, worst

"""Host-side mirror of src/Network/SR4DFlowNet.py: the network graph, expressed as a fixed schedule of calls
into lib4dflow_hip.so.  Same names and argument meaning as the reference (SR4DFlowNet(res_increase)
.build_network(u, v, w, u_mag, v_mag, w_mag, low_resblock, hi_resblock, channel_nr)); the returned model is
callable on 6 arrays (B,P,P,P,1), has .predict(list), .trainable_variables, .save / .load_weights, and adds
.backward(dpred) (what tape.gradient does in TrainerController.py:223).

Parameters live in ONE flat fp32 device buffer in layer CREATION order (kernel, bias per layer: conv3d, conv3d_1, ...
conv3d_35), gradients in a second flat buffer of the same layout, so the data-parallel all-reduce and the Adam step are one
call each.  Keras' own `trainable_variables` order (depth-sorted layers, keras_layer_order) differs from creation order and
is what optimizer.pkl uses: keras_variable_order() / trainable_variable_names() translate between the two."""
import contextlib
import math
import os

import numpy as np
import torch

from . import ops, ops_bf16
from ._lib import FdnError

ACT_NONE, ACT_RELU, ACT_LEAKY = ops.ACT_NONE, ops.ACT_RELU, ops.ACT_LEAKY


class Input:
    """Placeholder with the role of tf.keras.layers.Input(shape=..., name=...) (TrainerController.py:38-44)."""

    def __init__(self, shape, name=None):
        self.shape = (None,) + tuple(shape)
        self.name = name


def layer_specs(low_resblock=8, hi_resblock=4):
    """[(name, K, Cin, Cout, use_bias)] in Keras creation order (SR4DFlowNet.py:17-46)."""
    specs = []

    def add(k, cin, cout, bias):
        i = len(specs)
        specs.append(("conv3d" if i == 0 else "conv3d_%d" % i, k, cin, cout, bias))

    add(3, 3, 64, True); add(3, 64, 64, True)        # pc path        :17-18
    add(3, 3, 64, True); add(3, 64, 64, True)        # phase path     :20-21
    add(1, 128, 64, True); add(3, 64, 64, True)      # concat fuse    :24-25
    for _ in range(low_resblock + hi_resblock):      # ResBlocks      :29-30,35-36 (no bias)
        add(3, 64, 64, False); add(3, 64, 64, False)
    for _ in range(3):                               # u,v,w heads    :39-46
        add(3, 64, 64, True); add(3, 64, 1, True)
    return specs


def keras_layer_order(low_resblock=8, hi_resblock=4):
    """Indices (into layer_specs) of the conv layers in the order tf.keras lists them in `model.layers` -- and hence their
    variables in `model.trainable_variables` / `optimizer.get_weights()` (TrainerController.py:223,347-394).

    A functional Keras model sorts its layers by depth from the output, deepest first, and within one depth by the order
    in which a depth-first walk from the output reaches them [TF 2.x `Network._map_graph_network` / `_build_map`; restated from
    the published algorithm, TensorFlow is absent here].  Only layers at EQUAL depth can change places relative to creation
    order, and the graph of SR4DFlowNet.py:7-51 has two such groups:
      * the pc and phase branches -- the walk enters `concatenate([phase, pc])` (:23) through `phase` first, so at each depth
        the phase conv precedes the pc conv: conv3d_2, conv3d, conv3d_3, conv3d_1;
      * the three heads -- `concatenate([u_path, v_path, w_path])` (:49) is entered through u first: the three 64->64 convs
        (conv3d_30, _32, _34 at cfg2), then the three 64->1 convs (conv3d_31, _33, _35).
    tests/golden/make_golden_tf.py stores the variable names a real TensorFlow produces; tests/test_tf_golden.py checks this
    function against them when that file exists."""
    n_trunk = 6 + 2 * (low_resblock + hi_resblock)
    h = n_trunk
    return [2, 0, 3, 1] + list(range(4, n_trunk)) + [h, h + 2, h + 4, h + 1, h + 3, h + 5]


class _Layer:
    __slots__ = ("name", "k", "cin", "cout", "w", "b", "gw", "gb", "wp_f", "wp_d", "w_off", "b_off")


class _T:
    """An activation tensor + the activation of the op that produced it (needed to form act'(y) in backward) + (bf16 mode, training) the
    sign mask its 64->64 producer wrote beside it: the fused dgrad reads 8 B per voxel of that instead of the 128-B rows of y (ops_bf16)."""
    __slots__ = ("t", "act", "mask")

    def __init__(self, t, act, mask=None):
        self.t = t
        self.act = act
        self.mask = mask


class FlowNetModel:
    def __init__(self, res_increase, low_resblock=8, hi_resblock=4, device=None, seed=0, dtype="float32", conv_algo=None):
        """dtype: storage type of activations and activation gradients -- "float32" (the reference's arithmetic) or
        "bfloat16" (BASELINE.json configs[3]; parameters, their gradients, the prediction and the optimizer stay fp32).
        conv_algo (fp32 mode): algorithm of the 64->64 3x3x3 layers -- "auto" (FDN_ALGO_AUTO: 2-D Winograd forward / dgrad, F(4,3) along
        H x F(4,3) along W where H and W are multiples of 4, F(2,3) along H where H is only even, Winograd along W where only W allows
        it, else direct -- forward() warns once per grid that falls off the 2-D kernels), "winograd_h2" (FDN_ALGO_WINO_H2: never more
        than F(2,3) along H, round 4's kernels), "winograd_w" (FDN_ALGO_WINO_W: the 1-D kernels only), "direct" (FDN_ALGO_DIRECT
        everywhere), or a dict {layer name: "direct"} that pins single layers (forward, dgrad and wgrad of that layer) to the direct
        kernels; None reads FDN_CONV_ALGO (default "auto")."""
        if not torch.cuda.is_available():
            raise FdnError("FlowNetModel needs a ROCm GPU: the hot path is HIP-only (no CPU fallback)")
        dtype = {"float32": "float32", "fp32": "float32", "f32": "float32", torch.float32: "float32",
                 "bfloat16": "bfloat16", "bf16": "bfloat16", torch.bfloat16: "bfloat16"}.get(dtype)
        if dtype is None:
            raise ValueError("dtype must be 'float32' or 'bfloat16'")
        self.dtype = dtype
        self.ops = ops if dtype == "float32" else ops_bf16
        self.act_dtype = torch.float32 if dtype == "float32" else torch.bfloat16
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self.res_increase = int(res_increase)
        self.low_resblock = int(low_resblock)
        self.hi_resblock = int(hi_resblock)
        self.specs = layer_specs(low_resblock, hi_resblock)
        n = sum(k ** 3 * ci * co + (co if ub else 0) for _, k, ci, co, ub in self.specs)
        self.n_params = n
        self.flat_w = torch.zeros(n, device=self.device, dtype=torch.float32)
        # gradient buffer + one trailing slot that carries the per-rank batch size through the data-parallel all-reduce
        self.flat_g_ext = torch.zeros(n + 1, device=self.device, dtype=torch.float32)
        self.flat_g = self.flat_g_ext[:n]
        self.batch_slot = self.flat_g_ext[n:n + 1]
        is_kernel = np.zeros(n, dtype=np.uint8)
        self.layers = []
        off = 0
        n64 = sum(1 for _, k, ci, co, _ in self.specs if (k, ci, co) == (3, 64, 64))
        pack_elems = ops.CONV64_PACK_FLOATS if dtype == "float32" else 27 * 64 * 64
        # (zero-filled: a pack holds five streams and only the ones the model's grids read are kept current -- _pack_streams below; the others
        # stay zero, not garbage.  The packs are private to forward() / backward(): read L.wp_f / L.wp_d elsewhere only through
        # _require_pack_streams, or re-pack everything with ops.pack_conv64_weights_batch(..., streams=None).)
        self._packs = torch.zeros((n64, 2, pack_elems), device=self.device, dtype=self.act_dtype)
        i64 = 0
        for name, k, ci, co, ub in self.specs:
            L = _Layer()
            L.name, L.k, L.cin, L.cout = name, k, ci, co
            sz = k ** 3 * ci * co
            L.w_off = off
            L.w = self.flat_w[off:off + sz].view(k, k, k, ci, co)
            L.gw = self.flat_g[off:off + sz].view(k, k, k, ci, co)
            is_kernel[off:off + sz] = 1
            off += sz
            if ub:
                L.b_off = off
                L.b = self.flat_w[off:off + co]
                L.gb = self.flat_g[off:off + co]
                off += co
            else:
                L.b_off = -1
                L.b = L.gb = None
            if (k, ci, co) == (3, 64, 64):
                L.wp_f, L.wp_d = self._packs[i64, 0], self._packs[i64, 1]
                i64 += 1
            else:
                L.wp_f = L.wp_d = None
            self.layers.append(L)
        assert off == n
        self.is_kernel = torch.tensor(is_kernel, device=self.device)
        self._w64_offsets = torch.tensor([L.w_off for L in self.layers if L.wp_f is not None], device=self.device, dtype=torch.int64)
        self.weights_version = 0       # bumped by weights_changed(): lets the trainer know whether Adam's sum-of-squares is current
        self._ws = None
        self._ws_bias = None
        self._side = None              # second HIP stream for the weight-gradient launches
        # weight gradients of 64->64 layers on small grids are collected per gradient bucket and issued as ONE batched launch
        # (fdn_conv3d_wgrad_batch): at 8 x 24^3 a layer alone on the chip leaves a workgroup 4.6 tiles between prologue and output
        # transform.  The batch ends where the bucket does, so the data-parallel all-reduce of a bucket starts as early as before.
        self.batch_wgrad = os.environ.get("FDN_BATCH_WGRAD", "1") not in ("", "0")
        # the input gradients of the three heads' 64->64 convs (they all read the last ResBlock's output) as ONE multi-source launch
        # (FDN_MULTI_DGRAD=0: three chained launches; equal to fp32 rounding in fp32, bf16 roundings of the partial sums apart in bf16 mode)
        self.multi_dgrad = os.environ.get("FDN_MULTI_DGRAD", "1") not in ("", "0")
        self.batch_wgrad_max_voxels = 1 << 18          # per launch; the 48^3 layers of cfg2 (8 x 110 592 voxels) fill the chip on their own
        self._wg_pending = []
        # weight gradients on a second HIP stream (they are leaves of the backward graph: they fill the tails of the dgrad launches and
        # the gaps the thin kernels leave; bit-identical results).  FDN_OVERLAP_WGRAD=0 issues everything on one stream -- what a
        # per-kernel timing pass wants (bench.py takes its HIP-event pass that way, in steps of its own).
        self.overlap_wgrad = os.environ.get("FDN_OVERLAP_WGRAD", "1") not in ("", "0")
        self._ws_side = None           # the side stream's own workspace (allocated and re-allocated under that stream: see _workspace)
        self._side_keep = []           # operands of side-stream launches, kept alive until the streams are joined
        self.set_conv_algo(conv_algo)
        self._cache = None
        # Gradient buckets in the order backward() completes them: slices [lo, hi) of flat_g_ext that are final when the hi-res part
        # (heads + hi-res blocks, together with the trailing batch slot), the upper half of the low-res blocks and the rest are done.
        # The data-parallel trainer starts one all-reduce per bucket as soon as it is final (trainer.train_step).
        cut_hi = self.layers[6 + 2 * self.low_resblock].w_off
        cut_mid = self.layers[6 + 2 * (self.low_resblock // 2)].w_off
        self.grad_buckets = [b for b in ((cut_hi, n + 1), (cut_mid, cut_hi), (0, cut_mid)) if b[1] > b[0]]
        self._slow_warned = set()
        # streams of the 64->64 packs that are kept current (fp32 mode): [forward packs, dgrad packs].  A pack holds four streams
        # and a grid reads one or two of them, so the per-step re-pack writes only what forward() has found the grids of this model
        # to read (_require_pack_streams asks the library's own selection code, widens the set and re-packs when a new grid or
        # algorithm needs more); nothing else reads the packs.
        self._pack_streams = [0, 0]
        # bf16 training: 64->64 layers write a sign mask beside their output and the fused dgrad reads it for act' instead of the output
        # (FDN_BF16_SIGN_MASK=0: read y, the round-4 behaviour -- same results bit for bit, for A/B timing)
        self.sign_masks = os.environ.get("FDN_BF16_SIGN_MASK" if dtype == "bfloat16" else "FDN_SIGN_MASK", "1") not in ("", "0")
        self._mask_ok_cache = {}
        self._pack_need_cache = {}
        self.glorot_uniform_init(seed)

    def set_conv_algo(self, conv_algo=None):
        """See __init__.  Takes effect from the next forward()."""
        if conv_algo is None:
            conv_algo = os.environ.get("FDN_CONV_ALGO", "auto")
        names = {"auto": ops.ALGO_AUTO, "winograd": ops.ALGO_AUTO, "direct": ops.ALGO_DIRECT, "winograd_w": ops.ALGO_WINO_W, "winograd_h2": ops.ALGO_WINO_H2,
                 "winograd_bf16x3": ops.ALGO_WINO_BF16X3}
        per_layer = {}
        if isinstance(conv_algo, dict):
            per_layer, conv_algo = conv_algo, conv_algo.get("*", "auto")
        if conv_algo not in names or any(v not in names for v in per_layer.values()):
            raise ValueError("conv_algo must be 'auto', 'winograd_bf16x3', 'winograd_h2', 'winograd_w', 'direct' or {layer name: one of these}")
        known = set(L.name for L in self.layers)
        unknown = [k for k in per_layer if k != "*" and k not in known]
        if unknown:
            raise ValueError("conv_algo: unknown layer(s) %s" % unknown)
        self.conv_algo = dict((L.name, names[per_layer.get(L.name, conv_algo)]) for L in self.layers)
        # FDN_ALGO_WINO_BF16X3 runs persistent workgroups that hold every CU's registers and LDS for the whole launch: a weight-gradient
        # launch on the second stream cannot share a CU with them and the two streams take turns badly (measured 28.9 ms one stream,
        # 39.4 ms two at cfg2).  Models that use it keep everything on one stream unless FDN_OVERLAP_WGRAD says otherwise.
        if ops.ALGO_WINO_BF16X3 in self.conv_algo.values() and "FDN_OVERLAP_WGRAD" not in os.environ and hasattr(self, "overlap_wgrad"):
            self.overlap_wgrad = False

    # ------------------------------------------------------------------ parameters
    def glorot_uniform_init(self, seed=0):
        """Keras default for kernel_initializer=None: GlorotUniform, zero bias (SR4DFlowNet.py:104).
        Drawn with numpy default_rng(seed) layer by layer in creation order."""
        rng = np.random.default_rng(seed)
        flat = np.zeros(self.n_params, dtype=np.float32)
        for L in self.layers:
            fan_in, fan_out = L.k ** 3 * L.cin, L.k ** 3 * L.cout
            limit = math.sqrt(6.0 / (fan_in + fan_out))
            wv = rng.uniform(-limit, limit, size=(L.k, L.k, L.k, L.cin, L.cout)).astype(np.float32)
            flat[L.w_off:L.w_off + wv.size] = wv.reshape(-1)
        self.flat_w.copy_(torch.from_numpy(flat))
        self.weights_changed()

    def weights_changed(self):
        """Re-derive the MFMA operand streams after any parameter update (Adam step, load_weights)."""
        self.weights_version += 1
        if self.dtype == "float32":
            if self._w64_offsets.numel():                  # all 64->64 layers, the streams in use, one launch
                ops.pack_conv64_weights_batch(self.flat_w, self._w64_offsets, self._packs, streams=self._pack_streams)
            return
        if self._w64_offsets.numel():
            ops_bf16.pack_conv64_weights_batch(self.flat_w, self._w64_offsets, self._packs)   # one launch (37 per-layer launches before)

    @property
    def trainable_variables(self):
        out = []
        for L in self.layers:
            out.append(L.w)
            if L.b is not None:
                out.append(L.b)
        return out

    def keras_variable_order(self):
        """Positions (into self.trainable_variables) in Keras `trainable_variables` order: see keras_layer_order."""
        first = {}
        k = 0
        for i, L in enumerate(self.layers):
            first[i] = (k, 2 if L.b is not None else 1)
            k += first[i][1]
        out = []
        for i in keras_layer_order(self.low_resblock, self.hi_resblock):
            out.extend(range(first[i][0], first[i][0] + first[i][1]))
        return out

    def trainable_variable_names(self):
        """Keras variable names ('conv3d_7/kernel:0', 'conv3d_7/bias:0'), one per entry of self.trainable_variables (creation order)."""
        out = []
        for L in self.layers:
            out.append("%s/kernel:0" % L.name)
            if L.b is not None:
                out.append("%s/bias:0" % L.name)
        return out

    def get_weights(self):
        return [t.detach().cpu().numpy().copy() for t in self.trainable_variables]

    def set_weights(self, arrays):
        tv = self.trainable_variables
        if len(arrays) != len(tv):
            raise ValueError("set_weights: expected %d arrays, got %d" % (len(tv), len(arrays)))
        for t, a in zip(tv, arrays):
            a = np.asarray(a, dtype=np.float32)
            if tuple(a.shape) != tuple(t.shape):
                raise ValueError("set_weights: shape mismatch %s vs %s" % (a.shape, tuple(t.shape)))
            t.copy_(torch.from_numpy(np.ascontiguousarray(a)))
        self.weights_changed()

    def save(self, path):
        from . import weights_io
        weights_io.save_model_weights(self, path)

    def load_weights(self, path):
        from . import weights_io
        weights_io.load_model_weights(self, path)

    # ------------------------------------------------------------------ forward
    def _to_dev(self, a):
        if isinstance(a, torch.Tensor):
            t = a.to(device=self.device, dtype=torch.float32)
        else:
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32))).to(self.device)
        return t.contiguous()

    def _warn_slow_grid(self, N, D, H, W):
        """FDN_ALGO_AUTO picks the 64->64 kernel by the grid's extents (include/fdn.h); a grid that falls off the 2-D Winograd kernels is
        a silent 1.5x - 3x slowdown of every such layer -- say so once per grid (only for launches big enough to matter)."""
        if self.dtype != "float32" or (D, H, W) in self._slow_warned or N * D * H * W < (1 << 15):
            return
        self._slow_warned.add((D, H, W))
        if not any(a == ops.ALGO_AUTO for a in self.conv_algo.values()):
            return
        if W % 4:
            h4, w4 = H & ~3, W & ~3
            if h4 >= 4 and w4 >= 4 and 2 * h4 * w4 >= H * W and N * D * H * W >= 24576:   # (conv64_mfma.hip, split_box: the aligned box on F(4,3) x F(4,3), the strips direct)
                return
            how = "the direct kernels (about 3x the time of the 2-D Winograd kernels)"
        elif H % 2:
            how = "the 1-D Winograd kernels (about 1.5x the time of the 2-D Winograd kernels)"
        else:
            return
        import warnings
        warnings.warn("4dflownet_amd: the 64->64 3x3x3 layers of the %dx%dx%d grid run on %s: FDN_ALGO_AUTO takes the 2-D Winograd kernels "
                      "only where W %% 4 == 0 and H is even (fastest: H %% 4 == 0 as well).  patch_size * res_increase (and patch_size "
                      "itself for the low-res stack) a multiple of 4 avoids this." % (D, H, W, how), RuntimeWarning, stacklevel=3)

    def _require_pack_streams(self, N, D, H, W, training):
        """Make sure the pack streams the 64->64 layers read on this grid are current (see __init__)."""
        if self.dtype != "float32":
            return
        algos = tuple(sorted(set(self.conv_algo[L.name] for L in self.layers if L.wp_f is not None)))
        key = (N, D, H, W, training, algos)
        need = self._pack_need_cache.get(key)
        if need is None:
            f = d = 0
            for a in algos:
                f |= ops.conv64_pack_streams(N, D, H, W, a, ops.ROLE_FWD)
                if training:
                    d |= ops.conv64_pack_streams(N, D, H, W, a, ops.ROLE_DGRAD_FUSED)
            need = self._pack_need_cache[key] = (f, d)
        new = [need[0] & ~self._pack_streams[0], need[1] & ~self._pack_streams[1]]
        if new[0] or new[1]:
            self._pack_streams = [self._pack_streams[0] | new[0], self._pack_streams[1] | new[1]]
            if self._w64_offsets.numel():
                ops.pack_conv64_weights_batch(self.flat_w, self._w64_offsets, self._packs, streams=new)

    def _conv(self, x, L, act, residual=None, x2=None, out=None, ldy=None, y_coff=0, mask=None):
        if mask is not None:                               # bf16 training, 64->64: the output and its sign mask (ops_bf16.conv3d_fwd)
            return self.ops.conv3d_fwd(x, L.w, L.b, act, ops.LEAKY_ALPHA, residual, None, L.wp_f, out, ldy, y_coff, mask=mask)
        return self.ops.conv3d_fwd(x, L.w, L.b, act, ops.LEAKY_ALPHA, residual, x2, L.wp_f, out, ldy, y_coff, algo=self.conv_algo[L.name])

    def _mask_ok(self, x, L):
        """fp32: do the forward and the fused dgrad of this grid write / read sign masks (the plain F(4,3) x F(4,3) kernels)?"""
        key = (tuple(x.shape[:4]), self.conv_algo[L.name])
        ok = self._mask_ok_cache.get(key)
        if ok is None:
            ok = self._mask_ok_cache[key] = ops.conv64_mask_ok(*key[0], key[1])
        return ok

    def _conv_m(self, x, L, act, residual=None, want_mask=False):
        """A 64->64 layer and, in training, the sign mask of its output (None where the kernels of the grid do not write one)."""
        if want_mask and self.sign_masks and act != ACT_NONE:
            if self.dtype == "bfloat16":
                mask = ops_bf16.new_sign_mask(x)
                return self._conv(x, L, act, residual=residual, mask=mask), mask
            if self._mask_ok(x, L):
                mask = ops.new_sign_mask(x)
                return ops.conv3d_fwd(x, L.w, L.b, act, ops.LEAKY_ALPHA, residual, None, L.wp_f, algo=self.conv_algo[L.name], mask=mask), mask
        return self._conv(x, L, act, residual=residual), None

    def forward(self, inputs, training=False):
        """inputs: [u, v, w, u_mag, v_mag, w_mag], each (B,P,P,P,1) or (B,P,P,P).  Returns a device tensor
        (B,PR,PR,PR,3).  With training=True every activation backward needs is kept in self._cache."""
        u, v, w, mu, mv, mw = [self._to_dev(a) for a in inputs]
        if u.dim() == 5:
            B, D, H, W = u.shape[:4]
        else:
            B, D, H, W = u.shape
        R = self.res_increase
        Ls = self.layers
        self._warn_slow_grid(B, D, H, W)
        self._require_pack_streams(B, D, H, W, training)
        if R > 1:                                          # (the three head convs run on the upsampled grid whatever hi_resblock is)
            if self.hi_resblock > 0:
                self._warn_slow_grid(B, D * R, H * R, W * R)
            self._require_pack_streams(B, D * R, H * R, W * R, training)
        phase = torch.empty((B, D, H, W, 3), device=self.device, dtype=self.act_dtype)
        pc = torch.empty((B, D, H, W, 3), device=self.device, dtype=self.act_dtype)
        self.ops.input_features(u, v, w, mu, mv, mw, phase, pc)
        a0 = self._conv(pc, Ls[0], ACT_RELU)
        a1 = self._conv(a0, Ls[1], ACT_RELU)
        p0 = self._conv(phase, Ls[2], ACT_RELU)
        p1 = self._conv(p0, Ls[3], ACT_RELU)
        c0 = self._conv(p1, Ls[4], ACT_RELU, x2=a1)          # concat [phase, pc] never materialised (:23)
        c1, m_c1 = self._conv_m(c0, Ls[5], ACT_RELU, want_mask=training)
        rb = _T(c1, ACT_RELU, m_c1)
        blocks = []
        hmasks = []                                        # bf16 training: sign mask of every block's inner activation h (else None)
        up = None
        li = 6
        nb = self.low_resblock + self.hi_resblock
        for i in range(nb + 1):
            if i == self.low_resblock and R > 1:
                up_out = _T(self.ops.upsample_trilinear_fwd(rb.t, R), ACT_NONE)
                up = (rb, up_out)
                rb = up_out
            if i == nb:
                break
            h, m_h = self._conv_m(rb.t, Ls[li], ACT_LEAKY, want_mask=training)
            out, m_out = self._conv_m(h, Ls[li + 1], ACT_LEAKY, residual=rb.t, want_mask=training)
            blocks.append((rb, h, out))
            hmasks.append(m_h)
            rb = _T(out, ACT_LEAKY, m_out)
            li += 2
        pred = torch.empty(tuple(rb.t.shape[:4]) + (3,), device=self.device)
        heads = []
        gmasks = []                                         # bf16 training: sign masks of the three head activations (else None)
        for hidx in range(3):
            g, m_g = self._conv_m(rb.t, Ls[li], ACT_RELU, want_mask=training)
            self._conv(g, Ls[li + 1], ACT_NONE, out=pred, ldy=3, y_coff=hidx)
            heads.append(g)
            gmasks.append(m_g)
            li += 2
        if training:
            self._cache = dict(phase=phase, pc=pc, a0=a0, a1=a1, p0=p0, p1=p1, c0=c0, c1=c1, blocks=blocks, up=up,
                               rb=rb, heads=heads, hmasks=hmasks, gmasks=gmasks)
        return pred

    __call__ = forward

    def predict(self, inputs, batch_size=None):
        """Keras Model.predict: numpy in, numpy out (predictor.py:87)."""
        n = len(inputs[0])
        bs = n if not batch_size else batch_size
        outs = []
        for s in range(0, n, bs):
            outs.append(self.forward([a[s:s + bs] for a in inputs]).cpu().numpy())
        return np.concatenate(outs, axis=0)

    # ------------------------------------------------------------------ backward
    def _workspace(self, nbytes, side=False):
        """Scratch of the weight-gradient launches.  One block per stream, each allocated (and, when a later layer needs more,
        re-allocated) while ITS stream is current: the caching allocator then hands a freed block only to later work of the same
        stream, i.e. behind the kernels that may still be using it."""
        if side:
            if self._ws_side is None or self._ws_side.numel() * 4 < nbytes:
                with torch.cuda.stream(self._side):
                    self._ws_side = torch.empty((nbytes + 3) // 4, device=self.device, dtype=torch.float32)
            return self._ws_side
        if self._ws is None or self._ws.numel() * 4 < nbytes:
            self._ws = torch.empty((nbytes + 3) // 4, device=self.device, dtype=torch.float32)
        return self._ws

    def _wgrad(self, x, dz, L, x2=None, lddz=None, dz_coff=0, bias=True):
        """Weight (+bias) gradient of layer L into the flat gradient buffer.  Weight gradients are leaves of the backward
        graph (nothing downstream reads them before the optimizer), so they run on a second HIP stream and fill the tails
        of the dgrad chain's kernels; backward() joins the streams before returning."""
        N, D, H, W = x.shape[:4]
        if (self.batch_wgrad and (L.k, L.cin, L.cout) == (3, 64, 64) and x2 is None and lddz is None and
                N * D * H * W <= self.batch_wgrad_max_voxels and
                (self.dtype != "float32" or self.conv_algo[L.name] in (ops.ALGO_AUTO, ops.ALGO_WINO_H2, ops.ALGO_WINO_BF16X3))):
            self._wg_pending.append((x, dz, L, bias))        # issued by _flush_wgrads() at the end of the gradient bucket
            return
        nws = self.ops.wgrad_workspace_bytes(N, D, H, W, L.cin, L.cout, L.k)
        if not self.overlap_wgrad:
            self.ops.conv3d_wgrad(x, dz, L.k, L.cin, L.cout, x2=x2, dw=L.gw, dbias=L.gb if bias else None, workspace=self._workspace(nws),
                             lddz=lddz, dz_coff=dz_coff, algo=self.conv_algo[L.name])
            return
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        main = torch.cuda.current_stream()
        self._side.wait_stream(main)                      # dz is ready
        ws = self._workspace(nws, side=True)
        with torch.cuda.stream(self._side):
            self.ops.conv3d_wgrad(x, dz, L.k, L.cin, L.cout, x2=x2, dw=L.gw, dbias=L.gb if bias else None, workspace=ws, lddz=lddz,
                             dz_coff=dz_coff, algo=self.conv_algo[L.name])
        self._side_keep.extend(t for t in (x, dz, x2) if t is not None)     # alive until the streams are joined (see _join_side)

    def _flush_wgrads(self):
        """Issue the collected weight gradients: layers of one grid and algorithm as one batched launch, a lone layer as before."""
        pending, self._wg_pending = self._wg_pending, []
        groups = {}
        for x, dz, L, bias in pending:
            groups.setdefault((tuple(x.shape[:4]), self.conv_algo[L.name]), []).append((x, dz, L, bias))
        for (shape, algo), items in groups.items():
            N, D, H, W = shape
            side = self.overlap_wgrad
            if side:
                if self._side is None:
                    self._side = torch.cuda.Stream(device=self.device)
                self._side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._side) if side else contextlib.nullcontext():
                if len(items) == 1:
                    x, dz, L, bias = items[0]
                    ws = self._workspace(self.ops.wgrad_workspace_bytes(N, D, H, W, 64, 64, 3), side=side)
                    self.ops.conv3d_wgrad(x, dz, 3, 64, 64, dw=L.gw, dbias=L.gb if bias else None, workspace=ws, algo=algo)
                else:
                    ws = self._workspace(self.ops.wgrad_batch_workspace_bytes(len(items), N, D, H, W), side=side)
                    self.ops.conv3d_wgrad_batch([i[0] for i in items], [i[1] for i in items], [i[2].gw for i in items],
                                                [i[2].gb if i[3] else None for i in items], workspace=ws, algo=algo)
            if side:
                for x, dz, _, _ in items:
                    self._side_keep.extend((x, dz))

    def _join_side(self):
        """The main stream waits for everything the side stream has been given; the operands of those launches may be freed after it.
        (Holding references until the join instead of Tensor.record_stream: with record_stream the caching allocator could not reuse a
        block until the side stream had passed its free point, took fresh device memory for the main stream meanwhile and never came
        back -- 0.8 GB per step, 96 GB reserved after 120 cfg2 steps, and one 2.9-s step when it finally had to give the cache back.)"""
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        self._side_keep.clear()

    def _pad_like(self, t):
        N, D, H, W, C = t.shape
        return torch.empty((N, D + 2, H + 2, W + 2, C), device=t.device, dtype=torch.float32)

    def _dgrad_fold(self, dz, L, skip, y_prev, act, mask=None):
        """dz_prev = (MirrorPadGrad(Conv3DBackpropInput(dz)) + skip) * act'(y_prev) for a 64->64 layer: interior voxels
        are finished by the conv epilogue, the surface by one small border kernel.  mask: the sign mask of y_prev, read by the conv
        epilogue instead of y_prev itself (bf16 mode; fp32 on the grids of the F(4,3) x F(4,3) kernels)."""
        out = torch.empty_like(dz)
        pad = self._pad_like(dz)
        kw = {} if self.dtype == "bfloat16" else {"algo": self.conv_algo[L.name]}
        if mask is not None and y_prev is not None and (self.dtype == "bfloat16" or self._mask_ok(dz, L)):
            kw["mask"] = mask
        self.ops.conv3d_dgrad_fused(dz, L.wp_d, pad, out, skip=skip, y_prev=y_prev, act=act, **kw)
        self.ops.fold_halo_border([pad], out, skip, y_prev, act)
        return out

    def backward(self, dpred, grad_ready=None):
        """Fill self.flat_g with d(sum_b loss_b)/d(params) given dpred (B,PR,PR,PR,3); L2 is NOT included here
        (it is folded into the Adam kernel).  Consumes the cache of the last forward(training=True).
        grad_ready(lo, hi), if given, is called once per entry of self.grad_buckets, in that order, as soon as every launch that
        writes flat_g_ext[lo:hi] has been enqueued on the current stream."""
        def bucket_done(k):
            self._flush_wgrads()                            # the batched weight gradients of this bucket's layers
            if grad_ready is not None and k < len(self.grad_buckets):
                if self._side is not None and self.overlap_wgrad:
                    # the bucket is complete once BOTH streams have run what they hold: the callback (the trainer starts the bucket's
                    # all-reduce in it) is issued from the side stream after that stream has been made to wait for the main one, so the
                    # collective is ordered behind both and the dgrad chain on the main stream does not stop for it
                    self._side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(self._side):
                        grad_ready(*self.grad_buckets[k])
                else:
                    grad_ready(*self.grad_buckets[k])
        done = 0
        c = self._cache
        if c is None:
            raise FdnError("backward() without forward(training=True)")
        self._cache = None
        Ls = self.layers
        R = self.res_increase
        rb = c["rb"]
        li = len(Ls) - 6
        def act_of(t):                     # (tensor for act', act) of a producer; linear producers need no mask
            return (t.t if t.act != ACT_NONE else None), t.act

        # three heads fan into rb: chain the interior accumulation through `dz` (skip aliases the output),
        # apply act'(rb) on the last one, then one border fold over the three padded scratches
        dz = torch.empty_like(rb.t)
        pads = []
        # ONE multi-source launch forms the sum of the three input gradients in its registers (ops.conv3d_dgrad_fused_multi; fp32: on the
        # grids of the F(4,3) x F(4,3) kernels) instead of three chained launches that re-read and re-write the running sum
        multi = self.multi_dgrad and (self.dtype == "bfloat16" or (len({self.conv_algo[Ls[li + 2 * h].name] for h in range(3)}) == 1
                                                                   and self._mask_ok(rb.t, Ls[li])))
        dz_gs = []
        for hidx in range(3):
            L1, L2 = Ls[li], Ls[li + 1]
            g = c["heads"][hidx]
            c["heads"][hidx] = None
            self._wgrad(g, dpred, L2, lddz=3, dz_coff=hidx)
            # the folded head dgrad also emits the bias gradient of the 64->64 head conv (sum of dz_g) while it has it in registers
            if self._ws_bias is None:
                self._ws_bias = torch.empty(2048 * 64, device=self.device, dtype=torch.float32)
            m_g = c["gmasks"][hidx]
            mk = {} if m_g is None else {"mask": m_g}          # (bf16 mode: the head activation's sign mask instead of its rows)
            dz_g = self.ops.conv_cout1_dgrad_folded(dpred, L2.w, tuple(g.shape[:4]), g, ACT_RELU, lddz=3, dz_coff=hidx,
                                               dbias_prev=L1.gb, workspace=self._ws_bias, **mk)
            c["gmasks"][hidx] = None
            del m_g, mk
            del g
            self._wgrad(rb.t, dz_g, L1, bias=False)
            if multi:
                dz_gs.append(dz_g)
                del dz_g
                li += 2
                continue
            pad = self._pad_like(rb.t)
            y_m, a_m = act_of(rb) if hidx == 2 else (None, ACT_NONE)
            if y_m is not None and rb.mask is not None and (self.dtype == "bfloat16" or self._mask_ok(dz_g, L1)):
                self.ops.conv3d_dgrad_fused(dz_g, L1.wp_d, pad, dz, skip=dz if hidx > 0 else None, y_prev=y_m, act=a_m, mask=rb.mask,
                                            **({} if self.dtype == "bfloat16" else {"algo": self.conv_algo[L1.name]}))
            else:
                self.ops.conv3d_dgrad_fused(dz_g, L1.wp_d, pad, dz, skip=dz if hidx > 0 else None, y_prev=y_m, act=a_m,
                                            algo=self.conv_algo[L1.name])
            pads.append(pad)
            del dz_g
            li += 2
        y_m, a_m = act_of(rb)
        if multi:
            pad = self._pad_like(rb.t)
            use_mask = y_m is not None and rb.mask is not None
            self.ops.conv3d_dgrad_fused_multi(dz_gs, [Ls[li - 6 + 2 * h].wp_d for h in range(3)], pad, dz, y_prev=None if use_mask else y_m, act=a_m,
                                              mask=rb.mask if use_mask else None, algo=self.conv_algo[Ls[li - 6].name])
            pads.append(pad)
            del dz_gs
        self.ops.fold_halo_border(pads, dz, None, y_m, a_m)
        del pads, pad
        li = len(Ls) - 6
        nb = self.low_resblock + self.hi_resblock
        up = c["up"]
        for i in range(nb, -1, -1):
            if i == self.low_resblock or (done == 1 and i == self.low_resblock // 2 and len(self.grad_buckets) == 3):
                bucket_done(done)
                done += 1
            if up is not None and i == self.low_resblock:
                # dz currently holds d(up_out) (linear producer): pull it through the upsample
                y_m, a_m = act_of(up[0])
                dz = self.ops.upsample_trilinear_bwd(dz, R, y_m, a_m)
            if i == 0:
                break
            x, h, out = c["blocks"][i - 1]
            m_h = c["hmasks"][i - 1]
            c["blocks"][i - 1] = c["hmasks"][i - 1] = None
            li -= 2
            La, Lb = Ls[li], Ls[li + 1]
            self._wgrad(h, dz, Lb)
            dz_h = self._dgrad_fold(dz, Lb, None, h, ACT_LEAKY, mask=m_h)
            del m_h
            self._wgrad(x.t, dz_h, La)
            y_m, a_m = act_of(x)
            dz = self._dgrad_fold(dz_h, La, dz, y_m, a_m, mask=x.mask)
            del dz_h
        assert li == 6
        # dz == dz_c1
        self._wgrad(c["c0"], dz, Ls[5])
        dz_c0 = self._dgrad_fold(dz, Ls[5], None, c["c0"], ACT_RELU)
        self._wgrad(c["p1"], dz_c0, Ls[4], x2=c["a1"])
        dz_p1, dz_a1 = self.ops.conv1x1_dgrad(dz_c0, Ls[4].w, c["p1"], c["a1"])
        for (first, second, src, dzz) in ((Ls[2], Ls[3], "p", dz_p1), (Ls[0], Ls[1], "a", dz_a1)):
            x0 = c[src + "0"]
            self._wgrad(x0, dzz, second)
            dz0 = self._dgrad_fold(dzz, second, None, x0, ACT_RELU)
            self._wgrad(c["phase"] if src == "p" else c["pc"], dz0, first)
        self._flush_wgrads()
        self._join_side()                                            # all weight gradients have landed in flat_g
        while done < len(self.grad_buckets):
            bucket_done(done)
            done += 1
        return self.flat_g


class SR4DFlowNet:
    """Same constructor / build_network signature as the reference (SR4DFlowNet.py:4-7).  The six leading
    arguments are placeholders there (Keras Inputs); here they are accepted and only used to sanity-check
    the patch shape.  channel_nr is forced to 64 exactly like SR4DFlowNet.py:8."""

    def __init__(self, res_increase):
        self.res_increase = res_increase

    def build_network(self, u, v, w, u_mag, v_mag, w_mag, low_resblock=8, hi_resblock=4, channel_nr=64, device=None,
                      seed=0, dtype="float32", conv_algo=None):
        channel_nr = 64   # noqa: F841  (the reference overwrites the argument)
        for t in (u, v, w, u_mag, v_mag, w_mag):
            shp = getattr(t, "shape", None)
            if shp is not None and len(shp) == 5 and shp[-1] != 1:
                raise ValueError("inputs must have a single channel, got shape %s" % (tuple(shp),))
        return FlowNetModel(self.res_increase, low_resblock, hi_resblock, device=device, seed=seed, dtype=dtype, conv_algo=conv_algo)

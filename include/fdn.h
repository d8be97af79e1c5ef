/* lib4dflow_hip.so -- C-ABI of the MI355X-native 4DFlowNet hot path.
 *
 * The reference (EdwardFerdian/4DFlowNet) has no FFI: it calls TensorFlow/Keras ops from Python.
 * Each entry point below replaces the TF op(s) the reference invokes implicitly at the cited
 * file:line (paths relative to the reference root).  Conventions:
 *   - every pointer is a DEVICE pointer to fp32 data owned by the caller (torch tensors);
 *     the library allocates nothing and keeps no global mutable state (the only process-wide
 *     data is a lock-protected record of which kernels already had their LDS attribute set,
 *     per device).  Entry points are re-entrant per (device, stream) from any host thread;
 *   - scratch memory is caller-owned too: SURVEY.md 8(b) sketched fdn_workspace_create/destroy,
 *     which this ABI deliberately replaces by explicit `workspace` arguments sized by
 *     fdn_*_workspace_bytes() -- the caller's allocator (torch's caching allocator) already
 *     recycles such buffers stream-safely, and a library-held handle would be global state;
 *   - there is NO CPU build of these symbols (SURVEY.md 8(b) asked for a g++/OpenMP one so tests
 *     could run without a GPU): a second implementation behind the same names is exactly the
 *     silent-fallback hazard the parity claims must exclude.  Without a GPU the tests check that
 *     the library builds, loads and exports every symbol declared here (tests/test_abi.py); the
 *     arithmetic is checked on the GPU against oracle/ (a separate, test-only restatement);
 *   - the variant-forcing / ablation switches used by tests and tools (fdn_debug_*) are NOT
 *     part of this ABI and are not exported by lib4dflow_hip.so; they exist only in
 *     lib4dflow_hip_test.so (same sources, -DFDN_TEST_HOOKS);
 *   - tensors are NDHWC, channel innermost; conv kernels are Keras layout (kd,kh,kw,Cin,Cout);
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - return value: FDN_OK or a negative error code; fdn_last_error() gives a message
 *     (thread-local).  No exceptions cross the boundary.
 */
#ifndef FDN_H
#define FDN_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define FDN_OK 0
#define FDN_ERR_BAD_ARG (-1)
#define FDN_ERR_UNSUPPORTED (-2)
#define FDN_ERR_HIP (-3)
#define FDN_ERR_WORKSPACE (-4)

#define FDN_ACT_NONE 0
#define FDN_ACT_RELU 1  /* Conv3D(activation='relu')            src/Network/SR4DFlowNet.py:17-25,39,42,45 */
#define FDN_ACT_LEAKY 2 /* tf.keras.layers.LeakyReLU(alpha=0.2) src/Network/SR4DFlowNet.py:113,118 */

/* Algorithm selector of the 64->64 3x3x3 entry points (fdn_conv3d_fwd / _dgrad / _dgrad_fused[_part] / _wgrad; ignored by
 * every other (Cin,Cout,K)).  Per call, no global state.
 *   FDN_ALGO_AUTO   : the planner's choice.  Forward / dgrad: 2-D Winograd, F(4,3) along H x F(4,3) along W (a quarter of the
 *                     direct multiplies) when H and W are multiples of 4, F(2,3) along H x F(4,3) along W (a third) when H is
 *                     only even, else 1-D Winograd along W (F(4,3): half the multiplies) when W is a multiple of 4, else
 *                     direct.  wgrad: F(3,4) along W when W is a multiple of 4 (+ F(3,2) along D when D is even).
 *                     fp32 error up to ~1e-6 of sum|x||w| (F(4,3) x F(4,3)) instead of ~1e-7;
 *   FDN_ALGO_DIRECT : always the direct convolution (plain fp32 FMA chains over the 27 taps, no transform) -- for
 *                     parity-critical runs and for layers whose operands are too ill-conditioned for the transform;
 *   FDN_ALGO_WINO_W : Winograd along W only (the 1-D kernels), never the H transform;
 *   FDN_ALGO_WINO_H2: like AUTO, but never more than F(2,3) along H (round 4's kernels; a third of the error of F(4,3) x F(4,3));
 *   FDN_ALGO_WINO_BF16X3: like AUTO, but where AUTO takes F(4,3) x F(4,3) (forward, dgrad inner box) the Winograd-domain products run
 *                     on the bf16 matrix pipe: both operands (fp32, transformed exactly as under AUTO) are split EXACTLY into three
 *                     bf16 pieces v = hi + mid + lo, and the six cross terms hi.hi, mid.hi, lo.hi, hi.mid, mid.mid, hi.lo are
 *                     accumulated in fp32 (a bf16 x bf16 product is exact in fp32; the three dropped terms are <= 2^-25 |u||v|, below
 *                     the half ulp an fp32 multiply rounds away).  Inputs, outputs, transforms and accumulation stay fp32.  Every
 *                     other grid and the weight gradient behave as under AUTO. */
#define FDN_ALGO_AUTO 0
#define FDN_ALGO_DIRECT 1
#define FDN_ALGO_WINO_W 2
#define FDN_ALGO_WINO_H2 3
#define FDN_ALGO_WINO_BF16X3 4
#define FDN_ALGO_LAST FDN_ALGO_WINO_BF16X3

/* Version of this header; fdn_version() returns the version the library was built from.  A caller must see the two equal:
 * 140 -> 150 and 150 -> 160 grew FDN_CONV64_PACK_FLOATS (a pack buffer sized by an older header is too small for this library). */
#define FDN_VERSION 161
int fdn_version(void);
const char* fdn_last_error(void);

/* speed/mag/pcmr + the two channel concats.  src/Network/SR4DFlowNet.py:10-15.
 * u..mw: (nvox) each; phase, pc: (nvox,3). */
int fdn_input_features(const float* u, const float* v, const float* w, const float* mu, const float* mv,
                       const float* mw, float* phase, float* pc, int64_t nvox, void* stream);

/* Re-layout one 3x3x3 64->64 Keras kernel (27,64,64) into the MFMA operand streams used by
 * fdn_conv3d_fwd (wp_fwd) and fdn_conv3d_dgrad / fdn_conv3d_dgrad_fused (wp_dgrad: taps flipped,
 * Cin/Cout swapped).  Each output is FDN_CONV64_PACK_FLOATS floats: the direct-convolution stream
 * (27 taps), the Winograd F(4,3)-along-W stream U = G g (9 (kd,kh) taps x 6 transform coordinates)
 * and the two 2-D streams U = Gh g Gw^T (3 kd taps x 4 x 6 coordinates for F(2,3) along H, 3 x 6 x 6
 * for F(4,3) along H), and the F(4,3) x F(4,3) stream once more as three bf16 pieces per value (FDN_ALGO_WINO_BF16X3:
 * 162 * 4096 float-sized slots); the conv entry points select among them by the extents (see FDN_ALGO_*).
 * Either output may be NULL. */
#define FDN_CONV64_PACK_FLOATS (423 * 64 * 64)
int fdn_pack_conv64_weights(const float* w, float* wp_fwd, float* wp_dgrad, void* stream);
/* The same for n_layers kernels in ONE launch (after every optimizer step): layer i lives at
 * w_base + w_offsets[i] (w_offsets: DEVICE array of n_layers float offsets), its two streams at
 * packs + i * 2 * FDN_CONV64_PACK_FLOATS (forward stream first, dgrad stream second). */
int fdn_pack_conv64_weights_batch(const float* w_base, const int64_t* w_offsets, int n_layers, float* packs,
                                  void* stream);
/* A pack holds five streams and a given grid reads one or two of them.  fdn_conv64_pack_streams says which: the streams
 * (FDN_PACK_STREAM_* bits) that fdn_conv3d_fwd (role FDN_ROLE_FWD, reads wp_fwd), fdn_conv3d_dgrad (FDN_ROLE_DGRAD) or
 * fdn_conv3d_dgrad_fused[_part] (FDN_ROLE_DGRAD_FUSED; both read wp_dgrad) read for a 64->64 layer on an (N,D,H,W) grid
 * under `algo` -- answered by the launcher's own selection code, so it cannot drift from it; negative = error code.
 * fdn_pack_conv64_weights_batch_streams is fdn_pack_conv64_weights_batch restricted to the named streams of the forward
 * and of the dgrad packs (the others keep whatever they held: a caller that narrows the set must re-pack before a
 * grid that needs more).  cfg2 reads F(4,3)xF(4,3) forward and F(4,3)xF(4,3) + 1-D Winograd (shell faces) in dgrad: 270 of
 * the 522 x 4096 floats per layer.  The per-step re-pack after the optimizer, TrainerController.py:225. */
#define FDN_PACK_STREAM_DIRECT 1
#define FDN_PACK_STREAM_WINO_W 2
#define FDN_PACK_STREAM_WINO_H2 4
#define FDN_PACK_STREAM_WINO_H4 8
#define FDN_PACK_STREAM_WINO_H4S 16 /* F(4,3) x F(4,3), three bf16 pieces per value (FDN_ALGO_WINO_BF16X3) */
#define FDN_PACK_STREAM_ALL 31
#define FDN_ROLE_FWD 0
#define FDN_ROLE_DGRAD 1
#define FDN_ROLE_DGRAD_FUSED 2
int fdn_conv64_pack_streams(int N, int D, int H, int W, int algo, int role);
int fdn_pack_conv64_weights_batch_streams(const float* w_base, const int64_t* w_offsets, int n_layers, float* packs,
                                          int streams_fwd, int streams_dgrad, void* stream);

/* y = act(conv3d(sym_pad(x), w) + bias + residual).
 * Replaces tf.pad(SYMMETRIC,p=(K-1)/2) + Conv3D(valid) + BiasAdd + activation, and the
 * resnet_block add + LeakyReLU when `residual` is given.  src/Network/SR4DFlowNet.py:93-120.
 * Supported (Cin,Cout,K): (64,64,3) [MFMA path; needs wpack from fdn_pack_conv64_weights],
 * (3,64,3), (64,1,3), (128,64,1) [x = first 64 input channels, x2 = last 64: the concat at
 * SR4DFlowNet.py:23 is never materialised].  x2, wpack, bias, residual may be NULL where unused.
 * Output rows are written at y[voxel*ldy + y_coff + c] (ldy=Cout,y_coff=0 for a dense tensor;
 * the three 64->1 heads write straight into the (N,V,3) prediction: SR4DFlowNet.py:49).
 * algo: FDN_ALGO_AUTO | FDN_ALGO_DIRECT | FDN_ALGO_WINO_W, consulted by the (64,64,3) path only (see above); here and in the entry points below. */
int fdn_conv3d_fwd(const float* x, const float* x2, const float* w, const float* wpack, const float* bias,
                   const float* residual, float* y, int N, int D, int H, int W, int Cin, int Cout, int K,
                   int ldy, int y_coff, int act, float alpha, int algo, void* stream);

/* Gradient w.r.t. the PADDED conv input (Conv3DBackpropInputV2): dxpad (N,D+2,H+2,W+2,Cin) for K=3.
 * dz rows are read at dz[voxel*lddz + dz_coff + c].  Fold the halo with fdn_fold_halo.
 * Supported (Cin,Cout,K): (64,64,3) [needs wpack = wp_dgrad], (64,1,3). */
int fdn_conv3d_dgrad(const float* dz, const float* w, const float* wpack, float* dxpad, int N, int D, int H,
                     int W, int Cin, int Cout, int K, int lddz, int dz_coff, int algo, void* stream);

/* The 64->1 head conv's input gradient with the halo fold and the producer's activation gradient fused
 * (no padded intermediate): dz_prev[i] = act'(y_prev[i]) * sum_{(o,t): clamp(o+t-1)=i} w[t] * dz[o].
 * dz rows at dz[voxel*lddz + dz_coff]; y_prev may be NULL.  If dbias_prev != NULL it also receives
 * BiasAddGrad of the producing layer (per-channel sum of dz_prev, 64 floats), using `workspace`
 * (>= 2048*64*4 bytes).  SR4DFlowNet.py:39-46 under tape.gradient. */
int fdn_conv_cout1_dgrad_folded(const float* dz, const float* w, const float* y_prev, int act, float alpha,
                                float* dz_prev, float* dbias_prev, void* workspace, size_t workspace_bytes, int N,
                                int D, int H, int W, int lddz, int dz_coff, void* stream);
/* ... reading the producer's sign mask (fdn_conv64_fwd_mask: planar [cout / 16][voxel] words; W % 4 == 0) instead of y_prev: eight 8-B loads
 * per 16 voxels and lane instead of 32 rows, 7 MB instead of 226 MB per (8,48^3) launch.  Bit-identical to the y_prev form. */
int fdn_conv_cout1_dgrad_folded_mask(const float* dz, const float* w, const uint16_t* y_mask, int act, float alpha,
                                     float* dz_prev, float* dbias_prev, void* workspace, size_t workspace_bytes, int N,
                                     int D, int H, int W, int lddz, int dz_coff, void* stream);

/* MirrorPadGrad + gradient fan-in + activation gradient in one pass:
 * dz_prev[i] = (sum_s sum_{P: clamp(P)=i} dxpad_s[P] + skip[i]) * act'(y_prev[i]).
 * nsrc in 1..3; skip, y_prev may be NULL (act' = 1). */
int fdn_fold_halo(const float* dxpad0, const float* dxpad1, const float* dxpad2, int nsrc, const float* skip,
                  const float* y_prev, int act, float alpha, float* dz_prev, int N, int D, int H, int W,
                  int C, void* stream);

/* 64->64 dgrad with MirrorPadGrad fused for the voxels strictly inside the volume (each receives exactly one
 * contribution): there  dz_prev[i] = (dgrad[i] + skip[i]) * act'(y_prev[i])  is written by the conv epilogue;
 * every other padded position is written to dxpad (N,D+2,H+2,W+2,64) and finished by fdn_fold_halo_border,
 * which applies the same formula on the surface voxels with up to 3 padded sources.  skip may alias dz_prev
 * (gradient fan-in over several consumers: call with y_prev=NULL for all but the last).  wpack = wp_dgrad. */
int fdn_conv3d_dgrad_fused(const float* dz, const float* wpack, float* dxpad, const float* skip, const float* y_prev,
                           int act, float alpha, float* dz_prev, int N, int D, int H, int W, int algo, void* stream);
/* The same in two independent pieces, for callers that overlap them on two streams (they write disjoint positions; both must
 * have completed before fdn_fold_halo_border): FDN_DGRAD_INNER = the D x H x W box (finishes the interior of dz_prev, writes its
 * surface voxels to dxpad), FDN_DGRAD_SHELL = the one-voxel shell of the padded grid (dxpad only; ignores skip / y_prev). */
#define FDN_DGRAD_INNER 1
#define FDN_DGRAD_SHELL 2
int fdn_conv3d_dgrad_fused_part(const float* dz, const float* wpack, float* dxpad, const float* skip, const float* y_prev,
                                int act, float alpha, float* dz_prev, int N, int D, int H, int W, int parts, int algo,
                                void* stream);
int fdn_fold_halo_border(const float* dxpad0, const float* dxpad1, const float* dxpad2, int nsrc, const float* skip,
                         const float* y_prev, int act, float alpha, float* dz_prev, int N, int D, int H, int W,
                         void* stream);
/* Sign masks for the activation gradient (training; the fp32 twin of the bf16 mode's).  act'(y) needs one bit of y, and at the large
 * grids an epilogue operand costs the fused dgrad what streaming it from HBM costs: the forward of a 64->64 layer can write, beside y,
 * the mask y_mask[c / 16][voxel] (uint16_t words, N*D*H*W per plane, 4 planes: bit b of word (p, v) = (y[v][16 p + b] > 0); 8 B per voxel)
 * and the fused dgrad of its consumer read that instead of y_prev.  Only the plain F(4,3) x F(4,3) kernels do (H and W multiples of 4,
 * FDN_ALGO_AUTO): fdn_conv64_mask_ok says 1 when BOTH entry points below are available for the grid, 0 when the caller must keep to
 * fdn_conv3d_fwd / fdn_conv3d_dgrad_fused (they return FDN_ERR_UNSUPPORTED otherwise).  fdn_fold_halo_border still reads y_prev (surface
 * voxels only).  Results are bit-identical to the y_prev forms.  src/Network/SR4DFlowNet.py:93-120 and their gradients. */
int fdn_conv64_mask_ok(int N, int D, int H, int W, int algo);
int fdn_conv64_fwd_mask(const float* x, const float* wpack, const float* bias, const float* residual, float* y, uint16_t* y_mask,
                        int N, int D, int H, int W, int act, float alpha, int algo, void* stream);
int fdn_conv64_dgrad_fused_mask(const float* dz, const float* wpack, float* dxpad, const float* skip, const uint16_t* y_mask,
                                int act, float alpha, float* dz_prev, int N, int D, int H, int W, int algo, void* stream);
/* Fused dgrad of up to three 64->64 layers that share their INPUT (the u, v, w heads' first convs all read the last ResBlock's output,
 * src/Network/SR4DFlowNet.py:39-46; under tape.gradient their input gradients add up, TrainerController.py:223):
 *   dz_prev = (MirrorPadGrad(sum_s Conv3DBackpropInput(dz[s], W_s)) + skip) * act'(y_prev)
 * as ONE launch + one fdn_fold_halo_border instead of nsrc chained fdn_conv3d_dgrad_fused calls (each re-reading the running sum as
 * `skip` and writing it back): the kernel walks the sources inside its tile and keeps the sum in its output registers.  dz / wpack:
 * host arrays of nsrc (1..3) device pointers, wpack[s] = the dgrad pack of layer s; the packs must come from one
 * fdn_pack_conv64_weights_batch buffer (within 1 GiB of each other).  y_prev or y_mask (fdn_conv64_fwd_mask's) or neither (act =
 * FDN_ACT_NONE).  Available where fdn_conv64_mask_ok(N, D, H, W, algo) says 1 (FDN_ERR_UNSUPPORTED elsewhere: chain the single-source
 * entry point).  Equal to the chained launches to fp32 rounding (the sum is formed in the Winograd domain, in pack-address order). */
int fdn_conv64_dgrad_fused_multi(const float* const* dz, const float* const* wpack, int nsrc, float* dxpad, const float* skip,
                                 const float* y_prev, const uint16_t* y_mask, int act, float alpha, float* dz_prev, int N, int D,
                                 int H, int W, int algo, void* stream);

/* 1x1x1 128->64 conv backward w.r.t. its two 64-channel inputs, fused with their ReLU masks:
 * dxa = (dz . W[0:64,:]^T) * (ya>0), dxb = (dz . W[64:128,:]^T) * (yb>0).  SR4DFlowNet.py:23-24. */
int fdn_conv1x1_dgrad(const float* dz, const float* w, const float* ya, const float* yb, float* dxa,
                      float* dxb, int64_t nvox, void* stream);

/* dW (K,K,K,Cin,Cout) = Conv3DBackpropFilterV2(sym_pad(x), dz); dbias (Cout) = BiasAddGrad(dz) if
 * dbias != NULL.  Same (Cin,Cout,K) support and x/x2 convention as fdn_conv3d_fwd.
 * workspace: caller-owned scratch of at least fdn_conv3d_wgrad_workspace_bytes(...). */
size_t fdn_conv3d_wgrad_workspace_bytes(int N, int D, int H, int W, int Cin, int Cout, int K);

/* The weight gradients of n_layers 64->64 3x3x3 layers that share one (N, D, H, W) grid, as ONE launch + one reduction where the
 * Winograd kernel applies (W % 4 == 0, D even, algo AUTO / WINO_H2 / WINO_BF16X3), else layer by layer.  Results equal n_layers calls
 * of fdn_conv3d_wgrad to fp32 rounding, not bit for bit, where the batched kernel runs: it gives every layer 64 / n_layers splits of the
 * voxel sum, the single-layer launch 63 (the layer-by-layer fall-back IS those calls).  x, dz, dw (and dbias, which may be NULL or hold NULL entries): HOST arrays of n_layers device pointers
 * (dz rows dense, 64 channels).  Why: at the low-res grid of cfg2 (8 x 24^3) a layer that has the chip to itself gives a workgroup
 * 4.6 tiles between its prologue and its output transform; batched, the ResBlock layers of one gradient bucket share the chip and
 * each workgroup walks n_layers times more tiles of its layer.  src/Network/TrainerController.py:223 (tape.gradient). */
size_t fdn_conv3d_wgrad_batch_workspace_bytes(int n_layers, int N, int D, int H, int W);
int fdn_conv3d_wgrad_batch(const float* const* x, const float* const* dz, float* const* dw, float* const* dbias, int n_layers,
                           void* workspace, size_t workspace_bytes, int N, int D, int H, int W, int algo, void* stream);
int fdn_conv3d_wgrad(const float* x, const float* x2, const float* dz, float* dw, float* dbias,
                     void* workspace, size_t workspace_bytes, int N, int D, int H, int W, int Cin, int Cout,
                     int K, int lddz, int dz_coff, int algo, void* stream);

/* upsample3d: trilinear, align_corners=True, integer factor R.  src/Network/SR4DFlowNet.py:53-90.
 * fwd: x (N,D,H,W,C) -> y (N,DR,HR,WR,C).
 * bwd: dx = U^T dy, optionally * act'(y_prev) with y_prev (N,D,H,W,C). */
int fdn_upsample_trilinear_fwd(const float* x, float* y, int N, int D, int H, int W, int C, int R, void* stream);
int fdn_upsample_trilinear_bwd(const float* dy, const float* y_prev, int act, float alpha, float* dx, int N,
                               int D, int H, int W, int C, int R, void* stream);

/* Loss + metric + dPred in one call.  src/Network/TrainerController.py:84-127,143-156,
 * src/Network/loss_utils.py:64-103.
 * pred (N,V,3); uh,vh,wh (N,V); mask (N,V).  out (N,4) = {mse loss, rel-error %, sum(mask), sum(nonfluid)}.
 * dpred (N,V,3) = d(sum_b loss_b)/dpred, or NULL (test_step).  scratch: FDN_LOSS_SCRATCH_FLOATS(N) floats (per-block
 * partial sums, added up in a fixed order: the reported values are run-to-run identical). */
#define FDN_LOSS_BLOCKS 256
#define FDN_LOSS_SCRATCH_FLOATS(N) ((N) * (8 + 3 * FDN_LOSS_BLOCKS))
int fdn_loss_metrics(const float* pred, const float* uh, const float* vh, const float* wh, const float* mask,
                     float* out, float* dpred, float* scratch, int N, int64_t V, void* stream);

/* On-device input pipeline: the per-sample slicing / np.rot90 / sign / normalisation / mask threshold of
 * PatchHandler3D.load_patches_from_index_file (src/Network/PatchHandler3D.py:49-160) as one gather per output
 * tensor.  desc: B device-resident descriptors of 56 bytes each
 *   { const float* src (T,X,Y,Z volume); int32 X,Y,Z,t, x0,y0,z0, plane (0 none,1:(0,1),2:(0,2),3:(1,2)), k (rot90 count),
 *     mode (0: out = sign*(v/div), 1: out = v >= div ? 1 : 0); float sign, div }
 * out: (B,S,S,S) patches, S = patch edge. */
int fdn_gather_patches(const void* desc, float* out, int B, int S, void* stream);

/* sum of squares of the kernel (non-bias) parameters: the l2(5e-7) regulariser value is 5e-7 * out[0].
 * src/Network/TrainerController.py:129-141.  is_kernel: one byte per parameter. */
int fdn_l2_sumsq(const float* w, const uint8_t* is_kernel, int64_t n, float* out, void* stream);

/* Keras Adam on one flat buffer, with the L2-regulariser gradient folded in:
 * g' = g + l2_grad_scale * w (kernels only); m,v EMA; w -= lr_t * m / (sqrt(v) + eps).
 * lr_t = lr*sqrt(1-b2^t)/(1-b1^t) is computed by the caller.  If l2_scale_dev != NULL the scale is
 * l2_grad_scale * l2_scale_dev[0], read on the device (data-parallel: the global batch size rides in the
 * all-reduced gradient buffer, so no host synchronisation is needed).  TrainerController.py:73,225. */
int fdn_adam_step(float* w, const float* g, float* m, float* v, const uint8_t* is_kernel, int64_t n,
                  float lr_t, float b1, float b2, float eps, float l2_grad_scale, const float* l2_scale_dev,
                  float* sumsq_partials, void* stream);
/* If sumsq_partials != NULL (FDN_ADAM_PARTIALS floats) the step also leaves per-block sums of the UPDATED
 * kernel parameters' squares there; fdn_sum_partials adds them up in a fixed order (deterministic):
 * 5e-7 * that sum is the regulariser value of the next step's loss, so fdn_l2_sumsq need not stream the
 * parameters again.  src/Network/TrainerController.py:129-141. */
#define FDN_ADAM_PARTIALS 2048
int fdn_sum_partials(const float* partials, int n, float* out, void* stream);
/* The same FDN_ADAM_PARTIALS per-block sums for parameters no fdn_adam_step has touched yet (first step, after loading a
 * checkpoint): the multi-block form of fdn_l2_sumsq, which needs no scratch and runs as ONE block.  Follow with fdn_sum_partials. */
int fdn_l2_sumsq_partials(const float* w, const uint8_t* is_kernel, int64_t n, float* sumsq_partials, void* stream);

/* ------------------------------------------------------------------------------------------------
 * bf16 activation path (BASELINE.json configs[3]: patch 32, res x4, bf16).  The reference has no bf16
 * mode (TF runs fp32); semantics = the fp32 entry points above with every activation / activation-gradient
 * tensor stored as bfloat16 (uint16_t bit patterns, round-to-nearest-even on store), fp32 accumulation,
 * fp32 parameters, gradients of parameters and optimizer state.
 * ------------------------------------------------------------------------------------------------ */

/* fp32 Keras kernel (27,64,64) -> bf16 operand streams (27*64*64 uint16_t each) for the two entry points below. */
int fdn_pack_conv64_weights_bf16(const float* w, uint16_t* wp_fwd, uint16_t* wp_dgrad, void* stream);
/* The same for n_layers kernels in ONE launch (after every optimizer step, TrainerController.py:225): layer i at
 * w_base + w_offsets[i] (DEVICE array of float offsets), its streams at packs + i * 2 * 27*64*64 (forward, then dgrad). */
int fdn_pack_conv64_weights_bf16_batch(const float* w_base, const int64_t* w_offsets, int n_layers, uint16_t* packs,
                                       void* stream);

/* fdn_conv3d_fwd for (Cin,Cout,K) = (64,64,3) with bf16 x / residual / y.  SR4DFlowNet.py:93-120. */
int fdn_conv64_fwd_bf16(const uint16_t* x, const uint16_t* wpack, const float* bias, const uint16_t* residual,
                        uint16_t* y, int N, int D, int H, int W, int act, float alpha, void* stream);

/* fdn_conv3d_dgrad_fused / fdn_fold_halo_border with bf16 dz / skip / y_prev / dz_prev; the padded scratch
 * dxpad (N,D+2,H+2,W+2,64) stays fp32 (only positions the border fold reads are written). */
int fdn_conv64_dgrad_fused_bf16(const uint16_t* dz, const uint16_t* wpack, float* dxpad, const uint16_t* skip,
                                const uint16_t* y_prev, int act, float alpha, uint16_t* dz_prev, int N, int D, int H,
                                int W, void* stream);
int fdn_fold_halo_border_bf16(const float* dxpad0, const float* dxpad1, const float* dxpad2, int nsrc,
                              const uint16_t* skip, const uint16_t* y_prev, int act, float alpha, uint16_t* dz_prev,
                              int N, int D, int H, int W, void* stream);
/* Sign masks.  The activation gradient of SR4DFlowNet.py:113,118 (LeakyReLU) / :17-25 (ReLU) needs one bit of the producer's
 * output: y > 0.  fdn_conv64_fwd_bf16_mask is fdn_conv64_fwd_bf16 that also writes, when y_mask != NULL, 64 bits per voxel
 * -- four uint16_t words [voxel][cout / 16], bit c % 16 = (the stored bf16 y[c] > 0) -- i.e. (N*D*H*W*4) uint16_t;
 * fdn_conv64_dgrad_fused_bf16_mask reads that mask for act' instead of the 128-B rows of y_prev when y_mask != NULL
 * (y_prev may then be NULL; act must be RELU or LEAKY).  Same results bit for bit; at (4,128^3) the launch is 0.14 ms
 * (8 %) shorter -- an epilogue operand of these kernels costs what streaming its 1.07 GB costs.  The border fold
 * (fdn_fold_halo_border_bf16) still takes y_prev itself: it touches the surface voxels only. */
int fdn_conv64_fwd_bf16_mask(const uint16_t* x, const uint16_t* wpack, const float* bias, const uint16_t* residual,
                             uint16_t* y, uint16_t* y_mask, int N, int D, int H, int W, int act, float alpha,
                             void* stream);
int fdn_conv64_dgrad_fused_bf16_mask(const uint16_t* dz, const uint16_t* wpack, float* dxpad, const uint16_t* skip,
                                     const uint16_t* y_prev, const uint16_t* y_mask, int act, float alpha,
                                     uint16_t* dz_prev, int N, int D, int H, int W, void* stream);
/* fdn_conv64_dgrad_fused_multi for bf16 activations: the fused dgrad of up to three 64->64 layers that share their input as ONE launch
 * (+ one fdn_fold_halo_border_bf16), the sum over the sources kept in the fp32 accumulators -- the chained launches round the running sum
 * to bf16 after every source, so this form is the more accurate one (it differs from the chain by bf16 roundings of the partial sums).
 * Every grid (all kernel variants of the bf16 path walk the sources).  y_prev or y_mask or neither. */
int fdn_conv64_dgrad_fused_bf16_multi(const uint16_t* const* dz, const uint16_t* const* wpack, int nsrc, float* dxpad,
                                      const uint16_t* skip, const uint16_t* y_prev, const uint16_t* y_mask, int act, float alpha,
                                      uint16_t* dz_prev, int N, int D, int H, int W, void* stream);

/* The remaining entry points of the path in bf16 storage: same contracts as the fp32 functions of the same
 * name.  Parameters, parameter gradients, the 64->1 heads' output (the prediction, `y` of Cout=1) and its
 * gradient (`dz` of Cout=1) stay fp32. */
int fdn_input_features_bf16(const float* u, const float* v, const float* w, const float* mu, const float* mv,
                            const float* mw, uint16_t* phase, uint16_t* pc, int64_t nvox, void* stream);
int fdn_conv3d_fwd_bf16(const uint16_t* x, const uint16_t* x2, const float* w, const uint16_t* wpack, const float* bias,
                        const uint16_t* residual, void* y, int N, int D, int H, int W, int Cin, int Cout, int K, int ldy,
                        int y_coff, int act, float alpha, void* stream);
size_t fdn_conv3d_wgrad_bf16_workspace_bytes(int N, int D, int H, int W, int Cin, int Cout, int K);
int fdn_conv3d_wgrad_bf16(const uint16_t* x, const uint16_t* x2, const void* dz, float* dw, float* dbias, void* workspace,
                          size_t workspace_bytes, int N, int D, int H, int W, int Cin, int Cout, int K, int lddz,
                          int dz_coff, void* stream);
/* fdn_conv3d_wgrad_batch for bf16 activations: the weight gradients (TrainerController.py:223) of several 64->64 3x3x3 layers of
 * ONE grid -- x, dz, dw, dbias: HOST arrays of n_layers device pointers (dbias or its entries may be NULL) -- in chunks of up to
 * seven layers per launch (+ one reduction per chunk) where the grid allows, else layer by layer; results equal
 * fdn_conv3d_wgrad_bf16 per layer to fp32 rounding (a different split of the voxel sum). */
size_t fdn_conv3d_wgrad_bf16_batch_workspace_bytes(int n_layers, int N, int D, int H, int W);
int fdn_conv3d_wgrad_bf16_batch(const uint16_t* const* x, const uint16_t* const* dz, float* const* dw, float* const* dbias,
                                int n_layers, void* workspace, size_t workspace_bytes, int N, int D, int H, int W,
                                void* stream);
int fdn_conv_cout1_dgrad_folded_bf16(const float* dz, const float* w, const uint16_t* y_prev, int act, float alpha,
                                     uint16_t* dz_prev, float* dbias_prev, void* workspace, size_t workspace_bytes,
                                     int N, int D, int H, int W, int lddz, int dz_coff, void* stream);
/* ... with the sign mask of y_prev (written by fdn_conv64_fwd_bf16_mask for the 64->64 head conv that produced it) read for act'
 * instead of its 128-B rows when y_mask != NULL; same results bit for bit. */
int fdn_conv_cout1_dgrad_folded_bf16_mask(const float* dz, const float* w, const uint16_t* y_prev, const uint16_t* y_mask,
                                          int act, float alpha, uint16_t* dz_prev, float* dbias_prev, void* workspace,
                                          size_t workspace_bytes, int N, int D, int H, int W, int lddz, int dz_coff,
                                          void* stream);
int fdn_conv1x1_dgrad_bf16(const uint16_t* dz, const float* w, const uint16_t* ya, const uint16_t* yb, uint16_t* dxa,
                           uint16_t* dxb, int64_t nvox, void* stream);
int fdn_upsample_trilinear_fwd_bf16(const uint16_t* x, uint16_t* y, int N, int D, int H, int W, int C, int R, void* stream);
int fdn_upsample_trilinear_bwd_bf16(const uint16_t* dy, const uint16_t* y_prev, int act, float alpha, uint16_t* dx, int N,
                                    int D, int H, int W, int C, int R, void* stream);

#ifdef __cplusplus
}
#endif
#endif

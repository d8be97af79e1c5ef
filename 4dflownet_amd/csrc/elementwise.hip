// HBM-bound pieces of the train step: input features, halo fold (MirrorPadGrad) fused with gradient fan-in and
// activation gradient, trilinear upsample fwd/bwd, masked-MSE loss + relative-error metric + dPred, L2 sum, Adam.
// All of them stream 16-B vectors with 16 lanes per 256-B channel row.
#include "fdn_common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// src/Network/SR4DFlowNet.py:10-15
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void input_features_kernel(const float* __restrict__ u, const float* __restrict__ v, const float* __restrict__ w,
                                      const float* __restrict__ mu, const float* __restrict__ mv,
                                      const float* __restrict__ mw, T* __restrict__ phase, T* __restrict__ pc,
                                      int64_t nvox) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvox; i += (int64_t)gridDim.x * blockDim.x) {
        const float a = u[i], b = v[i], c = w[i];
        const float speed = sqrtf(a * a + b * b + c * c);
        const float ma = mu[i], mb = mv[i], mc = mw[i];
        const float mag = sqrtf(ma * ma + mb * mb + mc * mc);
        fdn_st1(phase + i * 3 + 0, a); fdn_st1(phase + i * 3 + 1, b); fdn_st1(phase + i * 3 + 2, c);
        fdn_st1(pc + i * 3 + 0, mag * speed); fdn_st1(pc + i * 3 + 1, mag); fdn_st1(pc + i * 3 + 2, speed);
    }
}

// ---------------------------------------------------------------------------------------------
// fold: adjoint of the SYMMETRIC p=1 pad (MirrorPadGrad) + fan-in of up to 3 padded gradients + skip + act'
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fold_halo_kernel(const float* __restrict__ s0, const float* __restrict__ s1,
                                                         const float* __restrict__ s2, int nsrc,
                                                         const float* __restrict__ skip, const float* __restrict__ yprev,
                                                         int act, float alpha, float* __restrict__ out, int N, int D, int H,
                                                         int W, int C4) {
    const int64_t total = (int64_t)N * D * H * W * C4;
    const int PH = H + 2, PW = W + 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        int64_t v = i / C4;
        const int w = (int)(v % W); v /= W;
        const int h = (int)(v % H); v /= H;
        const int d = (int)(v % D);
        const int n = (int)(v / D);
        // padded indices that clamp onto (d,h,w): p = i+1, plus 0 when i==0, plus dim+1 when i==dim-1
        int pd[3], ph[3], pw[3];
        int nd = 0, nh = 0, nw = 0;
        pd[nd++] = d + 1; if (d == 0) pd[nd++] = 0; if (d == D - 1) pd[nd++] = D + 1;
        ph[nh++] = h + 1; if (h == 0) ph[nh++] = 0; if (h == H - 1) ph[nh++] = H + 1;
        pw[nw++] = w + 1; if (w == 0) pw[nw++] = 0; if (w == W - 1) pw[nw++] = W + 1;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int a = 0; a < nd; ++a)
            for (int b = 0; b < nh; ++b)
                for (int c = 0; c < nw; ++c) {
                    const int64_t off = (((((int64_t)n * (D + 2) + pd[a]) * PH + ph[b]) * PW + pw[c]) * C4 + c4);
                    acc += ((const f32x4*)s0)[off];
                    if (nsrc > 1) acc += ((const f32x4*)s1)[off];
                    if (nsrc > 2) acc += ((const f32x4*)s2)[off];
                }
        if (skip) acc += ((const f32x4*)skip)[i];
        if (yprev) {
            const f32x4 y = ((const f32x4*)yprev)[i];
            acc.x *= fdn_act_grad(y.x, act, alpha); acc.y *= fdn_act_grad(y.y, act, alpha);
            acc.z *= fdn_act_grad(y.z, act, alpha); acc.w *= fdn_act_grad(y.w, act, alpha);
        }
        ((f32x4*)out)[i] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// upsample3d (src/Network/SR4DFlowNet.py:53-90): trilinear, align_corners=True.
// coefficient rule of tf resize_bilinear(align_corners): in = out * scale (fp32), scale = (n-1)/(nR-1).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void lerp_coeff(int o, float scale, int n, int& lo, int& hi, float& f) {
    const float src = (float)o * scale;
    const float fl = floorf(src);
    lo = (int)fl;
    hi = min(lo + 1, n - 1);
    f = src - fl;
}

// block = one output row (n, od, oh): the d / h interpolation coefficients are block-uniform; thread = one 16-B vector
// (4 fp32 / 8 bf16 channels) of one output voxel, consecutive threads write consecutive vectors (32-bit index math only).
// STAGED (round 6): the four input rows (d0|d1, h0|h1) of the output row go through LDS once, as coalesced 16-B loads, and the
// eight corner vectors of every output vector come from there -- the direct form asks the vector cache for eight 16-B loads per
// 16 B written (1.8 GB of L1 traffic for the 226 MB of the cfg2 upsample: L1-bound at 0.078 ms, 2.9 TB/s written); same
// arithmetic in the same order, bit-identical.  Rows too long for the LDS take the direct form.
template <typename T, bool STAGED>
__global__ __launch_bounds__(256) void upsample_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int D,
                                                            int H, int W, int CV, int R, float sd, float sh, float sw) {
    constexpr int E = FdnVec<T>::E;
    extern __shared__ __attribute__((aligned(16))) float rowbuf[];      // STAGED: [4][W * CV] 16-B vectors
    const int OD = D * R, OH = H * R, OW = W * R;
    for (int row = blockIdx.x; row < N * OD * OH; row += gridDim.x) {
        const int oh = row % OH;
        const int t = row / OH;
        const int od = t % OD, n = t / OD;
        int d0, d1, h0, h1;
        float fd, fh;
        lerp_coeff(od, sd, D, d0, d1, fd);
        lerp_coeff(oh, sh, H, h0, h1, fh);
        const T* r00 = x + (((int64_t)n * D + d0) * H + h0) * W * CV * E;
        const T* r01 = x + (((int64_t)n * D + d0) * H + h1) * W * CV * E;
        const T* r10 = x + (((int64_t)n * D + d1) * H + h0) * W * CV * E;
        const T* r11 = x + (((int64_t)n * D + d1) * H + h1) * W * CV * E;
        if constexpr (STAGED) {
            const int nv = W * CV;
            __syncthreads();                               // the previous row's corner reads are done
            f32x4* dst = (f32x4*)rowbuf;
            for (int i = threadIdx.x; i < nv; i += blockDim.x) {
                const f32x4 a = ((const f32x4*)r00)[i], b = ((const f32x4*)r01)[i], c = ((const f32x4*)r10)[i], d = ((const f32x4*)r11)[i];
                dst[i] = a; dst[nv + i] = b; dst[2 * nv + i] = c; dst[3 * nv + i] = d;
            }
            __syncthreads();
            r00 = (const T*)rowbuf; r01 = r00 + nv * E; r10 = r01 + nv * E; r11 = r10 + nv * E;
        }
        T* yr = y + (int64_t)row * OW * CV * E;
        for (int i = threadIdx.x; i < OW * CV; i += blockDim.x) {
            const int ow = i / CV, cv = i - ow * CV;
            int w0, w1;
            float fw;
            lerp_coeff(ow, sw, W, w0, w1, fw);
            const int o0 = (w0 * CV + cv) * E, o1 = (w1 * CV + cv) * E;
            float c000[E], c001[E], c010[E], c011[E], c100[E], c101[E], c110[E], c111[E];
            FdnVec<T>::ld(r00 + o0, c000); FdnVec<T>::ld(r00 + o1, c001);
            FdnVec<T>::ld(r01 + o0, c010); FdnVec<T>::ld(r01 + o1, c011);
            FdnVec<T>::ld(r10 + o0, c100); FdnVec<T>::ld(r10 + o1, c101);
            FdnVec<T>::ld(r11 + o0, c110); FdnVec<T>::ld(r11 + o1, c111);
            // innermost (z) first, then y, then x -- the order of the reference's two resize passes
            float o[E];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float a0 = c000[e] + (c001[e] - c000[e]) * fw, a1 = c010[e] + (c011[e] - c010[e]) * fw;
                const float b0 = c100[e] + (c101[e] - c100[e]) * fw, b1 = c110[e] + (c111[e] - c110[e]) * fw;
                const float aa = a0 + (a1 - a0) * fh, bb = b0 + (b1 - b0) * fh;
                o[e] = aa + (bb - aa) * fd;
            }
            FdnVec<T>::st(yr + (int64_t)i * E, o);
        }
    }
}

// weight of output index o on input index i along one axis (0 if o does not touch i)
__device__ __forceinline__ float axis_weight(int o, int i, float scale, int n) {
    int lo, hi;
    float f;
    lerp_coeff(o, scale, n, lo, hi, f);
    float wgt = 0.f;
    if (lo == i) wgt += 1.f - f;
    if (hi == i) wgt += f;
    return wgt;
}

// candidate output range [o0,o1] that can touch input i: o*scale in (i-1, i+1)
__device__ __forceinline__ void axis_range(int i, float inv_scale, int m, int& o0, int& o1) {
    o0 = max(0, (int)floorf((float)(i - 1) * inv_scale) - 1);
    o1 = min(m - 1, (int)ceilf((float)(i + 1) * inv_scale) + 1);
}

// Adjoint of the trilinear upsample, block = HB consecutive LOW-res rows (n, d, h0 .. h0 + HB - 1).  Phase A folds the high-res rows
// that touch them (block-uniform weights wd * wh per low-res row) into HB high-res-wide rows in LDS: coalesced 16-B reads of dy, eight
// rows requested before the first use, every loaded vector accumulated into all HB rows (weight 0 where it does not touch: exact).  A
// high-res row touches two low-res rows per axis, so with HB = 1 (rounds 1-5) every dy row went through the vector cache four times
// (1 GB of L2 -> L1 traffic for the 226 MB of cfg2: 0.071 ms); with HB = 2 it is three times.  Phase B folds each row along w, applies
// act'(y_prev), stores.  Per low-res row the non-zero terms are added in the same order as before: bit-identical for every HB.
template <typename T, int HB>
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ yprev,
                                                            int act, float alpha, T* __restrict__ dx, int N, int D,
                                                            int H, int W, int CV, int R, float sd, float sh, float sw) {
    constexpr int E = FdnVec<T>::E;
    extern __shared__ __attribute__((aligned(16))) float rowbuf[];      // [HB][OW][CV * E]
    constexpr int kCandD = 32, kCandH = 64;                             // candidate rows per axis (host checks R and HB against them)
    constexpr int kUpPairs = HB == 1 ? 1024 : 512;
    __shared__ int s_row[kUpPairs];
    __shared__ float s_wgt[kUpPairs * HB];
    __shared__ int s_od[kCandD], s_oh[kCandH], s_nd, s_nh;
    __shared__ float s_wd[kCandD], s_wh[kCandH * HB];
    const int OD = D * R, OH = H * R, OW = W * R;
    const int C = CV * E;
    const float isd = sd > 0.f ? 1.f / sd : 0.f, ish = sh > 0.f ? 1.f / sh : 0.f, isw = sw > 0.f ? 1.f / sw : 0.f;
    // XCD-aware block order: block ids are dealt round-robin to the 8 XCDs; an XCD takes one contiguous run of low-res row blocks,
    // so the high-res rows shared by neighbouring blocks are fetched into ITS L2 once (plain order: 3.8x dy from HBM)
    const int nhblk = (H + HB - 1) / HB;
    const int nrows = N * D * nhblk;
    for (int it = blockIdx.x; it < nrows; it += gridDim.x) {
        const int blk0 = it - (int)blockIdx.x, span = min((int)gridDim.x, nrows - blk0);     // blocks of this sweep
        const int q = span >> 3, rem = span & 7, xc = blockIdx.x & 7;
        const int row = blk0 + xc * q + (xc < rem ? xc : rem) + ((int)blockIdx.x >> 3);
        const int h0 = (row % nhblk) * HB;
        const int nhb = min(HB, H - h0);
        const int t = row / nhblk;
        const int d = t % D, n = t / D;
        int od0, od1, oh0, oh1, tmp;
        axis_range(d, isd, OD, od0, od1);
        axis_range(h0, ish, OH, oh0, tmp);
        axis_range(h0 + nhb - 1, ish, OH, tmp, oh1);
        if (sd == 0.f) { od0 = 0; od1 = OD - 1; }
        if (sh == 0.f) { oh0 = 0; oh1 = OH - 1; }
        __syncthreads();                                   // rowbuf and the pair list of the previous block fully consumed
        // The high-res rows (od, oh) with a non-zero weight on one of these low-res rows, as a compact block-uniform list: the fold
        // below then has no data-dependent branch around its loads (a `continue` per candidate row made every load its own round
        // trip: 103 us for 280 MB at cfg2).  Wave 0: the depth candidates od0.. (lanes 0..31), wave 1: the height candidates oh0..;
        // ballot + prefix count compacts the ones with a non-zero weight in index order (deterministic).
        if (threadIdx.x < 128) {
            const int lane = threadIdx.x & 63;
            const bool dl = threadIdx.x < 64;
            const int c = dl ? od0 + lane : oh0 + lane;
            float wgt[HB];
            bool any = false;
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) {
                wgt[hb] = 0.f;
                if (dl) { if (hb == 0 && lane < kCandD && c <= od1) wgt[0] = axis_weight(c, d, sd, D); }
                else if (hb < nhb && c <= oh1) wgt[hb] = axis_weight(c, h0 + hb, sh, H);
                any = any || wgt[hb] != 0.f;
            }
            const unsigned long long m = __ballot(any);
            if (any) {
                const int pos = __popcll(m & ((1ull << lane) - 1ull));
                if (dl) { s_od[pos] = c; s_wd[pos] = wgt[0]; }
                else {
                    s_oh[pos] = c;
#pragma unroll
                    for (int hb = 0; hb < HB; ++hb) s_wh[pos * HB + hb] = wgt[hb];
                }
            }
            if (lane == 0) { if (dl) s_nd = __popcll(m); else s_nh = __popcll(m); }
        }
        __syncthreads();
        const int nd_ = s_nd, nh_ = s_nh, npairs = nd_ * nh_;
        for (int k = threadIdx.x; k < npairs; k += blockDim.x) {
            const int a = k / nh_, b = k - a * nh_;
            s_row[k] = (n * OD + s_od[a]) * OH + s_oh[b];
#pragma unroll
            for (int hb = 0; hb < HB; ++hb) s_wgt[k * HB + hb] = s_wd[a] * s_wh[b * HB + hb];
        }
        __syncthreads();
        constexpr int U = 8;                               // high-res rows requested before the first use (16 measured slower: 0.070 vs 0.057 ms at cfg2)
        for (int i = threadIdx.x; i < OW * CV; i += blockDim.x) {
            float acc[HB][E];
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                for (int e = 0; e < E; ++e) acc[hb][e] = 0.f;
            for (int k0 = 0; k0 < npairs; k0 += U) {
                float g[U][E];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int k = min(k0 + u, npairs - 1);         // tail: re-read the last row with weight 0 (cache hit)
                    FdnVec<T>::ld(dy + ((int64_t)s_row[k] * OW * CV + i) * E, g[u]);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int k = min(k0 + u, npairs - 1);
#pragma unroll
                    for (int hb = 0; hb < HB; ++hb) {
                        const float wv = k0 + u < npairs ? s_wgt[k * HB + hb] : 0.f;
#pragma unroll
                        for (int e = 0; e < E; ++e) acc[hb][e] += g[u][e] * wv;
                    }
                }
            }
#pragma unroll
            for (int hb = 0; hb < HB; ++hb)
#pragma unroll
                for (int e = 0; e < E; ++e) rowbuf[(hb * OW * CV + i) * E + e] = acc[hb][e];
        }
        __syncthreads();
        for (int j = threadIdx.x; j < nhb * W * CV; j += blockDim.x) {
            const int hb = j / (W * CV), i = j - hb * (W * CV);
            const int w = i / CV, cv = i - w * CV;
            int ow0, ow1;
            axis_range(w, isw, OW, ow0, ow1);
            if (sw == 0.f) { ow0 = 0; ow1 = OW - 1; }
            const float* rb = rowbuf + hb * OW * C;
            float acc[E];
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] = 0.f;
            for (int ow = ow0; ow <= ow1; ++ow) {
                const float ww = axis_weight(ow, w, sw, W);
                if (ww == 0.f) continue;
#pragma unroll
                for (int e = 0; e < E; ++e) acc[e] += rb[ow * C + cv * E + e] * ww;
            }
            const int64_t o = ((((int64_t)n * D + d) * H + h0 + hb) * W * CV + i) * E;
            if (yprev) {
                float yv[E];
                FdnVec<T>::ld(yprev + o, yv);
#pragma unroll
                for (int e = 0; e < E; ++e) acc[e] *= fdn_act_grad(yv[e], act, alpha);
            }
            FdnVec<T>::st(dx + o, acc);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// loss / metric: TrainerController.py:84-127,152-156 ; loss_utils.py:64-103
// scratch layout per sample: [0]=sum mask [1]=sum nonfluid [2]=sum mse*mask [3]=sum mse*nf [4]=sum rel
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int wv = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wv] = v;
    __syncthreads();
    float s = 0.f;
    if (threadIdx.x == 0)
        for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s += red[k];
    return s;   // valid in thread 0
}

__global__ __launch_bounds__(256) void mask_sums_kernel(const float* __restrict__ mask, float* __restrict__ scratch, int64_t V) {
    __shared__ float red[4];
    const int n = blockIdx.y;
    float sm = 0.f, snf = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (int64_t)gridDim.x * blockDim.x) {
        const float m = mask[(int64_t)n * V + i];
        sm += m;
        snf += m < 0.5f ? 1.f : 0.f;
    }
    const float a = block_sum(sm, red);
    const float b = block_sum(snf, red);
    if (threadIdx.x == 0) { atomicAdd(&scratch[n * 8 + 0], a); atomicAdd(&scratch[n * 8 + 1], b); }
}

__global__ __launch_bounds__(256) void loss_main_kernel(const float* __restrict__ pred, const float* __restrict__ uh,
                                                         const float* __restrict__ vh, const float* __restrict__ wh,
                                                         const float* __restrict__ mask, float* __restrict__ scratch,
                                                         float* __restrict__ dpred, int64_t V) {
    __shared__ float red[4];
    const int n = blockIdx.y;
    const float inv_f = 1.f / (scratch[n * 8 + 0] + 1.f);
    const float inv_nf = 1.f / (scratch[n * 8 + 1] + 1.f);
    float sf = 0.f, snf = 0.f, srel = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t g = (int64_t)n * V + i;
        const float tu = uh[g], tv = vh[g], tw = wh[g];
        const float du = pred[g * 3] - tu, dv = pred[g * 3 + 1] - tv, dw = pred[g * 3 + 2] - tw;
        const float m = mask[g];
        const float nf = m < 0.5f ? 1.f : 0.f;
        const float mse = du * du + dv * dv + dw * dw;
        sf += mse * m;
        snf += mse * nf;
        // relative error (loss_utils.py:64-103); rintf == round-half-to-even == tf.round
        const float diff = sqrtf(mse);
        const float actual = sqrtf(tu * tu + tv * tv + tw * tw);
        float rel = diff / (actual + 1e-5f);
        rel = fminf(fmaxf(rel, 0.f), 1.f);
        float corr = actual != 0.f ? rel : diff;
        corr = rintf(corr * 1e4f) / 1e4f;
        if (m == 1.0f) srel += corr;
        if (dpred) {
            const float wgt = 2.f * (m * inv_f + nf * inv_nf);
            dpred[g * 3] = du * wgt; dpred[g * 3 + 1] = dv * wgt; dpred[g * 3 + 2] = dw * wgt;
        }
    }
    const float a = block_sum(sf, red);
    const float b = block_sum(snf, red);
    const float c = block_sum(srel, red);
    // per-block partials, summed by loss_finalize_kernel in block order: the reported loss / metric is run-to-run identical
    // (the mask counts above are integers < 2^24, so their atomic sums are exact in any order)
    if (threadIdx.x == 0) {
        float* part = scratch + (size_t)gridDim.y * 8 + ((size_t)n * gridDim.x + blockIdx.x) * 3;
        part[0] = a; part[1] = b; part[2] = c;
    }
}

// one wave per sample: lane t sums the partials of blocks t, t + 64, ..., then a shuffle tree -- a fixed order, so the reported loss /
// metric stays run-to-run identical (one THREAD per sample walked the 256 partials as a chain of dependent loads: 24 us)
__global__ __launch_bounds__(64) void loss_finalize_kernel(const float* __restrict__ scratch, float* __restrict__ out, int N, int nblk) {
    const int n = blockIdx.x;
    const float* part = scratch + (size_t)N * 8 + (size_t)n * nblk * 3;
    float sf = 0.f, sn = 0.f, sr = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 64) { sf += part[b * 3]; sn += part[b * 3 + 1]; sr += part[b * 3 + 2]; }
    for (int o = 32; o > 0; o >>= 1) { sf += __shfl_down(sf, o, 64); sn += __shfl_down(sn, o, 64); sr += __shfl_down(sr, o, 64); }
    if (threadIdx.x) return;
    const float sm = scratch[n * 8 + 0], snf = scratch[n * 8 + 1];
    out[n * 4 + 0] = sf / (sm + 1.f) + sn / (snf + 1.f);
    out[n * 4 + 1] = sr / (sm + 1.f) * 100.f;
    out[n * 4 + 2] = sm;
    out[n * 4 + 3] = snf;
}

// ONE block (fixed summation order, no atomics): this pass only runs when the parameters changed outside the optimizer
// (first step, load_weights); afterwards the Adam kernel supplies the sum
__global__ __launch_bounds__(1024) void l2_sumsq_kernel(const float* __restrict__ w, const uint8_t* __restrict__ isk,
                                                         int64_t n, float* __restrict__ out) {
    __shared__ float red[16];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;           // four independent chains: the loads of a trip overlap
    int64_t i = threadIdx.x;
    for (; i + 3 * 1024 < n; i += 4 * 1024) {
        const float a0 = isk[i] ? w[i] : 0.f, a1 = isk[i + 1024] ? w[i + 1024] : 0.f;
        const float a2 = isk[i + 2048] ? w[i + 2048] : 0.f, a3 = isk[i + 3072] ? w[i + 3072] : 0.f;
        s0 += a0 * a0; s1 += a1 * a1; s2 += a2 * a2; s3 += a3 * a3;
    }
    for (; i < n; i += 1024)
        if (isk[i]) s0 += w[i] * w[i];
    const float s = (s0 + s1) + (s2 + s3);
    const float a = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = a;
}

// sumsq != null: block b also writes sum(w_new^2) over its kernel (non-bias) elements to sumsq[b] (fixed order: the
// regulariser value of the NEXT step's loss, so fdn_l2_sumsq does not have to stream the parameters again).
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, const uint8_t* __restrict__ isk, int64_t n, float lr_t,
                                                   float b1, float b2, float eps, float l2s_host,
                                                   const float* __restrict__ l2s_dev, float* __restrict__ sumsq) {
    const float l2s = l2s_dev ? l2s_host * l2s_dev[0] : l2s_host;
    float ss = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float wi = w[i];
        float gi = g[i];
        const bool k = isk[i] != 0;
        if (k) gi += l2s * wi;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float wn = wi - lr_t * mi / (sqrtf(vi) + eps);
        w[i] = wn;
        if (k) ss += wn * wn;
    }
    if (sumsq) {
        __shared__ float red[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_down(ss, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
        __syncthreads();
        if (threadIdx.x == 0) sumsq[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
    }
}

// The same per-block sums the Adam kernel leaves behind (identical blocking: block b covers the elements b * 256 + t + k * grid * 256),
// for the steps that have no Adam step before them (first step, load_weights): FDN_ADAM_PARTIALS blocks stream the parameters at the
// rate of the chip instead of one block at 9 GB/s (l2_sumsq_kernel: 1.87 ms for the 13.4 MB of cfg2)
__global__ __launch_bounds__(256) void l2_sumsq_partials_kernel(const float* __restrict__ w, const uint8_t* __restrict__ isk, int64_t n,
                                                                 float* __restrict__ sumsq) {
    float ss = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        if (isk[i]) ss += w[i] * w[i];
    __shared__ float red[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_down(ss, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    if (threadIdx.x == 0) sumsq[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = sum of n partials, one block, fixed order (deterministic)
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = (red[0] + red[1]) + (red[2] + red[3]);
}

int grid_for(int64_t items, int cap = 4096) {
    int64_t b = (items + 255) / 256;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

float axis_scale(int n, int R) {
    const int m = n * R;
    return m > 1 ? (float)(n - 1) / (float)(m - 1) : 0.f;
}

}  // namespace

template <typename T>
static int input_features_t(const float* u, const float* v, const float* w, const float* mu, const float* mv, const float* mw,
                            T* phase, T* pc, int64_t nvox, void* stream) {
    FDN_REQUIRE(u && v && w && mu && mv && mw && phase && pc && nvox > 0, "fdn_input_features: NULL argument or nvox<=0");
    hipLaunchKernelGGL(input_features_kernel<T>, dim3(grid_for(nvox)), dim3(256), 0, (hipStream_t)stream, u, v, w, mu, mv, mw,
                       phase, pc, nvox);
    FDN_CHECK_LAUNCH("input_features_kernel");
    return FDN_OK;
}
extern "C" int fdn_input_features(const float* u, const float* v, const float* w, const float* mu, const float* mv,
                                  const float* mw, float* phase, float* pc, int64_t nvox, void* stream) {
    return input_features_t<float>(u, v, w, mu, mv, mw, phase, pc, nvox, stream);
}
extern "C" int fdn_input_features_bf16(const float* u, const float* v, const float* w, const float* mu, const float* mv,
                                       const float* mw, uint16_t* phase, uint16_t* pc, int64_t nvox, void* stream) {
    return input_features_t<uint16_t>(u, v, w, mu, mv, mw, phase, pc, nvox, stream);
}

extern "C" int fdn_fold_halo(const float* dxpad0, const float* dxpad1, const float* dxpad2, int nsrc, const float* skip,
                             const float* y_prev, int act, float alpha, float* dz_prev, int N, int D, int H, int W, int C,
                             void* stream) {
    FDN_REQUIRE(dxpad0 && dz_prev, "fdn_fold_halo: NULL argument");
    FDN_REQUIRE(nsrc >= 1 && nsrc <= 3 && (nsrc < 2 || dxpad1) && (nsrc < 3 || dxpad2), "fdn_fold_halo: bad nsrc %d", nsrc);
    FDN_REQUIRE(C % 4 == 0 && N > 0 && D > 0 && H > 0 && W > 0, "fdn_fold_halo: bad dims (C must be a multiple of 4)");
    const int64_t total = (int64_t)N * D * H * W * (C / 4);
    hipLaunchKernelGGL(fold_halo_kernel, dim3(grid_for(total, 8192)), dim3(256), 0, (hipStream_t)stream, dxpad0, dxpad1,
                       dxpad2, nsrc, skip, y_prev, act, alpha, dz_prev, N, D, H, W, C / 4);
    FDN_CHECK_LAUNCH("fold_halo_kernel");
    return FDN_OK;
}

template <typename T>
static int upsample_fwd_t(const T* x, T* y, int N, int D, int H, int W, int C, int R, void* stream) {
    constexpr int E = 16 / (int)sizeof(T);
    FDN_REQUIRE(x && y && C % E == 0 && R >= 1 && N > 0 && D > 0 && H > 0 && W > 0, "fdn_upsample_trilinear_fwd: bad argument");
    const int64_t rows = (int64_t)N * D * R * H * R;
    FDN_REQUIRE(rows < (1ll << 31), "fdn_upsample_trilinear_fwd: too many rows");
    const size_t lds = (size_t)4 * W * C * sizeof(T);
    if (lds <= 64 * 1024) {
        if (lds > 48 * 1024)
            if (int rc = fdn_func_max_lds((const void*)upsample_fwd_kernel<T, true>, 64 * 1024, "upsample_fwd")) return rc;
        hipLaunchKernelGGL((upsample_fwd_kernel<T, true>), dim3((unsigned)(rows < 262144 ? rows : 262144)), dim3(256), lds, (hipStream_t)stream,
                           x, y, N, D, H, W, C / E, R, axis_scale(D, R), axis_scale(H, R), axis_scale(W, R));
    } else {
        hipLaunchKernelGGL((upsample_fwd_kernel<T, false>), dim3((unsigned)(rows < 262144 ? rows : 262144)), dim3(256), 0, (hipStream_t)stream,
                           x, y, N, D, H, W, C / E, R, axis_scale(D, R), axis_scale(H, R), axis_scale(W, R));
    }
    FDN_CHECK_LAUNCH("upsample_fwd_kernel");
    return FDN_OK;
}
FDN_HOOK_VAR(int, fdn_upsample_bwd_hb, 0);        // test / bench hook: low-res rows per block of upsample_bwd_kernel (0 = planner)
#ifdef FDN_TEST_HOOKS
extern "C" int fdn_debug_set_upsample_bwd_hb(int hb) { fdn_upsample_bwd_hb = hb; return FDN_OK; }
#endif
template <typename T>
static int upsample_bwd_t(const T* dy, const T* y_prev, int act, float alpha, T* dx, int N, int D, int H, int W, int C, int R,
                          void* stream) {
    constexpr int E = 16 / (int)sizeof(T);
    FDN_REQUIRE(dy && dx && C % E == 0 && R >= 1 && N > 0 && D > 0 && H > 0 && W > 0, "fdn_upsample_trilinear_bwd: bad argument");
    const int64_t rows = (int64_t)N * D * H;
    const size_t row_lds = (size_t)W * R * C * sizeof(float);
    FDN_REQUIRE(rows * R * R < (1ll << 31) && row_lds <= 150 * 1024, "fdn_upsample_trilinear_bwd: row of %d x %d channels does not fit the LDS stage", W * R, C);
    // the adjoint compacts the high-res rows that can touch a low-res row with 32 lanes along d: a low-res row reaches
    // 2 (OD-1)/(D-1) high-res rows (align_corners scale) + the rounding margin, which exceeds 2R+1 on short axes; along h the HB rows of a
    // block share 64 lanes
    auto span = [](int n, int r, int hb) { return n > 1 ? ((hb + 1) * (n * r - 1) + (n - 2)) / (n - 1) + 3 : n * r; };
    FDN_REQUIRE(span(D, R, 1) <= 32 && span(H, R, 1) <= 32, "fdn_upsample_trilinear_bwd: R = %d on a %d x %d grid needs %d / %d candidate rows per axis "
                "(the adjoint keeps <= 32)", R, D, H, span(D, R, 1), span(H, R, 1));
    // HB = 2 low-res rows per block where the LDS rows leave room for two workgroups per CU and the candidates fit the tables, else 1
    // (measured, tools/bench_upsample.py --hb: fp32 (8,24^3) x2 0.062 / 0.057 / 0.087 ms with 1 / 2 / 4 rows -- four rows cost more
    // occupancy than their L1 traffic saves; bf16 (4,32^3) x4 0.370 / 0.302)
    auto fits = [&](int hb) { return hb * row_lds + 14 * 1024 <= 80 * 1024 && span(H, R, hb) <= 64 && span(D, R, 1) * span(H, R, hb) <= 512; };
    int hb = fits(2) ? 2 : 1;
    if (fdn_upsample_bwd_hb) {                                             // (test build: forced)
        FDN_REQUIRE(fdn_upsample_bwd_hb == 1 || fits(fdn_upsample_bwd_hb), "fdn_upsample_trilinear_bwd: %d rows per block do not fit", fdn_upsample_bwd_hb);
        hb = fdn_upsample_bwd_hb;
    }
    const size_t lds = (size_t)hb * row_lds;
    const int64_t blocks = (int64_t)N * D * ((H + hb - 1) / hb);
    const unsigned grid = (unsigned)(blocks < 65536 ? blocks : 65536);
    const float sd = axis_scale(D, R), sh = axis_scale(H, R), sw = axis_scale(W, R);
    auto launch = [&](auto tag) -> int {
        constexpr int HB = decltype(tag)::value;
        if (lds > 48 * 1024) {
            if (int rc = fdn_func_max_lds((const void*)upsample_bwd_kernel<T, HB>, 150 * 1024, "upsample_bwd")) return rc;
        }
        hipLaunchKernelGGL((upsample_bwd_kernel<T, HB>), dim3(grid), dim3(256), lds, (hipStream_t)stream, dy, y_prev, act, alpha, dx, N, D, H, W,
                           C / E, R, sd, sh, sw);
        return FDN_OK;
    };
    FDN_REQUIRE(hb == 1 || hb == 2, "fdn_upsample_trilinear_bwd: %d rows per block", hb);
    if (int rc = hb == 2 ? launch(std::integral_constant<int, 2>{}) : launch(std::integral_constant<int, 1>{})) return rc;
    FDN_CHECK_LAUNCH("upsample_bwd_kernel");
    return FDN_OK;
}
extern "C" int fdn_upsample_trilinear_fwd(const float* x, float* y, int N, int D, int H, int W, int C, int R, void* stream) {
    return upsample_fwd_t<float>(x, y, N, D, H, W, C, R, stream);
}
extern "C" int fdn_upsample_trilinear_bwd(const float* dy, const float* y_prev, int act, float alpha, float* dx, int N, int D,
                                          int H, int W, int C, int R, void* stream) {
    return upsample_bwd_t<float>(dy, y_prev, act, alpha, dx, N, D, H, W, C, R, stream);
}
extern "C" int fdn_upsample_trilinear_fwd_bf16(const uint16_t* x, uint16_t* y, int N, int D, int H, int W, int C, int R,
                                               void* stream) {
    return upsample_fwd_t<uint16_t>(x, y, N, D, H, W, C, R, stream);
}
extern "C" int fdn_upsample_trilinear_bwd_bf16(const uint16_t* dy, const uint16_t* y_prev, int act, float alpha, uint16_t* dx,
                                               int N, int D, int H, int W, int C, int R, void* stream) {
    return upsample_bwd_t<uint16_t>(dy, y_prev, act, alpha, dx, N, D, H, W, C, R, stream);
}

extern "C" int fdn_loss_metrics(const float* pred, const float* uh, const float* vh, const float* wh, const float* mask,
                                float* out, float* dpred, float* scratch, int N, int64_t V, void* stream) {
    FDN_REQUIRE(pred && uh && vh && wh && mask && out && scratch && N > 0 && V > 0, "fdn_loss_metrics: bad argument");
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(scratch, 0, (size_t)N * 8 * sizeof(float), s);
    if (e != hipSuccess) { fdn_set_error("fdn_loss_metrics: memset: %s", hipGetErrorString(e)); return FDN_ERR_HIP; }
    const int gx = grid_for(V, FDN_LOSS_BLOCKS);
    // (16 blocks per sample: with 256 the 2 x 256 x N atomics on 2 N words took longer than reading the mask -- 29 us at cfg2)
    hipLaunchKernelGGL(mask_sums_kernel, dim3(gx < 16 ? gx : 16, N), dim3(256), 0, s, mask, scratch, V);
    FDN_CHECK_LAUNCH("mask_sums_kernel");
    hipLaunchKernelGGL(loss_main_kernel, dim3(gx, N), dim3(256), 0, s, pred, uh, vh, wh, mask, scratch, dpred, V);
    FDN_CHECK_LAUNCH("loss_main_kernel");
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(N), dim3(64), 0, s, (const float*)scratch, out, N, gx);
    FDN_CHECK_LAUNCH("loss_finalize_kernel");
    return FDN_OK;
}

extern "C" int fdn_l2_sumsq(const float* w, const uint8_t* is_kernel, int64_t n, float* out, void* stream) {
    FDN_REQUIRE(w && is_kernel && out && n > 0, "fdn_l2_sumsq: bad argument");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(l2_sumsq_kernel, dim3(1), dim3(1024), 0, s, w, is_kernel, n, out);
    FDN_CHECK_LAUNCH("l2_sumsq_kernel");
    return FDN_OK;
}

extern "C" int fdn_l2_sumsq_partials(const float* w, const uint8_t* is_kernel, int64_t n, float* sumsq_partials, void* stream) {
    FDN_REQUIRE(w && is_kernel && sumsq_partials && n > 0, "fdn_l2_sumsq_partials: bad argument");
    hipLaunchKernelGGL(l2_sumsq_partials_kernel, dim3(FDN_ADAM_PARTIALS), dim3(256), 0, (hipStream_t)stream, w, is_kernel, n, sumsq_partials);
    FDN_CHECK_LAUNCH("l2_sumsq_partials_kernel");
    return FDN_OK;
}

extern "C" int fdn_adam_step(float* w, const float* g, float* m, float* v, const uint8_t* is_kernel, int64_t n, float lr_t,
                             float b1, float b2, float eps, float l2_grad_scale, const float* l2_scale_dev,
                             float* sumsq_partials, void* stream) {
    FDN_REQUIRE(w && g && m && v && is_kernel && n > 0, "fdn_adam_step: bad argument");
    // fixed grid of FDN_ADAM_PARTIALS blocks when partials are requested (blocks beyond the data write 0)
    const int grid = sumsq_partials ? FDN_ADAM_PARTIALS : grid_for(n, 2048);
    hipLaunchKernelGGL(adam_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, g, m, v, is_kernel, n,
                       lr_t, b1, b2, eps, l2_grad_scale, l2_scale_dev, sumsq_partials);
    FDN_CHECK_LAUNCH("adam_kernel");
    return FDN_OK;
}

extern "C" int fdn_sum_partials(const float* partials, int n, float* out, void* stream) {
    FDN_REQUIRE(partials && out && n > 0, "fdn_sum_partials: bad argument");
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, n, out);
    FDN_CHECK_LAUNCH("sum_partials_kernel");
    return FDN_OK;
}

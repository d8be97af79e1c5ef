// The thin layers of SR4DFlowNet (src/Network/SR4DFlowNet.py:17,20,24,40,43,46) -- 3->64 and 64->1 3x3x3 convs and
// the 1x1x1 128->64 fuse conv -- forward, input-gradient and weight-gradient.  Together they are <0.6 % of the
// network's FLOPs and are HBM/L2-bound, so they are plain coalesced VALU kernels (no MFMA reshaping):
// 256-B channel rows are always read/written by 16 lanes x 16 B or 64 lanes x 4 B.
#include "fdn_common.h"

namespace {

__device__ __forceinline__ int clampi(int v, int hi) { return min(max(v, 0), hi); }

// -------------------------------------------------------------------------------------------------
// forward 3 -> 64, k=3.  Block = 64 voxels x 4 cout-groups of 16; weights (81 x 64) in LDS.
// -------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv_cin3_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, T* __restrict__ y,
                                                             int N, int D, int H, int W, int act, float alpha) {
    __shared__ __attribute__((aligned(16))) float ws[81 * 64];
    for (int i = threadIdx.x; i < 81 * 64; i += 256) ws[i] = w[i];
    __syncthreads();
    const int64_t nvox = (int64_t)N * D * H * W;
    const int64_t v = (int64_t)blockIdx.x * 64 + (threadIdx.x >> 2);
    if (v >= nvox) return;
    const int q = threadIdx.x & 3;
    int r = (int)(v % ((int64_t)D * H * W));
    const int64_t nb = v - r;
    const int d = r / (H * W); r -= d * H * W;
    const int h = r / W;
    const int wv = r - h * W;
    float acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = bias ? bias[q * 16 + j] : 0.f;
    for (int a = 0; a < 3; ++a) {
        const int qd = clampi(d + a - 1, D - 1);
        for (int b = 0; b < 3; ++b) {
            const int qh = clampi(h + b - 1, H - 1);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int qw = clampi(wv + c - 1, W - 1);
                const T* xp = x + (nb + ((int64_t)qd * H + qh) * W + qw) * 3;
                const float x0 = fdn_ld1(xp), x1 = fdn_ld1(xp + 1), x2 = fdn_ld1(xp + 2);
                const float* wp = ws + ((a * 3 + b) * 3 + c) * 192 + q * 16;
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j] += x0 * wp[j] + x1 * wp[64 + j] + x2 * wp[128 + j];
            }
        }
    }
    T* yp = y + v * 64 + q * 16;
#pragma unroll
    for (int j = 0; j < 16; j += 4) {
        f32x4 o = {fdn_act(acc[j], act, alpha), fdn_act(acc[j + 1], act, alpha), fdn_act(acc[j + 2], act, alpha),
                   fdn_act(acc[j + 3], act, alpha)};
        fdn_st4(yp + j, o);
    }
}

// -------------------------------------------------------------------------------------------------
// forward 64 -> 1, k=3.  16 lanes per voxel (4 channels each), 27 coalesced row reads, 16-lane reduce.
// -------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv_cout1_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ bias, float* __restrict__ y,
                                                              int N, int D, int H, int W, int ldy, int y_coff, int act,
                                                              float alpha) {
    __shared__ __attribute__((aligned(16))) float ws[27 * 64];
    for (int i = threadIdx.x; i < 27 * 64; i += 256) ws[i] = w[i];
    __syncthreads();
    const int64_t nvox = (int64_t)N * D * H * W;
    const int c4 = threadIdx.x & 15;
    for (int64_t v = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4); v < nvox; v += (int64_t)gridDim.x * 16) {
        int r = (int)(v % ((int64_t)D * H * W));
        const int64_t nb = v - r;
        const int d = r / (H * W); r -= d * H * W;
        const int h = r / W;
        const int wv = r - h * W;
        float s = 0.f;
        for (int a = 0; a < 3; ++a) {
            const int qd = clampi(d + a - 1, D - 1);
            for (int b = 0; b < 3; ++b) {
                const int qh = clampi(h + b - 1, H - 1);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int qw = clampi(wv + c - 1, W - 1);
                    const f32x4 xv = fdn_ld4(x + (nb + ((int64_t)qd * H + qh) * W + qw) * 64 + c4 * 4);
                    const f32x4 wt = *(const f32x4*)(ws + ((a * 3 + b) * 3 + c) * 64 + c4 * 4);
                    s += xv.x * wt.x + xv.y * wt.y + xv.z * wt.z + xv.w * wt.w;
                }
            }
        }
        s += __shfl_xor(s, 8, 16);
        s += __shfl_xor(s, 4, 16);
        s += __shfl_xor(s, 2, 16);
        s += __shfl_xor(s, 1, 16);
        if (c4 == 0) y[v * ldy + y_coff] = fdn_act(s + (bias ? bias[0] : 0.f), act, alpha);
    }
}

// -------------------------------------------------------------------------------------------------
// forward 1x1x1 (64 + 64) -> 64.  Thread = (voxel, 16 couts); weights (128 x 64) in LDS.
// -------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv1x1_fwd_kernel(const T* __restrict__ xa, const T* __restrict__ xb,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           T* __restrict__ y, int64_t nvox, int act, float alpha) {
    __shared__ __attribute__((aligned(16))) float ws[128 * 64];
    for (int i = threadIdx.x; i < 128 * 64; i += 256) ws[i] = w[i];
    __syncthreads();
    const int q = threadIdx.x & 3;
    for (int64_t v = (int64_t)blockIdx.x * 64 + (threadIdx.x >> 2); v < nvox; v += (int64_t)gridDim.x * 64) {
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = bias ? bias[q * 16 + j] : 0.f;
#pragma unroll 1
        for (int src = 0; src < 2; ++src) {
            const T* xp = (src ? xb : xa) + v * 64;
            const float* wsrc = ws + src * 64 * 64 + q * 16;
#pragma unroll 4
            for (int k4 = 0; k4 < 16; ++k4) {
                const f32x4 xv = fdn_ld4(xp + k4 * 4);
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const float xs = xv[kk];
                    const float* wr = wsrc + (k4 * 4 + kk) * 64;
#pragma unroll
                    for (int j = 0; j < 16; ++j) acc[j] += xs * wr[j];
                }
            }
        }
        T* yp = y + v * 64 + q * 16;
#pragma unroll
        for (int j = 0; j < 16; j += 4) {
            f32x4 o = {fdn_act(acc[j], act, alpha), fdn_act(acc[j + 1], act, alpha), fdn_act(acc[j + 2], act, alpha),
                       fdn_act(acc[j + 3], act, alpha)};
            fdn_st4(yp + j, o);
        }
    }
}

// 1x1 backward w.r.t. inputs, fused with the ReLU masks of the two producers.
// thread = (voxel, 16 of the 128 input channels); Wt[co][ci] in LDS.
template <typename T>
__global__ __launch_bounds__(256) void conv1x1_dgrad_kernel(const T* __restrict__ dz, const float* __restrict__ w,
                                                             const T* __restrict__ ya, const T* __restrict__ yb,
                                                             T* __restrict__ dxa, T* __restrict__ dxb, int64_t nvox) {
    __shared__ __attribute__((aligned(16))) float wt[64 * 128];
    for (int i = threadIdx.x; i < 128 * 64; i += 256) {
        const int ci = i >> 6, co = i & 63;
        wt[co * 128 + ci] = w[i];
    }
    __syncthreads();
    const int q = threadIdx.x & 7;        // 16-channel group of the 128 inputs
    for (int64_t v = (int64_t)blockIdx.x * 32 + (threadIdx.x >> 3); v < nvox; v += (int64_t)gridDim.x * 32) {
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        const T* dp = dz + v * 64;
#pragma unroll 4
        for (int k4 = 0; k4 < 16; ++k4) {
            const f32x4 dv = fdn_ld4(dp + k4 * 4);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const float ds = dv[kk];
                const float* wr = wt + (k4 * 4 + kk) * 128 + q * 16;
#pragma unroll
                for (int j = 0; j < 16; ++j) acc[j] += ds * wr[j];
            }
        }
        const int half = q >> 2;
        const int cofs = (q & 3) * 16;
        const T* yp = (half ? yb : ya) + v * 64 + cofs;
        T* op = (half ? dxb : dxa) + v * 64 + cofs;
#pragma unroll
        for (int j = 0; j < 16; j += 4) {
            const f32x4 yv = fdn_ld4(yp + j);
            f32x4 o = {yv.x > 0.f ? acc[j] : 0.f, yv.y > 0.f ? acc[j + 1] : 0.f, yv.z > 0.f ? acc[j + 2] : 0.f,
                       yv.w > 0.f ? acc[j + 3] : 0.f};
            fdn_st4(op + j, o);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// dgrad of 64 -> 1 (k=3) onto the padded grid: dxpad[P][ci] = sum_u w[-u][ci] * dz[P+u]  (zero outside).
// One wave per padded position at a time, lanes = ci; the 27 dz scalars are wave-uniform.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_cout1_dgrad_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                                                float* __restrict__ dxpad, int N, int D, int H, int W,
                                                                int lddz, int dz_coff) {
    const int lane = threadIdx.x & 63;
    float wr[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) wr[t] = w[t * 64 + lane];
    const int PD = D + 2, PH = H + 2, PW = W + 2;
    const int64_t npos = (int64_t)N * PD * PH * PW;
    const int64_t wave_id = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    for (int64_t pos = wave_id; pos < npos; pos += nwaves) {
        int r = (int)(pos % ((int64_t)PD * PH * PW));
        const int n = (int)(pos / ((int64_t)PD * PH * PW));
        const int pd = r / (PH * PW); r -= pd * PH * PW;
        const int ph = r / PW;
        const int pw = r - ph * PW;
        float s = 0.f;
        // padded index p <-> position P = p-1; contributing output voxel o = P - (t-1) = p - t, t = tap index
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int od = pd - a;
            if ((unsigned)od >= (unsigned)D) continue;
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int oh = ph - b;
                if ((unsigned)oh >= (unsigned)H) continue;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int ow = pw - c;
                    if ((unsigned)ow >= (unsigned)W) continue;
                    const float g = dz[(((int64_t)n * D + od) * H + oh) * W * lddz + (int64_t)ow * lddz + dz_coff];
                    s += wr[(a * 3 + b) * 3 + c] * g;
                }
            }
        }
        dxpad[pos * 64 + lane] = s;
    }
}

// -------------------------------------------------------------------------------------------------
// weight gradients of the thin layers: per-block partials in workspace, then reduce_partials.
// -------------------------------------------------------------------------------------------------
// 3 -> 64: one wave owns the full (81 x 64) output (lane = cout, 81 accumulators), waves take voxels round-robin.
template <typename T>
__global__ __launch_bounds__(256) void wgrad_cin3_kernel(const T* __restrict__ x, const T* __restrict__ dz,
                                                          float* __restrict__ partial, int N, int D, int H, int W) {
    __shared__ float red[81 * 64];
    const int lane = threadIdx.x & 63;
    // provably wave-uniform: the 81 x values of a voxel are then fetched with scalar loads and enter the FMAs as SGPR operands
    // (as vector loads of a uniform address they were 81 memory instructions per voxel and wave: 169 us per launch at 8x24^3)
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float acc[81];
#pragma unroll
    for (int t = 0; t < 81; ++t) acc[t] = 0.f;
    // every wave owns one contiguous run of voxels: (n,d,h,w) is decoded once and advanced by carries -- the per-voxel 64-bit
    // modulo + two divisions of the first version cost ~400 instructions per voxel (169 us per launch at 8x24^3)
    const int64_t nvox = (int64_t)N * D * H * W;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int64_t per_wave = (nvox + nwaves - 1) / nwaves;
    const int64_t v_begin = ((int64_t)blockIdx.x * 4 + wv) * per_wave;
    const int64_t v_end = v_begin + per_wave < nvox ? v_begin + per_wave : nvox;
    int w0 = 0, h = 0, d = 0;
    int64_t nb = 0;                       // first voxel of the sample
    if (v_begin < v_end) {
        int r = (int)(v_begin % ((int64_t)D * H * W));
        nb = v_begin - r;
        d = r / (H * W); r -= d * H * W;
        h = r / W;
        w0 = r - h * W;
    }
    for (int64_t v = v_begin; v < v_end; ++v) {
        const float g = fdn_ld1(dz + v * 64 + lane);
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const int qd = clampi(d + a - 1, D - 1);
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const int qh = clampi(h + b - 1, H - 1);
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const int qw = clampi(w0 + c - 1, W - 1);
                    const T* xp = x + (nb + ((int64_t)qd * H + qh) * W + qw) * 3;
                    const int t = ((a * 3 + b) * 3 + c) * 3;
                    acc[t] += fdn_ld1(xp) * g;
                    acc[t + 1] += fdn_ld1(xp + 1) * g;
                    acc[t + 2] += fdn_ld1(xp + 2) * g;
                }
            }
        }
        if (++w0 == W) { w0 = 0; if (++h == H) { h = 0; if (++d == D) { d = 0; nb += (int64_t)D * H * W; } } }
    }
    for (int ph = 0; ph < 4; ++ph) {       // waves fold their accumulators into LDS one after another
        if (wv == ph) {
#pragma unroll
            for (int t = 0; t < 81; ++t) {
                if (ph == 0) red[t * 64 + lane] = acc[t];
                else red[t * 64 + lane] += acc[t];
            }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < 81 * 64; i += 256) partial[(size_t)blockIdx.x * (81 * 64) + i] = red[i];
}

// 64 -> 1: iterate over rows of PADDED input positions so each x row is read once.  Thread = (position slot g,
// 4 channels q); the 9 dz rows a padded (d,h) row can touch are staged in LDS (zero-padded), 27 float4 accumulators.
template <typename T>
__global__ __launch_bounds__(256) void wgrad_cout1_kernel(const T* __restrict__ x, const float* __restrict__ dz,
                                                           float* __restrict__ partial, int N, int D, int H, int W,
                                                           int lddz, int dz_coff) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int ZW = W + 4;                       // z[k][j] = dz[od_k][oh_k][j-2], zero outside
    float* z = sm;                              // 9 * ZW
    float* red = sm + ((9 * ZW + 3) & ~3);      // 4 waves * 27 * 64
    const int q = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int PD = D + 2, PH = H + 2, PW = W + 2;
    f32x4 acc[27];
#pragma unroll
    for (int t = 0; t < 27; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nrows = N * PD * PH;
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int n = row / (PD * PH);
        const int r2 = row - n * (PD * PH);
        const int pd = r2 / PH, ph = r2 - (r2 / PH) * PH;
        __syncthreads();
        for (int i = threadIdx.x; i < 9 * ZW; i += 256) {
            const int k = i / ZW, j = i - k * ZW;
            const int od = pd - k / 3, oh = ph - k % 3, ow = j - 2;      // padded p is tap t's input of output o = p - t
            float v = 0.f;
            if ((unsigned)od < (unsigned)D && (unsigned)oh < (unsigned)H && (unsigned)ow < (unsigned)W)
                v = dz[((((int64_t)n * D + od) * H + oh) * W + ow) * lddz + dz_coff];
            z[i] = v;
        }
        __syncthreads();
        const int qd = clampi(pd - 1, D - 1), qh = clampi(ph - 1, H - 1);
        const T* xrow = x + (((int64_t)n * D + qd) * H + qh) * W * 64 + q * 4;
        for (int pw = g; pw < PW; pw += 16) {
            const f32x4 xv = fdn_ld4(xrow + (int64_t)clampi(pw - 1, W - 1) * 64);
#pragma unroll
            for (int k = 0; k < 9; ++k)
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[k * 3 + c] += xv * z[k * ZW + pw - c + 2];
        }
    }
    // fold the 4 position slots of a wave (lane bits 4,5), then the 4 waves through LDS
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[t][e];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            acc[t][e] = v;
        }
    }
    __syncthreads();
    if ((threadIdx.x & 63) < 16) {
#pragma unroll
        for (int t = 0; t < 27; ++t) *(f32x4*)(red + (wv * 27 + t) * 64 + q * 4) = acc[t];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 27 * 64; i += 256)
        partial[(size_t)blockIdx.x * (27 * 64) + i] = (red[i] + red[27 * 64 + i]) + (red[2 * 27 * 64 + i] + red[3 * 27 * 64 + i]);
}

// dgrad of 64 -> 1 (k=3) FUSED with the halo fold and the producer's activation gradient:
//   dz_prev[i][c] = act'(y[i][c]) * sum over the <=27 (output voxel o, tap t) pairs with clamp(o + t - 1) == i of w[t][c]*dz[o].
// Per dimension a voxel i always has exactly the pairs {(o,a): clamp(o+a-1)=i}, at most 3 of them.
__device__ __forceinline__ int pairs_of(int i, int n, int* o, int* a) {
    int cnt = 0;
#pragma unroll
    for (int t = 0; t < 3; ++t) {            // candidate padded positions: i+1 always, 0 if i==0, n+1 if i==n-1
#pragma unroll
        for (int which = 0; which < 3; ++which) {
            const int p = which == 0 ? i + 1 : (which == 1 ? 0 : n + 1);
            if ((which == 1 && i != 0) || (which == 2 && i != n - 1)) continue;
            const int oo = p - t;
            if ((unsigned)oo < (unsigned)n) { o[cnt] = oo; a[cnt] = t; ++cnt; }
        }
    }
    return cnt;   // <= 3 by construction
}

template <typename T>
__global__ __launch_bounds__(256) void conv_cout1_dgrad_folded_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                                                       const T* __restrict__ yprev, int act, float alpha,
                                                                       T* __restrict__ out, float* __restrict__ bpart,
                                                                       int N, int D, int H, int W, int lddz, int dz_coff) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* wl = sm;                 // 27 * 64 weights
    float* z = sm + 27 * 64;        // 9 rows * W
    int* meta = (int*)(z + 9 * W);  // [0]=number of (d,h) pairs, then tap base (a*3+b)*3 per row
    for (int i = threadIdx.x; i < 27 * 64; i += 256) wl[i] = w[i];
    const int q = threadIdx.x & 15, g = threadIdx.x >> 4;
    const int nrows = N * D * H;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    for (int row = blockIdx.x; row < nrows; row += gridDim.x) {
        const int n = row / (D * H);
        const int r2 = row - n * (D * H);
        const int d = r2 / H, h = r2 - (r2 / H) * H;
        int od[4], ta[4], oh[4], tb[4];
        const int nd = pairs_of(d, D, od, ta), nh = pairs_of(h, H, oh, tb);
        __syncthreads();
        if (threadIdx.x == 0) meta[0] = nd * nh;
        if (threadIdx.x < nd * nh) meta[1 + threadIdx.x] = (ta[threadIdx.x / nh] * 3 + tb[threadIdx.x % nh]) * 3;
        for (int i = threadIdx.x; i < nd * nh * W; i += 256) {
            const int k = i / W, ow = i - k * W;
            z[i] = dz[((((int64_t)n * D + od[k / nh]) * H + oh[k % nh]) * W + ow) * lddz + dz_coff];
        }
        __syncthreads();
        const int nk = meta[0];
        for (int wv = g; wv < W; wv += 16) {
            int ow[4], tc[4];
            const int nw = pairs_of(wv, W, ow, tc);
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int k = 0; k < nk; ++k) {
                const int tbase = meta[1 + k];
                for (int c = 0; c < nw; ++c)
                    acc += *(const f32x4*)(wl + (tbase + tc[c]) * 64 + q * 4) * z[k * W + ow[c]];
            }
            const int64_t o = ((((int64_t)n * D + d) * H + h) * W + wv) * 64 + q * 4;
            if (yprev) {
                const f32x4 y = fdn_ld4(yprev + o);
                acc.x *= fdn_act_grad(y.x, act, alpha); acc.y *= fdn_act_grad(y.y, act, alpha);
                acc.z *= fdn_act_grad(y.z, act, alpha); acc.w *= fdn_act_grad(y.w, act, alpha);
            }
            fdn_st4(out + o, acc);
            bsum += acc;
        }
    }
    if (bpart) {
        // BiasAddGrad of the producing layer = per-channel sum of dz_prev: fold the 4 position slots of a wave
        // (lane bits 4,5), then the 4 waves through LDS; one 64-float partial per block
        __syncthreads();
        float* red = sm;                        // reuse the weight area
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = bsum[e];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            bsum[e] = v;
        }
        if ((threadIdx.x & 63) < 16) *(f32x4*)(red + (threadIdx.x >> 6) * 64 + q * 4) = bsum;
        __syncthreads();
        if (threadIdx.x < 64)
            bpart[(size_t)blockIdx.x * 64 + threadIdx.x] = (red[threadIdx.x] + red[64 + threadIdx.x]) +
                                                            (red[128 + threadIdx.x] + red[192 + threadIdx.x]);
    }
}

// 1x1 (64+64) -> 64: thread owns ci = tid>>1 and 32 couts; voxel chunks staged through LDS.
template <typename T>
__global__ __launch_bounds__(256) void wgrad_1x1_kernel(const T* __restrict__ xa, const T* __restrict__ xb,
                                                         const T* __restrict__ dz, float* __restrict__ partial,
                                                         int64_t nvox) {
    constexpr int CH = 32;   // voxels per chunk
    __shared__ __attribute__((aligned(16))) float xs[CH * 128];
    __shared__ __attribute__((aligned(16))) float zs[CH * 64];
    const int ci = threadIdx.x >> 1;
    const int ch = threadIdx.x & 1;
    float acc[32];
#pragma unroll
    for (int j = 0; j < 32; ++j) acc[j] = 0.f;
    for (int64_t v0 = (int64_t)blockIdx.x * CH; v0 < nvox; v0 += (int64_t)gridDim.x * CH) {
        __syncthreads();
        for (int i = threadIdx.x; i < CH * 32; i += 256) {      // float4 units: 16 from xa + 16 from xb per voxel
            const int vv = i >> 5, c4 = i & 31;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (v0 + vv < nvox) val = fdn_ld4((c4 < 16 ? xa : xb) + (v0 + vv) * 64 + (c4 & 15) * 4);
            *(f32x4*)(xs + vv * 128 + c4 * 4) = val;
        }
        for (int i = threadIdx.x; i < CH * 16; i += 256) {
            const int vv = i >> 4, c4 = i & 15;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (v0 + vv < nvox) val = fdn_ld4(dz + (v0 + vv) * 64 + c4 * 4);
            *(f32x4*)(zs + vv * 64 + c4 * 4) = val;
        }
        __syncthreads();
#pragma unroll 2
        for (int vv = 0; vv < CH; ++vv) {
            const float xv = xs[vv * 128 + ci];
            const float* zr = zs + vv * 64 + ch * 32;
#pragma unroll
            for (int j = 0; j < 32; j += 4) {
                const f32x4 g = *(const f32x4*)(zr + j);
                acc[j] += xv * g.x; acc[j + 1] += xv * g.y; acc[j + 2] += xv * g.z; acc[j + 3] += xv * g.w;
            }
        }
    }
    float* out = partial + (size_t)blockIdx.x * (128 * 64) + ci * 64 + ch * 32;
#pragma unroll
    for (int j = 0; j < 32; ++j) out[j] = acc[j];
}

// bias gradient: db[c] = sum_v dz[v*ld + coff + c]
template <typename T>
__global__ __launch_bounds__(256) void bias_grad_kernel(const T* __restrict__ dz, float* __restrict__ partial,
                                                         int64_t nvox, int C, int lddz, int dz_coff) {
    __shared__ float red[256];
    float s = 0.f;
    if (C == 64 && lddz == 64 && dz_coff == 0) {
        // dense rows (every 64-channel caller): 16-B vectors, four rows per thread in flight -- the scalar form below kept ONE 4-B load per
        // thread in flight (15 us for the 28 MB of an (8,24^3) tensor)
        constexpr int E = FdnVec<T>::E, CV = 64 / E, RPB = 256 / CV;          // vectors per row, rows per block and sweep
        __shared__ float redv[256 * E];
        const int cv = threadIdx.x % CV, rs = threadIdx.x / CV;
        float acc[E];
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = 0.f;
        const int64_t stride = (int64_t)gridDim.x * RPB;
        int64_t v = (int64_t)blockIdx.x * RPB + rs;
        for (; v + 3 * stride < nvox; v += 4 * stride) {
            float t[4][E];
#pragma unroll
            for (int u = 0; u < 4; ++u) FdnVec<T>::ld(dz + (v + u * stride) * 64 + cv * E, t[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < E; ++e) acc[e] += t[u][e];
        }
        for (; v < nvox; v += stride) {
            float t[E];
            FdnVec<T>::ld(dz + v * 64 + cv * E, t);
#pragma unroll
            for (int e = 0; e < E; ++e) acc[e] += t[e];
        }
#pragma unroll
        for (int e = 0; e < E; ++e) redv[(rs * CV + cv) * E + e] = acc[e];
        __syncthreads();
        if (threadIdx.x < 64) {                                           // channel c: row slots summed in index order
            float r = 0.f;
            for (int k = 0; k < RPB; ++k) r += redv[k * 64 + threadIdx.x];
            partial[(size_t)blockIdx.x * 64 + threadIdx.x] = r;
        }
    } else if (C == 64) {
        const int c = threadIdx.x & 63;
        for (int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); v < nvox; v += (int64_t)gridDim.x * 4)
            s += fdn_ld1(dz + v * lddz + dz_coff + c);
        red[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x < 64)
            partial[(size_t)blockIdx.x * 64 + c] = (red[c] + red[64 + c]) + (red[128 + c] + red[192 + c]);
    } else {   // C == 1
        for (int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x; v < nvox; v += (int64_t)gridDim.x * 256)
            s += fdn_ld1(dz + v * lddz + dz_coff);
        red[threadIdx.x] = s;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
    }
}

// out[e] = sum_p partial[p][e].  Block = 32 elements x 8 part-lanes: the nparts-long chain is split 8 ways (and unrolled
// 4x) so the reduction is not one long dependent latency chain per element; 128-B coalesced reads per part.
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, float* __restrict__ out,
                                                               int nparts, int nelem) {
    __shared__ float red[8][32];
    const int ei = threadIdx.x & 31, pl = threadIdx.x >> 5;
    const int e = blockIdx.x * 32 + ei;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < nelem) {
        int p = pl;
        for (; p + 24 < nparts; p += 32) {
            s0 += partial[(size_t)p * nelem + e];
            s1 += partial[(size_t)(p + 8) * nelem + e];
            s2 += partial[(size_t)(p + 16) * nelem + e];
            s3 += partial[(size_t)(p + 24) * nelem + e];
        }
        for (; p < nparts; p += 8) s0 += partial[(size_t)p * nelem + e];
    }
    red[pl][ei] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (pl == 0 && e < nelem)
        out[e] = ((red[0][ei] + red[1][ei]) + (red[2][ei] + red[3][ei])) + ((red[4][ei] + red[5][ei]) + (red[6][ei] + red[7][ei]));
}

constexpr int kSmallBlocks = 512;   // partial-sum blocks of the thin-layer wgrad / bias kernels
constexpr int kSmallRows = 768;     // partial rows the workspace holds: the bf16 head wgrad runs three workgroups per CU (heads_mfma.hip)

int reduce_partials(const float* partial, float* out, int nparts, int nelem, hipStream_t s) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3((nelem + 31) / 32), dim3(256), 0, s, partial, out, nparts, nelem);
    FDN_CHECK_LAUNCH("reduce_partials_kernel");
    return FDN_OK;
}

int nblocks_for(int64_t items, int per_block, int cap) {
    int64_t b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (int)b;
}

}  // namespace

size_t fdn_small_wgrad_workspace_bytes(int Cin, int Cout, int K) {
    size_t elems = (size_t)K * K * K * Cin * Cout;
    if (elems < 64) elems = 64;
    return (size_t)kSmallRows * elems * sizeof(float);
}

// Launchers, templated on the activation storage type T (float, or uint16_t = bf16 bits); parameters, parameter
// gradients, the 64->1 heads' output (the prediction) and its gradient are always fp32.
// cin3_mfma.hip: the im2col-GEMM formulation of the two 3 -> 64 kernels (default); the VALU kernels of this file stay selectable
// for A/B runs in the test build (fdn_debug_set_cin3_mfma(0)) and serve odd W in the weight gradient
template <typename T> int fdn_conv_cin3_fwd_mfma_launch(const T* x, const float* w, const float* bias, T* y, int N, int D, int H,
                                                        int W, int act, float alpha, hipStream_t s);
template <typename T> int fdn_wgrad_cin3_mfma_launch(const T* x, const T* dz, float* partial, int nblocks, int N, int D, int H,
                                                     int W, hipStream_t s);
FDN_HOOK_VAR(int, fdn_cin3_use_mfma, 1);
// conv1x1_mfma.hip: the 1x1x1 (64+64) -> 64 layer and its gradients as MFMA GEMMs (default; the VALU kernels stay selectable in the
// test build: fdn_debug_set_conv1x1_mfma(0))
template <typename T> int fdn_conv1x1_fwd_mfma_launch(const T* xa, const T* xb, const float* w, const float* bias, T* y, int64_t nvox,
                                                      int act, float alpha, hipStream_t s);
template <typename T> int fdn_conv1x1_dgrad_mfma_launch(const T* dz, const float* w, const T* ya, const T* yb, T* dxa, T* dxb,
                                                        int64_t nvox, hipStream_t s);
template <typename T> int fdn_wgrad_1x1_mfma_launch(const T* xa, const T* xb, const T* dz, float* partial, int nblocks, int64_t nvox,
                                                    hipStream_t s);
FDN_HOOK_VAR(int, fdn_conv1x1_use_mfma, 1);
#ifdef FDN_TEST_HOOKS
extern "C" int fdn_debug_set_cin3_mfma(int on) { fdn_cin3_use_mfma = on; return FDN_OK; }
extern "C" int fdn_debug_set_conv1x1_mfma(int on) { fdn_conv1x1_use_mfma = on; return FDN_OK; }
#endif

template <typename T>
int fdn_conv_cin3_fwd_launch(const T* x, const float* w, const float* bias, T* y, int N, int D, int H, int W, int act,
                             float alpha, hipStream_t s) {
#ifdef FDN_TEST_HOOKS                                   // (the VALU kernel is compiled into the test build only: A/B reference)
    if (!fdn_cin3_use_mfma) {
        const int64_t nvox = (int64_t)N * D * H * W;
        hipLaunchKernelGGL(conv_cin3_fwd_kernel<T>, dim3((unsigned)((nvox + 63) / 64)), dim3(256), 0, s, x, w, bias, y, N, D, H,
                           W, act, alpha);
        FDN_CHECK_LAUNCH("conv_cin3_fwd_kernel");
        return FDN_OK;
    }
#endif
    return fdn_conv_cin3_fwd_mfma_launch<T>(x, w, bias, y, N, D, H, W, act, alpha, s);
}

// heads_mfma.hip: the MFMA formulation of the three 64 -> 1 head kernels (default); the VALU kernels of this file stay
// selectable for A/B runs (fdn_debug_set_heads_mfma(0))
template <typename T> int fdn_head_fwd_launch(const T* x, const float* w, const float* bias, float* y, int N, int D, int H,
                                              int W, int ldy, int y_coff, int act, float alpha, hipStream_t s, int xcd_walk);
template <typename T> int fdn_head_dgrad_launch(const float* dz, const float* w, const T* y_prev, int act, float alpha, T* dz_prev,
                                                float* bpart, int N, int D, int H, int W, int lddz, int dz_coff, hipStream_t s,
                                                const uint16_t* ymask);
int fdn_head_dgrad_blocks(int N, int D, int H, int W);
template <typename T> int fdn_head_wgrad_launch(const T* x, const float* dz, float* partial, int N, int D, int H, int W, int lddz,
                                                int dz_coff, hipStream_t s);
int fdn_head_wgrad_blocks(int N, int D, int H, int W, int elem_bytes);
FDN_HOOK_VAR(int, fdn_heads_use_mfma, 1);
#ifdef FDN_TEST_HOOKS
extern "C" int fdn_debug_set_heads_mfma(int on) { fdn_heads_use_mfma = on; return FDN_OK; }
#endif

template <typename T>
int fdn_conv_cout1_fwd_launch(const T* x, const float* w, const float* bias, float* y, int N, int D, int H, int W, int ldy,
                              int y_coff, int act, float alpha, hipStream_t s) {
#ifdef FDN_TEST_HOOKS
    if (!fdn_heads_use_mfma) {
        const int64_t nvox = (int64_t)N * D * H * W;
        hipLaunchKernelGGL(conv_cout1_fwd_kernel<T>, dim3(nblocks_for(nvox, 16, 8192)), dim3(256), 0, s, x, w, bias, y, N, D, H,
                           W, ldy, y_coff, act, alpha);
        FDN_CHECK_LAUNCH("conv_cout1_fwd_kernel");
        return FDN_OK;
    }
#endif
    return fdn_head_fwd_launch<T>(x, w, bias, y, N, D, H, W, ldy, y_coff, act, alpha, s, !(fdn_heads_use_mfma & 2));
}

template <typename T>
int fdn_conv1x1_fwd_launch(const T* xa, const T* xb, const float* w, const float* bias, T* y, int64_t nvox, int act,
                           float alpha, hipStream_t s) {
#ifdef FDN_TEST_HOOKS
    if (!fdn_conv1x1_use_mfma) {
        hipLaunchKernelGGL(conv1x1_fwd_kernel<T>, dim3(nblocks_for(nvox, 64, 2048)), dim3(256), 0, s, xa, xb, w, bias, y, nvox,
                           act, alpha);
        FDN_CHECK_LAUNCH("conv1x1_fwd_kernel");
        return FDN_OK;
    }
#endif
    return fdn_conv1x1_fwd_mfma_launch<T>(xa, xb, w, bias, y, nvox, act, alpha, s);
}

template <typename T>
int fdn_conv1x1_dgrad_launch(const T* dz, const float* w, const T* ya, const T* yb, T* dxa, T* dxb, int64_t nvox,
                             hipStream_t s) {
#ifdef FDN_TEST_HOOKS
    if (!fdn_conv1x1_use_mfma) {
        hipLaunchKernelGGL(conv1x1_dgrad_kernel<T>, dim3(nblocks_for(nvox, 32, 2048)), dim3(256), 0, s, dz, w, ya, yb, dxa, dxb,
                           nvox);
        FDN_CHECK_LAUNCH("conv1x1_dgrad_kernel");
        return FDN_OK;
    }
#endif
    return fdn_conv1x1_dgrad_mfma_launch<T>(dz, w, ya, yb, dxa, dxb, nvox, s);
}

int fdn_conv_cout1_dgrad_launch(const float* dz, const float* w, float* dxpad, int N, int D, int H, int W, int lddz,
                                int dz_coff, hipStream_t s) {
    const int64_t npos = (int64_t)N * (D + 2) * (H + 2) * (W + 2);
    hipLaunchKernelGGL(conv_cout1_dgrad_kernel, dim3(nblocks_for(npos, 4 * 16, 4096)), dim3(256), 0, s, dz, w, dxpad, N,
                       D, H, W, lddz, dz_coff);
    FDN_CHECK_LAUNCH("conv_cout1_dgrad_kernel");
    return FDN_OK;
}

template <typename T>
int fdn_wgrad_cin3_launch(const T* x, const T* dz, float* dw, void* ws, size_t, int N, int D, int H, int W, hipStream_t s) {
    const int64_t nvox = (int64_t)N * D * H * W;
    const int nb = nblocks_for(nvox, 4 * 32, kSmallBlocks);
    if (fdn_cin3_use_mfma && (W & 1) == 0) {
        if (int rc = fdn_wgrad_cin3_mfma_launch<T>(x, dz, (float*)ws, nb, N, D, H, W, s)) return rc;
        return reduce_partials((const float*)ws, dw, nb, 81 * 64, s);
    }
    hipLaunchKernelGGL(wgrad_cin3_kernel<T>, dim3(nb), dim3(256), 0, s, x, dz, (float*)ws, N, D, H, W);
    FDN_CHECK_LAUNCH("wgrad_cin3_kernel");
    return reduce_partials((const float*)ws, dw, nb, 81 * 64, s);
}

template <typename T>
int fdn_wgrad_cout1_launch(const T* x, const float* dz, float* dw, void* ws, size_t, int N, int D, int H, int W, int lddz,
                           int dz_coff, hipStream_t s) {
#ifdef FDN_TEST_HOOKS
    if (!fdn_heads_use_mfma) {
        const int nrows = N * (D + 2) * (H + 2);
        const int nb = nrows < kSmallBlocks ? nrows : kSmallBlocks;
        const size_t lds = (size_t)(((9 * (W + 4) + 3) & ~3) + 4 * 27 * 64) * sizeof(float);
        hipLaunchKernelGGL(wgrad_cout1_kernel<T>, dim3(nb), dim3(256), lds, s, x, dz, (float*)ws, N, D, H, W, lddz, dz_coff);
        FDN_CHECK_LAUNCH("wgrad_cout1_kernel");
        return reduce_partials((const float*)ws, dw, nb, 27 * 64, s);
    }
#endif
    const int nbm = fdn_head_wgrad_blocks(N, D, H, W, (int)sizeof(T));          // <= kSmallBlocks partial rows of 27*64
    const int rc = fdn_head_wgrad_launch<T>(x, dz, (float*)ws, N, D, H, W, lddz, dz_coff, s);
    if (rc != FDN_OK) return rc;
    return reduce_partials((const float*)ws, dw, nbm, 27 * 64, s);
}

template <typename T>
int fdn_conv_cout1_dgrad_folded_launch(const float* dz, const float* w, const T* y_prev, int act, float alpha, T* dz_prev,
                                       float* dbias_prev, void* workspace, size_t workspace_bytes, int N, int D, int H, int W,
                                       int lddz, int dz_coff, hipStream_t s, const uint16_t* ymask) {
    FDN_REQUIRE(dz && w && dz_prev && N > 0 && D > 0 && H > 0 && W > 0, "fdn_conv_cout1_dgrad_folded: bad argument");
    FDN_REQUIRE(!(ymask && sizeof(T) == 4) || (W % 4 == 0 && fdn_heads_use_mfma), "fdn_conv_cout1_dgrad_folded: the fp32 sign mask (planar words) needs W %% 4 == 0");
#ifdef FDN_TEST_HOOKS
    if (!fdn_heads_use_mfma) {
        FDN_REQUIRE(W <= 4096, "fdn_conv_cout1_dgrad_folded: W too large for the row stage");
        const int nrows = N * D * H;
        const int nb = nrows < 2048 ? nrows : 2048;
        if (dbias_prev && (!workspace || workspace_bytes < (size_t)nb * 64 * sizeof(float))) {
            fdn_set_error("fdn_conv_cout1_dgrad_folded: workspace %zu < %zu bytes", workspace_bytes, (size_t)nb * 64 * sizeof(float));
            return FDN_ERR_WORKSPACE;
        }
        const size_t lds = (size_t)(27 * 64 + 9 * W + 16) * sizeof(float);
        hipLaunchKernelGGL(conv_cout1_dgrad_folded_kernel<T>, dim3(nb), dim3(256), lds, s, dz, w, y_prev, act, alpha, dz_prev,
                           dbias_prev ? (float*)workspace : nullptr, N, D, H, W, lddz, dz_coff);
        FDN_CHECK_LAUNCH("conv_cout1_dgrad_folded_kernel");
        if (dbias_prev) return reduce_partials((const float*)workspace, dbias_prev, nb, 64, s);
        return FDN_OK;
    }
#endif
    const int nb = fdn_head_dgrad_blocks(N, D, H, W);
    if (dbias_prev && (!workspace || workspace_bytes < (size_t)nb * 64 * sizeof(float))) {
        fdn_set_error("fdn_conv_cout1_dgrad_folded: workspace %zu < %zu bytes", workspace_bytes, (size_t)nb * 64 * sizeof(float));
        return FDN_ERR_WORKSPACE;
    }
    const int rc = fdn_head_dgrad_launch<T>(dz, w, y_prev, act, alpha, dz_prev, dbias_prev ? (float*)workspace : nullptr, N, D,
                                            H, W, lddz, dz_coff, s, ymask);
    if (rc != FDN_OK) return rc;
    if (dbias_prev) return reduce_partials((const float*)workspace, dbias_prev, nb, 64, s);
    return FDN_OK;
}

template <typename T>
int fdn_wgrad_1x1_launch(const T* xa, const T* xb, const T* dz, float* dw, void* ws, size_t, int64_t nvox, hipStream_t s) {
#ifdef FDN_TEST_HOOKS
    if (!fdn_conv1x1_use_mfma) {
        const int nb = nblocks_for(nvox, 32 * 4, kSmallBlocks);
        hipLaunchKernelGGL(wgrad_1x1_kernel<T>, dim3(nb), dim3(256), 0, s, xa, xb, dz, (float*)ws, nvox);
        FDN_CHECK_LAUNCH("wgrad_1x1_kernel");
        return reduce_partials((const float*)ws, dw, nb, 128 * 64, s);
    }
#endif
    // one workgroup per CU (128 KB of LDS for the cross-wave sum); >= 64 voxel pairs per wave
    int nbm = (int)((nvox / 2 + 255) / 256);
    nbm = nbm < 1 ? 1 : (nbm > 256 ? 256 : nbm);
    if (int rc = fdn_wgrad_1x1_mfma_launch<T>(xa, xb, dz, (float*)ws, nbm, nvox, s)) return rc;
    return reduce_partials((const float*)ws, dw, nbm, 128 * 64, s);
}

template <typename T>
int fdn_bias_grad_launch(const T* dz, float* db, void* ws, size_t ws_bytes, int64_t nvox, int C, int lddz, int dz_coff,
                         hipStream_t s) {
    if (C != 64 && C != 1) { fdn_set_error("bias_grad: unsupported C=%d", C); return FDN_ERR_UNSUPPORTED; }
    const int nb = nblocks_for(nvox, C == 64 ? 4 * 64 : 256 * 16, C == 64 ? 256 : kSmallBlocks);
    if (ws_bytes < (size_t)nb * C * sizeof(float)) { fdn_set_error("bias_grad: workspace too small"); return FDN_ERR_WORKSPACE; }
    hipLaunchKernelGGL(bias_grad_kernel<T>, dim3(nb), dim3(256), 0, s, dz, (float*)ws, nvox, C, lddz, dz_coff);
    FDN_CHECK_LAUNCH("bias_grad_kernel");
    return reduce_partials((const float*)ws, db, nb, C, s);
}

// explicit instantiations used by api.hip
#define FDN_INSTANTIATE_SMALL(T)                                                                                               \
    template int fdn_conv_cin3_fwd_launch<T>(const T*, const float*, const float*, T*, int, int, int, int, int, float,         \
                                             hipStream_t);                                                                     \
    template int fdn_conv_cout1_fwd_launch<T>(const T*, const float*, const float*, float*, int, int, int, int, int, int, int, \
                                              float, hipStream_t);                                                             \
    template int fdn_conv1x1_fwd_launch<T>(const T*, const T*, const float*, const float*, T*, int64_t, int, float,            \
                                           hipStream_t);                                                                       \
    template int fdn_conv1x1_dgrad_launch<T>(const T*, const float*, const T*, const T*, T*, T*, int64_t, hipStream_t);        \
    template int fdn_wgrad_cin3_launch<T>(const T*, const T*, float*, void*, size_t, int, int, int, int, hipStream_t);         \
    template int fdn_wgrad_cout1_launch<T>(const T*, const float*, float*, void*, size_t, int, int, int, int, int, int,        \
                                           hipStream_t);                                                                       \
    template int fdn_conv_cout1_dgrad_folded_launch<T>(const float*, const float*, const T*, int, float, T*, float*, void*,    \
                                                       size_t, int, int, int, int, int, int, hipStream_t, const uint16_t*);   \
    template int fdn_wgrad_1x1_launch<T>(const T*, const T*, const T*, float*, void*, size_t, int64_t, hipStream_t);           \
    template int fdn_bias_grad_launch<T>(const T*, float*, void*, size_t, int64_t, int, int, int, hipStream_t);
FDN_INSTANTIATE_SMALL(float)
FDN_INSTANTIATE_SMALL(uint16_t)

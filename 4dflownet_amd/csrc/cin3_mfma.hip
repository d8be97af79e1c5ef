// The two 3 -> 64 input convolutions (SR4DFlowNet.py:17,20) as im2col GEMMs on v_mfma_f32_32x32x2_f32:
//   forward : y[v][co]  = act(b[co] + sum_k P[v][k] Wk[k][co]),   P[v][k = 3 tap + c] = x[clamp(v + tap - 1)][c]   (81 columns)
//   wgrad   : dW[k][co] = sum_v P[v][k] dz[v][co]
// 1.15 GFLOP each at (8,24^3): 7 us of matrix time, against 62 / 105 us of the VALU kernels they replace (small_convs.hip:
// one thread per (voxel, 16 cout) resp. scalar loads + 81 FMAs per voxel and wave).  x is 1.3 MB and stays in L1/L2; the patch
// columns are gathered straight from it (edge clamp == SYMMETRIC p=1), so nothing is staged in LDS.
// Storage type T = float or bf16 bits (uint16_t); arithmetic fp32 in both.
#include "fdn_common.h"

namespace {

__device__ __forceinline__ int clampi3(int v, int hi) { return min(max(v, 0), hi); }

// Persistent waves over blocks of 32 voxels; lane (li, kh): A = P[voxel li][k = 2 s + kh], B = Wk[k][32 nt + li] (82 VGPRs,
// loaded once), 41 K-steps x 2 cout tiles.  D[voxel][cout]: register r of lane (li, kh) is voxel (r&3) + 8 (r>>2) + 4 kh,
// channel 32 nt + li -> every store instruction writes 128-B runs.
template <typename T>
__global__ __launch_bounds__(256, 2) void conv_cin3_fwd_mfma_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                                    const float* __restrict__ bias, T* __restrict__ y, int N, int D,
                                                                    int H, int W, int act, float alpha) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, kh = lane >> 5;
    float wr[2][41];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int s = 0; s < 41; ++s) wr[nt][s] = (2 * s + kh) < 81 ? w[(2 * s + kh) * 64 + 32 * nt + li] : 0.f;
    const float b0 = bias ? bias[li] : 0.f, b1 = bias ? bias[32 + li] : 0.f;
    const float slope = act == FDN_ACT_RELU ? 0.f : (act == FDN_ACT_LEAKY ? alpha : 1.f);
    const int64_t nvox = (int64_t)N * D * H * W;
    const int64_t nblk = (nvox + 31) / 32;
    const int HW = H * W, DHW = D * HW;
    for (int64_t blk = (int64_t)blockIdx.x * 4 + wave; blk < nblk; blk += (int64_t)gridDim.x * 4) {
        int64_t v = blk * 32 + li;
        v = v < nvox ? v : nvox - 1;
        const int n = (int)(v / DHW);
        int r = (int)(v - (int64_t)n * DHW);
        const int d = r / HW; r -= d * HW;
        const int h = r / W;
        const int wv = r - h * W;
        // element offsets of the 27 neighbour rows (3 channels each)
        int qd[3], qh[3], qw[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            qd[t] = clampi3(d + t - 1, D - 1) * HW;
            qh[t] = clampi3(h + t - 1, H - 1) * W;
            qw[t] = clampi3(wv + t - 1, W - 1);
        }
        const T* xn = x + (int64_t)n * DHW * 3;
        float av[41];
#pragma unroll
        for (int s = 0; s < 41; ++s) {
            // k = 2 s + kh -> (tap, channel): both candidates are compile-time, the lane half selects
            const int k0 = 2 * s, k1 = 2 * s + 1 < 81 ? 2 * s + 1 : 80;
            const int t0 = k0 / 3, c0 = k0 % 3, t1 = k1 / 3, c1 = k1 % 3;
            const int o0 = (qd[t0 / 9] + qh[(t0 / 3) % 3] + qw[t0 % 3]) * 3 + c0;
            const int o1 = (qd[t1 / 9] + qh[(t1 / 3) % 3] + qw[t1 % 3]) * 3 + c1;
            av[s] = fdn_ld1(xn + (kh ? o1 : o0));
        }
        f32x16 acc0, acc1;
#pragma unroll
        for (int q = 0; q < 16; ++q) { acc0[q] = b0; acc1[q] = b1; }
#pragma unroll
        for (int s = 0; s < 41; ++s) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], wr[0][s], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], wr[1][s], acc1, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int64_t vo = blk * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh;
            if (vo < nvox) {
                const float z0 = acc0[q], z1 = acc1[q];
                fdn_st1(y + vo * 64 + li, fmaxf(z0, slope * z0));
                fdn_st1(y + vo * 64 + 32 + li, fmaxf(z1, slope * z1));
            }
        }
    }
}

// wgrad: D[k][co] over K = voxels.  A wave owns a contiguous run of voxel PAIRS along W (W even) and keeps the 3 x 2 accumulator
// tiles (k in 3 tiles of 32, 81 used; co in 2 tiles) for the whole run; lane (li, kh): A = P[voxel pair-element kh][k = 32 mt + li]
// (its tap and channel are lane constants), B = dz[that voxel][32 nt + li].  Partials per workgroup, summed by reduce_partials.
template <typename T>
__global__ __launch_bounds__(256, 2) void wgrad_cin3_mfma_kernel(const T* __restrict__ x, const T* __restrict__ dz,
                                                                 float* __restrict__ partial, int N, int D, int H, int W) {
    __shared__ float red[81 * 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, kh = lane >> 5;
    int ta[3], tb[3], tc[3], ch[3];
    bool live[3];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt) {
        const int k = 32 * mt + li;
        live[mt] = k < 81;
        const int kk = live[mt] ? k : 80;
        const int t = kk / 3;
        ch[mt] = kk - 3 * t;
        ta[mt] = t / 9 - 1; tb[mt] = (t / 3) % 3 - 1; tc[mt] = t % 3 - 1;
    }
    f32x16 acc[3][2];
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[mt][nt][q] = 0.f;
    const int W2 = W / 2;
    const int64_t npair = (int64_t)N * D * H * W2;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int64_t per_wave = (npair + nwaves - 1) / nwaves;
    const int64_t p_begin = ((int64_t)blockIdx.x * 4 + wave) * per_wave;
    const int64_t p_end = p_begin + per_wave < npair ? p_begin + per_wave : npair;
    int64_t row = p_begin / W2;                      // (n, d, h) row index
    int pw = (int)(p_begin - row * W2);              // pair within the row
    for (int64_t pr = p_begin; pr < p_end;) {
        const int h = (int)(row % H);
        const int64_t nd = row / H;
        const int d = (int)(nd % D);
        const int64_t n = nd / D;
        int64_t rb[3];
#pragma unroll
        for (int mt = 0; mt < 3; ++mt)
            rb[mt] = ((n * D + clampi3(d + ta[mt], D - 1)) * H + clampi3(h + tb[mt], H - 1)) * (int64_t)W;
        const int64_t zrow = row * W;
        const int pstop = (int)((p_end - pr) < (W2 - pw) ? pw + (p_end - pr) : W2);
        for (; pw < pstop; ++pw, ++pr) {
            const int wv = 2 * pw + kh;
            float a[3];
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                const float t = fdn_ld1(x + (rb[mt] + clampi3(wv + tc[mt], W - 1)) * 3 + ch[mt]);
                a[mt] = live[mt] ? t : 0.f;
            }
            const float g0 = fdn_ld1(dz + (zrow + wv) * 64 + li), g1 = fdn_ld1(dz + (zrow + wv) * 64 + 32 + li);
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], g0, acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], g1, acc[mt][1], 0, 0, 0);
            }
        }
        if (pw == W2) { pw = 0; ++row; }
    }
    for (int ph = 0; ph < 4; ++ph) {       // waves fold their accumulators into LDS one after another (fixed order)
        if (wave == ph) {
#pragma unroll
            for (int mt = 0; mt < 3; ++mt)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int k = 32 * mt + (q & 3) + 8 * (q >> 2) + 4 * kh;
                        if (k < 81) {
                            float* dst = red + k * 64 + 32 * nt + li;
                            if (ph == 0) *dst = acc[mt][nt][q];
                            else *dst += acc[mt][nt][q];
                        }
                    }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < 81 * 64; i += 256) partial[(size_t)blockIdx.x * (81 * 64) + i] = red[i];
}

}  // namespace

template <typename T>
int fdn_conv_cin3_fwd_mfma_launch(const T* x, const float* w, const float* bias, T* y, int N, int D, int H, int W, int act,
                                  float alpha, hipStream_t s) {
    const int64_t nblk = ((int64_t)N * D * H * W + 31) / 32;
    int64_t grid = (nblk + 3) / 4;
    if (grid > 512) grid = 512;                     // 2 workgroups per CU, each wave walks several blocks with its weights resident
    hipLaunchKernelGGL(conv_cin3_fwd_mfma_kernel<T>, dim3((unsigned)grid), dim3(256), 0, s, x, w, bias, y, N, D, H, W, act, alpha);
    FDN_CHECK_LAUNCH("conv_cin3_fwd_mfma_kernel");
    return FDN_OK;
}

// nblocks = number of partial rows (81*64 floats each) the kernel writes
template <typename T>
int fdn_wgrad_cin3_mfma_launch(const T* x, const T* dz, float* partial, int nblocks, int N, int D, int H, int W, hipStream_t s) {
    hipLaunchKernelGGL(wgrad_cin3_mfma_kernel<T>, dim3((unsigned)nblocks), dim3(256), 0, s, x, dz, partial, N, D, H, W);
    FDN_CHECK_LAUNCH("wgrad_cin3_mfma_kernel");
    return FDN_OK;
}

template int fdn_conv_cin3_fwd_mfma_launch<float>(const float*, const float*, const float*, float*, int, int, int, int, int, float, hipStream_t);
template int fdn_conv_cin3_fwd_mfma_launch<uint16_t>(const uint16_t*, const float*, const float*, uint16_t*, int, int, int, int, int, float, hipStream_t);
template int fdn_wgrad_cin3_mfma_launch<float>(const float*, const float*, float*, int, int, int, int, int, hipStream_t);
template int fdn_wgrad_cin3_mfma_launch<uint16_t>(const uint16_t*, const uint16_t*, float*, int, int, int, int, int, hipStream_t);

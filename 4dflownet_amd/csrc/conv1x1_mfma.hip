// The 1x1x1 (64 + 64) -> 64 convolution that fuses the phase / pc branches (SR4DFlowNet.py:23-24: concat + Conv3D(k=1) + ReLU) and
// its two gradients as plain GEMMs on v_mfma_f32_32x32x2_f32 (round 3; the VALU + LDS kernels of small_convs.hip they replace ran
// 39 / 83 / 51 us at (8,24^3) against ~20 us of HBM time each):
//   forward : y[v][co]   = act(b[co] + sum_k x[v][k] W[k][co]),            x = [xa | xb]  (the concat is never materialised)
//   dgrad   : dx[v][ci]  = (sum_co dz[v][co] W[ci][co]) * (y_in[v][ci] > 0)   (ReLU masks of the two producers)
//   wgrad   : dW[ci][co] = sum_v x[v][ci] dz[v][co]
// MFMA lane (li = lane & 31, kh = lane >> 5): A = A[row li][k = kh], B = B[k = kh][col li]; accumulator register r of that lane is
// D[row (r & 3) + 8 (r >> 2) + 4 kh][col li].  The weights (32 KB) live in REGISTERS for the whole kernel (forward, dgrad: 128 per
// lane), the voxel operand comes straight from global memory with 16-B loads -- nothing is staged in LDS.
// Storage type T = float or bf16 bits (uint16_t); arithmetic is fp32 in both (exact products of the stored values).
#include "fdn_common.h"

namespace {

// 16 consecutive stored elements -> 16 floats (4 x 16-B loads for fp32, 2 for bf16)
__device__ __forceinline__ void ld16(const float* p, float (&v)[16]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) { const f32x4 t = *(const f32x4*)(p + 4 * q); v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w; }
}
__device__ __forceinline__ void ld16(const uint16_t* p, float (&v)[16]) {
    float a[8], b[8];
    FdnVec<uint16_t>::ld(p, a); FdnVec<uint16_t>::ld(p + 8, b);
#pragma unroll
    for (int q = 0; q < 8; ++q) { v[q] = a[q]; v[8 + q] = b[q]; }
}

// ---- forward.  Wave = persistent over blocks of 32 voxels.  K = 128 input channels as 64 steps of the pair (xa[s], xb[s]):
// lane half kh = 0 reads voxel li's row of xa, kh = 1 its row of xb (64 contiguous elements each); B[k][co] = W[64 kh + s][32 nt + li].
template <typename T>
__global__ __launch_bounds__(256, 2) void conv1x1_fwd_mfma_kernel(const T* __restrict__ xa, const T* __restrict__ xb,
                                                                  const float* __restrict__ w, const float* __restrict__ bias,
                                                                  T* __restrict__ y, int64_t nvox, int act, float alpha) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, kh = lane >> 5;
    float wr[2][64];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int s = 0; s < 64; ++s) wr[nt][s] = w[(64 * kh + s) * 64 + 32 * nt + li];
    const float b0 = bias ? bias[li] : 0.f, b1 = bias ? bias[32 + li] : 0.f;
    const float slope = act == FDN_ACT_RELU ? 0.f : (act == FDN_ACT_LEAKY ? alpha : 1.f);
    const T* xsrc = kh ? xb : xa;
    const int64_t nblk = (nvox + 31) / 32;
    for (int64_t blk = (int64_t)blockIdx.x * 4 + wave; blk < nblk; blk += (int64_t)gridDim.x * 4) {
        int64_t v = blk * 32 + li;
        v = v < nvox ? v : nvox - 1;
        const T* row = xsrc + v * 64;
        f32x16 acc0, acc1;
#pragma unroll
        for (int q = 0; q < 16; ++q) { acc0[q] = b0; acc1[q] = b1; }
#pragma unroll
        for (int c = 0; c < 4; ++c) {                       // 16 input channels per chunk
            float a[16];
            ld16(row + 16 * c, a);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], wr[0][16 * c + s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], wr[1][16 * c + s], acc1, 0, 0, 0);
            }
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

// ---- dgrad.  K = 64 output channels as 32 steps of the pair (dz[s], dz[32 + s]): lane half kh reads its half row of voxel li;
// B[k = co][col = ci] = W[ci = 32 nt + li][co = 32 kh + s], nt = 0..3 (ci 0..63 -> dxa, 64..127 -> dxb): 128 weight registers
// (+ 32 operand, 32 accumulator, 32 mask registers: one workgroup per CU, no spills).
template <typename T>
__global__ __launch_bounds__(256, 1) void conv1x1_dgrad_mfma_kernel(const T* __restrict__ dz, const float* __restrict__ w,
                                                                    const T* __restrict__ ya, const T* __restrict__ yb,
                                                                    T* __restrict__ dxa, T* __restrict__ dxb, int64_t nvox) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 31, kh = lane >> 5;
    float wr[4][32];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
            const f32x4 t = *(const f32x4*)(w + (32 * nt + li) * 64 + 32 * kh + 4 * s4);
            wr[nt][4 * s4] = t.x; wr[nt][4 * s4 + 1] = t.y; wr[nt][4 * s4 + 2] = t.z; wr[nt][4 * s4 + 3] = t.w;
        }
    const int64_t nblk = (nvox + 31) / 32;
    for (int64_t blk = (int64_t)blockIdx.x * 4 + wave; blk < nblk; blk += (int64_t)gridDim.x * 4) {
        int64_t v = blk * 32 + li;
        v = v < nvox ? v : nvox - 1;
        const T* row = dz + v * 64 + 32 * kh;
        float a[32];
        {
            float lo[16], hi[16];
            ld16(row, lo); ld16(row + 16, hi);
#pragma unroll
            for (int s = 0; s < 16; ++s) { a[s] = lo[s]; a[16 + s] = hi[s]; }
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {              // input channels 0..63 (-> dxa, mask ya), then 64..127 (-> dxb, mask yb)
            f32x16 acc0, acc1;
#pragma unroll
            for (int q = 0; q < 16; ++q) { acc0[q] = 0.f; acc1[q] = 0.f; }
#pragma unroll
            for (int s = 0; s < 32; ++s) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], wr[2 * half][s], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], wr[2 * half + 1][s], acc1, 0, 0, 0);
            }
            const T* ym = half ? yb : ya;
            T* dx = half ? dxb : dxa;
            // the masks of a block are requested together, before the first store (loads next to their stores would serialise)
            float m0[16], m1[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                int64_t vo = blk * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh;
                vo = vo < nvox ? vo : nvox - 1;
                m0[q] = fdn_ld1(ym + vo * 64 + li);
                m1[q] = fdn_ld1(ym + vo * 64 + 32 + li);
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int64_t vo = blk * 32 + (q & 3) + 8 * (q >> 2) + 4 * kh;
                if (vo < nvox) {
                    fdn_st1(dx + vo * 64 + li, m0[q] > 0.f ? acc0[q] : 0.f);
                    fdn_st1(dx + vo * 64 + 32 + li, m1[q] > 0.f ? acc1[q] : 0.f);
                }
            }
        }
    }
}

// ---- wgrad.  D[ci][co] over K = voxels (pairs: lane half kh takes voxel 2 p + kh).  One 16-B load of a voxel's x row gives lane li the
// channels 4 li .. 4 li + 3 of [xa | xb] = the A operands of FOUR row tiles (tile j holds ci = 4 i + j), one 8-B load of its dz row the
// channels 2 li, 2 li + 1 = the B operands of two column tiles (tile j holds co = 2 i + j): 2 loads per 8 MFMAs.  A wave keeps the
// 8 accumulator tiles for its whole voxel range; the four waves of a workgroup are summed through LDS (fixed order), one partial per
// workgroup goes to the workspace and reduce_partials adds the workgroups.
template <typename T> struct Ld4x;   // 4 channels of the activation tensor -> f32x4
template <> struct Ld4x<float> { __device__ __forceinline__ static f32x4 ld(const float* p) { return *(const f32x4*)p; } };
template <> struct Ld4x<uint16_t> { __device__ __forceinline__ static f32x4 ld(const uint16_t* p) { return fdn_ld4(p); } };

template <typename T>
__global__ __launch_bounds__(256, 1) void wgrad_1x1_mfma_kernel(const T* __restrict__ xa, const T* __restrict__ xb,
                                                                const T* __restrict__ dz, float* __restrict__ partial, int64_t nvox) {
    extern __shared__ __attribute__((aligned(16))) float red[];          // [4 waves][128 * 64]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int li = lane & 31, kh = lane >> 5;
    f32x16 acc[4][2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[mt][nt][q] = 0.f;
    const T* xsrc = li < 16 ? xa + 4 * li : xb + 4 * (li - 16);
    const int64_t npair = (nvox + 1) / 2;
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    const int64_t per_wave = (npair + nwaves - 1) / nwaves;
    const int64_t p_begin = ((int64_t)blockIdx.x * 4 + wave) * per_wave;
    const int64_t p_end = p_begin + per_wave < npair ? p_begin + per_wave : npair;
    constexpr int PF = 4;                                    // voxel pairs in flight
    f32x4 xq[PF];
    f32x2 zq[PF];
    auto fetch = [&](int slot, int64_t pr) {
        const int64_t v = 2 * pr + kh;
        if (pr < p_end && v < nvox) {
            xq[slot] = Ld4x<T>::ld(xsrc + v * 64);
            const f32x4 t = Ld4x<T>::ld(dz + v * 64 + 4 * (li >> 1));            // 16-B aligned load, this lane's pair selected below
            zq[slot] = (li & 1) ? (f32x2){t.z, t.w} : (f32x2){t.x, t.y};
        } else {
            xq[slot] = (f32x4){0.f, 0.f, 0.f, 0.f};
            zq[slot] = (f32x2){0.f, 0.f};
        }
    };
#pragma unroll
    for (int j = 0; j < PF; ++j) fetch(j, p_begin + j);
    for (int64_t pr = p_begin; pr < p_end; pr += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const f32x4 xv = xq[j];
            const f32x2 zv = zq[j];
            fetch(j, pr + PF + j);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                acc[mt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[mt], zv.x, acc[mt][0], 0, 0, 0);
                acc[mt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[mt], zv.y, acc[mt][1], 0, 0, 0);
            }
        }
    }
    // D[row i][col li'] of tile (mt, nt) is dW[ci = 4 i + mt][co = 2 li' + nt]
    float* mine = red + wave * (128 * 64);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int i = (q & 3) + 8 * (q >> 2) + 4 * kh;
                mine[(4 * i + mt) * 64 + 2 * li + nt] = acc[mt][nt][q];
            }
    __syncthreads();
    float* out = partial + (size_t)blockIdx.x * (128 * 64);
    for (int e = threadIdx.x; e < 128 * 64; e += 256)
        out[e] = (red[e] + red[128 * 64 + e]) + (red[2 * 128 * 64 + e] + red[3 * 128 * 64 + e]);
}

}  // namespace

template <typename T>
int fdn_conv1x1_fwd_mfma_launch(const T* xa, const T* xb, const float* w, const float* bias, T* y, int64_t nvox, int act, float alpha,
                                hipStream_t s) {
    const int64_t nblk = (nvox + 31) / 32;
    const int nb = (int)((nblk + 3) / 4 < 512 ? (nblk + 3) / 4 : 512);
    hipLaunchKernelGGL(conv1x1_fwd_mfma_kernel<T>, dim3(nb), dim3(256), 0, s, xa, xb, w, bias, y, nvox, act, alpha);
    FDN_CHECK_LAUNCH("conv1x1_fwd_mfma_kernel");
    return FDN_OK;
}

template <typename T>
int fdn_conv1x1_dgrad_mfma_launch(const T* dz, const float* w, const T* ya, const T* yb, T* dxa, T* dxb, int64_t nvox, hipStream_t s) {
    const int64_t nblk = (nvox + 31) / 32;
    const int nb = (int)((nblk + 3) / 4 < 512 ? (nblk + 3) / 4 : 512);
    hipLaunchKernelGGL(conv1x1_dgrad_mfma_kernel<T>, dim3(nb), dim3(256), 0, s, dz, w, ya, yb, dxa, dxb, nvox);
    FDN_CHECK_LAUNCH("conv1x1_dgrad_mfma_kernel");
    return FDN_OK;
}

// partial: nblocks x (128 * 64) floats (summed by reduce_partials, small_convs.hip)
template <typename T>
int fdn_wgrad_1x1_mfma_launch(const T* xa, const T* xb, const T* dz, float* partial, int nblocks, int64_t nvox, hipStream_t s) {
    constexpr int lds = 4 * 128 * 64 * (int)sizeof(float);
    if (int rc = fdn_func_max_lds((const void*)wgrad_1x1_mfma_kernel<T>, lds, "wgrad_1x1_mfma")) return rc;
    hipLaunchKernelGGL(wgrad_1x1_mfma_kernel<T>, dim3(nblocks), dim3(256), lds, s, xa, xb, dz, partial, nvox);
    FDN_CHECK_LAUNCH("wgrad_1x1_mfma_kernel");
    return FDN_OK;
}

#define FDN_INST_1X1(T)                                                                                                            \
    template int fdn_conv1x1_fwd_mfma_launch<T>(const T*, const T*, const float*, const float*, T*, int64_t, int, float, hipStream_t); \
    template int fdn_conv1x1_dgrad_mfma_launch<T>(const T*, const float*, const T*, const T*, T*, T*, int64_t, hipStream_t);       \
    template int fdn_wgrad_1x1_mfma_launch<T>(const T*, const T*, const T*, float*, int, int64_t, hipStream_t);
FDN_INST_1X1(float)
FDN_INST_1X1(uint16_t)

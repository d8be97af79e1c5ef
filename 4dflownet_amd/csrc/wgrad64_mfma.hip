// Weight gradient of the 3x3x3 64 -> 64 conv (Conv3DBackpropFilterV2 behind tape.gradient,
// src/Network/TrainerController.py:223, for the layers of src/Network/SR4DFlowNet.py:18-46):
//   dW[a,b,c][ci][co] = sum_{n,o} x[n, clamp(o + (a,b,c) - 1)][ci] * dz[n,o][co]
// as 27 GEMMs  (64 ci) x (64 co)  with the reduction running over all N*D*H*W voxels.
//
// Decomposition: grid = (S splits of the voxel-tile list) x (3 kernel-depth taps a).
//   A workgroup (4 waves) walks its share of box tiles (TD x TH x TW voxels).  Per tile it stages
//   the x rows the 9 taps (a fixed; b,c in 0..2) touch -- TD x (TH+2) x (TW+2) voxel rows with the
//   edge clamp applied while staging -- and the tile's dz rows into LDS, then every wave contracts over
//   the tile's voxels for its own 32x32 quadrant of (ci,co) and all 9 taps:
//     v_mfma_f32_32x32x2_f32  A[i=ci][k=voxel] = x row (ds_read_b32, lanes along ci: conflict-free),
//                             B[k=voxel][j=co] = dz row (shared by the 9 taps).
//   9 accumulators x 16 VGPRs stay in registers across ALL tiles of the workgroup; partial sums are
//   written once per workgroup to workspace[S][27][64][64] and a second kernel reduces over S.
//   x is re-read by the 3 depth-tap groups through L2; per tile a wave issues 9*TD*TH*TW/2 MFMAs
//   (64 cyc each) against 10 LDS reads per k-step, so the matrix pipe is the bottleneck by construction.
#include "fdn_common.h"

struct Wgrad64Args {
    const float* x;
    const float* dz;
    float* partial;
    int N, D, H, W;
    int ntd, nth, ntw, ntiles, S;
};

template <int TD, int TH, int TW>
__global__ __launch_bounds__(256, 2) void wgrad64_mfma_kernel(Wgrad64Args p) {
    constexpr int XH = TH + 2, XW = TW + 2;
    constexpr int XROWS = TD * XH * XW;
    constexpr int ZROWS = TD * TH * TW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs = smem;
    char* zs = smem + XROWS * 256;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int kh = lane >> 5;
    const int mq = wave & 1, nq = wave >> 1;
    const int a = blockIdx.y;        // kernel-depth tap
    const int split = blockIdx.x;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int tiles_per_n = p.ntd * p.nth * p.ntw;
    const int c16 = tid & 15;   // 16-B chunk within a 256-B row
    const int rsub = tid >> 4;  // 0..15

    // Tile staging is software-pipelined through registers: the K loop below reads only LDS (no vector-memory waits), so
    // the global loads of tile t+1 issued before it stay in flight under tile t's MFMAs and are written to LDS afterwards.
    constexpr int XP = (XROWS + 15) / 16, ZP = (ZROWS + 15) / 16;     // float4 per thread for the x rows / dz rows
    f32x4 xv[XP], zv[ZP];
    auto prefetch = [&](int tile) {
        int b = tile;
        const int n = b / tiles_per_n;
        b -= n * tiles_per_n;
        const int tdi = b / (p.nth * p.ntw);
        b -= tdi * (p.nth * p.ntw);
        const int thi = b / p.ntw;
        const int p0d = tdi * TD, p0h = thi * TH, p0w = (b - thi * p.ntw) * TW;
        const size_t vox_n = (size_t)n * p.D * p.H * p.W;
#pragma unroll
        for (int u = 0; u < XP; ++u) {          // x rows, edge clamp applied here
            const int r = u * 16 + rsub;
            xv[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (r < XROWS) {
                const int zd = r / (XH * XW);
                const int r2 = r - zd * (XH * XW);
                const int zh = r2 / XW;
                const int qd = min(max(p0d + zd + a - 1, 0), p.D - 1);
                const int qh = min(max(p0h + zh - 1, 0), p.H - 1);
                const int qw = min(max(p0w + (r2 - zh * XW) - 1, 0), p.W - 1);
                xv[u] = *(const f32x4*)(p.x + (vox_n + ((size_t)qd * p.H + qh) * p.W + qw) * 64 + c16 * 4);
            }
        }
#pragma unroll
        for (int u = 0; u < ZP; ++u) {          // dz rows, zero outside the volume
            const int r = u * 16 + rsub;
            zv[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (r < ZROWS) {
                const int zd = r / (TH * TW);
                const int r2 = r - zd * (TH * TW);
                const int zh = r2 / TW;
                const int qd = p0d + zd, qh = p0h + zh, qw = p0w + (r2 - zh * TW);
                if (qd < p.D && qh < p.H && qw < p.W)
                    zv[u] = *(const f32x4*)(p.dz + (vox_n + ((size_t)qd * p.H + qh) * p.W + qw) * 64 + c16 * 4);
            }
        }
    };
    if (split < p.ntiles) prefetch(split);
    for (int tile = split; tile < p.ntiles; tile += p.S) {
        __syncthreads();   // previous tile fully consumed
#pragma unroll
        for (int u = 0; u < XP; ++u) {
            const int r = u * 16 + rsub;
            if (r < XROWS) *(f32x4*)(xs + r * 256 + c16 * 16) = xv[u];
        }
#pragma unroll
        for (int u = 0; u < ZP; ++u) {
            const int r = u * 16 + rsub;
            if (r < ZROWS) *(f32x4*)(zs + r * 256 + c16 * 16) = zv[u];
        }
        __syncthreads();
        if (tile + p.S < p.ntiles) prefetch(tile + p.S);

        // ---- contract over the tile's voxels, two per MFMA (lane half kh picks the voxel of the pair) ----
        const char* xa = xs + (mq * 32 + li) * 4 + kh * 256;
        const char* zb = zs + (nq * 32 + li) * 4 + kh * 256;
#pragma unroll 1
        for (int kd = 0; kd < TD; ++kd) {
#pragma unroll 1
            for (int kk = 0; kk < TH; ++kk) {
                const char* xr = xa + ((kd * XH + kk) * XW) * 256;
                const char* zr = zb + ((kd * TH + kk) * TW) * 256;
                // the 10 LDS reads of voxel pair w2 + 1 are issued before the 9 MFMAs of pair w2 (pinned: hipcc would sink them)
                float bq[2], aq[2][9];
                auto issue = [&](int w2, float& bv, float (&av)[9]) {
                    bv = *(const float*)(zr + w2 * 512);
#pragma unroll
                    for (int t = 0; t < 9; ++t) av[t] = *(const float*)(xr + w2 * 512 + ((t / 3) * XW + (t % 3)) * 256);
                };
                issue(0, bq[0], aq[0]);
#pragma unroll
                for (int w2 = 0; w2 < TW / 2; ++w2) {
                    if (w2 + 1 < TW / 2) issue(w2 + 1, bq[(w2 + 1) & 1], aq[(w2 + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < 9; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[w2 & 1][t], bq[w2 & 1], acc[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }

    // ---- write this workgroup's partial dW for taps (a, b, c) ----
    float* out = p.partial + ((size_t)split * 27 + a * 9) * 4096;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ci = mq * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            out[(size_t)t * 4096 + ci * 64 + nq * 32 + li] = acc[t][r];
        }
}

// dw[e] = sum_s partial[s][e]   (e over 27*64*64).  Block = 64 float4 columns x 4 quarters of S (combined through LDS in a
// fixed order): 432 blocks / 1728 waves keep enough loads in flight to stream the 75 MB of partials; the first version
// (108 blocks, one thread per column over all S) left more than half of the CUs idle.
__global__ __launch_bounds__(256) void wgrad64_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int S) {
    __shared__ f32x4 red[3][64];
    const int col = threadIdx.x & 63, qtr = threadIdx.x >> 6;
    const int e4 = blockIdx.x * 64 + col;                       // 27*1024 float4 columns, a multiple of 64
    const f32x4* p = (const f32x4*)partial + e4;
    const int s0q = (S * qtr) >> 2, s1q = (S * (qtr + 1)) >> 2;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0, s2 = s0, s3 = s0;
    int s = s0q;
    for (; s + 4 <= s1q; s += 4) {
        s0 += p[(size_t)(s + 0) * (27 * 1024)];
        s1 += p[(size_t)(s + 1) * (27 * 1024)];
        s2 += p[(size_t)(s + 2) * (27 * 1024)];
        s3 += p[(size_t)(s + 3) * (27 * 1024)];
    }
    for (; s < s1q; ++s) s0 += p[(size_t)s * (27 * 1024)];
    const f32x4 t = (s0 + s1) + (s2 + s3);
    if (qtr) red[qtr - 1][col] = t;
    __syncthreads();
    if (qtr == 0) ((f32x4*)dw)[e4] = (t + red[0][col]) + (red[1][col] + red[2][col]);
}

namespace {
constexpr int kTD = 2, kTH = 4, kTW = 8;
int wgrad64_splits(int N, int D, int H, int W) {
    const long long ntiles = (long long)N * ((D + kTD - 1) / kTD) * ((H + kTH - 1) / kTH) * ((W + kTW - 1) / kTW);
    // 3 tap groups x S workgroups; two workgroups fit a CU -> aim for ~512 resident, at least 4 tiles each
    long long S = 170;                 // 3*170 = 510 workgroups ~ 2 per CU
    if (ntiles / 4 < S) S = ntiles / 4 > 0 ? ntiles / 4 : 1;
    return (int)S;
}
}  // namespace

size_t fdn_wgrad64_workspace_bytes(int N, int D, int H, int W) {
    return (size_t)wgrad64_splits(N, D, H, W) * 27 * 4096 * sizeof(float);
}

int fdn_wgrad64_launch(const float* x, const float* dz, float* dw, void* ws, size_t ws_bytes, int N, int D, int H,
                       int W, hipStream_t s) {
    Wgrad64Args a;
    a.x = x; a.dz = dz; a.partial = (float*)ws;
    a.N = N; a.D = D; a.H = H; a.W = W;
    a.ntd = (D + kTD - 1) / kTD; a.nth = (H + kTH - 1) / kTH; a.ntw = (W + kTW - 1) / kTW;
    a.ntiles = N * a.ntd * a.nth * a.ntw;
    a.S = wgrad64_splits(N, D, H, W);
    const size_t lds = (size_t)(kTD * (kTH + 2) * (kTW + 2) + kTD * kTH * kTW) * 256;
    hipLaunchKernelGGL((wgrad64_mfma_kernel<kTD, kTH, kTW>), dim3(a.S, 3), dim3(256), lds, s, a);
    FDN_CHECK_LAUNCH("wgrad64_mfma_kernel");
    hipLaunchKernelGGL(wgrad64_reduce_kernel, dim3(27 * 1024 / 64), dim3(256), 0, s, (const float*)ws, dw, a.S);
    FDN_CHECK_LAUNCH("wgrad64_reduce_kernel");
    return FDN_OK;
}

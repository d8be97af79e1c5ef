// Weight gradient of the 3x3x3 64 -> 64 conv (Conv3DBackpropFilterV2 behind tape.gradient,
// src/Network/TrainerController.py:223, for the layers of src/Network/SR4DFlowNet.py:18-46):
//   dW[a,b,c][ci][co] = sum_{n,o} x[n, clamp(o + (a,b,c) - 1)][ci] * dz[n,o][co]
// as 27 GEMMs  (64 ci) x (64 co)  with the reduction running over all N*D*H*W voxels.
//
// Decomposition: grid = (S splits of the voxel-tile list) x (3 kernel-depth taps a).
//   A workgroup (4 waves) walks its share of box tiles (1 x TH x TW voxels).  Per tile it stages
//   the x rows the 9 taps (a fixed; b,c in 0..2) touch -- (TH+2) x (TW+2) voxel rows with the
//   edge clamp applied while staging -- and the tile's dz rows into LDS, then every wave contracts over
//   the tile's voxels for its own 32x32 quadrant of (ci,co) and all 9 taps:
//     v_mfma_f32_32x32x2_f32  A[i=ci][k=voxel] = x row (ds_read_b32, lanes along ci: conflict-free),
//                             B[k=voxel][j=co] = dz row (shared by the 9 taps).
//   9 accumulators x 16 VGPRs stay in registers across ALL tiles of the workgroup; partial sums are
//   written once per workgroup to workspace[S][27][64][64] and a second kernel reduces over S.
//   x is re-read by the 3 depth-tap groups through L2; per tile a wave issues 9*TH*TW/2 MFMAs
//   (64 cyc each) against 10 LDS reads per k-step, so the matrix pipe is the bottleneck by construction.
#include "fdn_common.h"

struct Wgrad64Args {
    const float* x;
    const float* dz;
    float* partial;
    int N, D, H, W;
    int ntd, nth, ntw, ntiles, S;
    unsigned bytes;              // size of x (= of dz) in bytes; < 4 GB (checked by the launcher)
};

// ---------------------------------------------------------------------------------------------------------------------
// Tile pipeline: tiles of 1 x TH x TW voxels go through THREE LDS buffers, so that nothing but one barrier per tile
// interrupts the MFMA stream of a wave.  While tile k is contracted out of buffer k%3:
//   pairs 0..2   the registers holding tile k+1 (loaded during tile k-1) are written to buffer (k+1)%3,
//   pairs 3..10  the rows of tile k+2 are loaded into those registers (buffer loads, scalar tile offsets),
//   pair  8      one barrier: every wave has written its part of tile k+1 (and finished tile k-1, so buffer (k+2)%3 may be
//                overwritten during tile k+1) -- with two buffers a wave running at twice the speed of a sibling (its SIMD
//                partner idle) could overwrite rows the sibling still reads,
//   pair  15     the first LDS reads of tile k+1 are issued: the read-ahead ring crosses tile boundaries.
// Measured on MI355X: the first version (2x4x8 tiles, one buffer, barrier / LDS write / barrier / prefetch between the K
// loops of consecutive tiles) spent ~8 % of the launch there (ablation: 1.445 -> 1.342 ms at 8x48^3 without staging) and
// co-resident workgroups did not fill those bubbles; this pipeline runs 1.393 ms (0.210 -> 0.196 ms at 8x24^3).
// ---------------------------------------------------------------------------------------------------------------------
template <int TH, int TW>
__global__ __launch_bounds__(256, 2) void wgrad64_pipe_kernel(Wgrad64Args p) {
    constexpr int XH = TH + 2, XW = TW + 2;
    constexpr int XROWS = XH * XW, ZROWS = TH * TW;
    constexpr int BUFB = (XROWS + ZROWS) * 256;
    constexpr int XP = (XROWS + 15) / 16, ZP = (ZROWS + 15) / 16;
    constexpr int NPAIR = TH * TW / 2, PPR = TW / 2;
    static_assert(NPAIR % 2 == 0 && NPAIR >= 12, "two register sets; the pipeline stages are pinned to pairs 0..10");
    static_assert(XP <= 4 && ZP <= 2, "stage placement below");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int kh = lane >> 5;
    const int mq = wave & 1, nq = wave >> 1;
    const int a = blockIdx.y;        // kernel-depth tap
    const int split = blockIdx.x;
    const int c16 = tid & 15;        // 16-B chunk within a 256-B row
    const int rsub = tid >> 4;       // 0..15

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // per-thread staged rows: (h,w) inside the halo box / the tile do not depend on the tile
    int xh[XP], xw[XP], zh[ZP], zw[ZP];
#pragma unroll
    for (int u = 0; u < XP; ++u) { const int r = min(u * 16 + rsub, XROWS - 1); xh[u] = r / XW; xw[u] = r - xh[u] * XW; }
#pragma unroll
    for (int u = 0; u < ZP; ++u) { const int r = min(u * 16 + rsub, ZROWS - 1); zh[u] = r / TW; zw[u] = r - zh[u] * TW; }

    // tile walk: tile = split + k*S, decoded incrementally (n, d, th, tw) with S pre-split the same way -- scalar work only
    const int per_d = p.nth * p.ntw, per_n = p.D * per_d;
    int tn, td, th, tw;
    {
        int b = split;
        tn = b / per_n; b -= tn * per_n;
        td = b / per_d; b -= td * per_d;
        th = b / p.ntw; tw = b - th * p.ntw;
    }
    int sn, sd, sh, sw;
    {
        int b = p.S;
        sn = b / per_n; b -= sn * per_n;
        sd = b / per_d; b -= sd * per_d;
        sh = b / p.ntw; sw = b - sh * p.ntw;
    }
    const int nk = (p.ntiles - split + p.S - 1) / p.S;      // tiles of this workgroup (>= 1 by construction of S)
    int kload = 0;                                          // index of the tile the cursor (tn,td,th,tw) points at
    auto advance = [&]() {                                  // cursor -> next tile of this workgroup (stays on the last one)
        if (kload + 1 < nk) {
            ++kload;
            tw += sw; if (tw >= p.ntw) { tw -= p.ntw; ++th; }
            th += sh; if (th >= p.nth) { th -= p.nth; ++td; }
            td += sd; if (td >= p.D) { td -= p.D; ++tn; }
            tn += sn;
        }
    };

    // Staged rows come in through buffer loads: the tensor is the buffer, the (sample, depth plane, tile origin) part of the
    // address is a scalar offset and the row part a per-thread constant, so a tile whose halo box lies inside its plane costs
    // NO vector ALU work (on this part every VALU instruction takes its cycles from the fp32 MFMA stream); only tiles on the
    // plane border compute clamped per-thread offsets.  dz rows outside the volume read 0 through the range check.
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.dz, 0, p.bytes, 0x00020000);
    unsigned xc[XP], zc[ZP];
#pragma unroll
    for (int u = 0; u < XP; ++u) xc[u] = (unsigned)((xh[u] * p.W + xw[u]) * 256 + c16 * 16);
#pragma unroll
    for (int u = 0; u < ZP; ++u) zc[u] = (unsigned)((zh[u] * p.W + zw[u]) * 256 + c16 * 16);
    f32x4 xv[XP], zv[ZP];
    auto bload = [&](__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)vo, (int)so, 0));
    };
    auto load_x = [&](int u) {                              // x row (edge clamp applied here) of the cursor tile
        const int qd = min(max(td + a - 1, 0), p.D - 1);
        const unsigned plane = (unsigned)((tn * p.D + qd) * p.H * p.W) * 256u;
        const int h0 = th * TH - 1, w0 = tw * TW - 1;
        if (h0 >= 0 && h0 + XH <= p.H && w0 >= 0 && w0 + XW <= p.W) {
            xv[u] = bload(xrs, xc[u], plane + (unsigned)((h0 * p.W + w0) * 256));
        } else {
            const int qh = min(max(h0 + xh[u], 0), p.H - 1);
            const int qw = min(max(w0 + xw[u], 0), p.W - 1);
            xv[u] = bload(xrs, (unsigned)((qh * p.W + qw) * 256 + c16 * 16), plane);
        }
    };
    auto load_z = [&](int u) {                              // dz row, zero outside the volume
        const unsigned plane = (unsigned)((tn * p.D + td) * p.H * p.W) * 256u;
        const int h0 = th * TH, w0 = tw * TW;
        if (h0 + TH <= p.H && w0 + TW <= p.W) {
            zv[u] = bload(zrs, zc[u], plane + (unsigned)((h0 * p.W + w0) * 256));
        } else {
            const int qh = h0 + zh[u], qw = w0 + zw[u];
            zv[u] = bload(zrs, qh < p.H && qw < p.W ? (unsigned)((qh * p.W + qw) * 256 + c16 * 16) : 0xffffffffu, plane);
        }
    };
    auto write_x = [&](int u, char* buf) {
        const int r = u * 16 + rsub;
        if ((u + 1) * 16 <= XROWS || r < XROWS) *(f32x4*)(buf + r * 256 + c16 * 16) = xv[u];      // branch-free for full passes
    };
    auto write_z = [&](int u, char* buf) {
        const int r = u * 16 + rsub;
        if ((u + 1) * 16 <= ZROWS || r < ZROWS) *(f32x4*)(buf + XROWS * 256 + r * 256 + c16 * 16) = zv[u];
    };

    // ---- prologue: tile 0 -> buffer 0, tile 1 -> registers ----
#pragma unroll
    for (int u = 0; u < XP; ++u) load_x(u);
#pragma unroll
    for (int u = 0; u < ZP; ++u) load_z(u);
#pragma unroll
    for (int u = 0; u < XP; ++u) write_x(u, smem);
#pragma unroll
    for (int u = 0; u < ZP; ++u) write_z(u, smem);
    advance();
#pragma unroll
    for (int u = 0; u < XP; ++u) load_x(u);
#pragma unroll
    for (int u = 0; u < ZP; ++u) load_z(u);
    __syncthreads();

    const int lane_x = (mq * 32 + li) * 4 + kh * 256;
    const int lane_z = XROWS * 256 + (nq * 32 + li) * 4 + kh * 256;
    float bq[2], aq[2][9];
    auto issue = [&](const char* buf, int q, float& bv, float (&av)[9]) {     // the 10 LDS reads of voxel pair q
        const int kk = q / PPR, w2 = q % PPR;
        bv = *(const float*)(buf + lane_z + (kk * TW) * 256 + w2 * 512);
#pragma unroll
        for (int t = 0; t < 9; ++t) av[t] = *(const float*)(buf + lane_x + ((kk + t / 3) * XW + (t % 3)) * 256 + w2 * 512);
    };
    int bcur = 0;                                           // buffer of tile k
    issue(smem, 0, bq[0], aq[0]);
#pragma unroll 1
    for (int k = 0; k < nk; ++k) {
        const int bnxt = bcur == 2 ? 0 : bcur + 1;
        const char* cur = smem + bcur * BUFB;
        char* nxt = smem + bnxt * BUFB;
#pragma unroll
        for (int q = 0; q < NPAIR; ++q) {
            if (q == 8) __syncthreads();
            if (q + 1 < NPAIR) issue(cur, q + 1, bq[(q + 1) & 1], aq[(q + 1) & 1]);
            else issue(nxt, 0, bq[0], aq[0]);
            __builtin_amdgcn_sched_barrier(0);
            if (q == 0) { write_x(0, nxt); if (XP > 1) write_x(1, nxt); }
            if (q == 1) { if (XP > 2) write_x(2, nxt); if (XP > 3) write_x(3, nxt); }
            if (q == 2) { write_z(0, nxt); if (ZP > 1) write_z(1, nxt); }
            if (q == 3) advance();
            if (q >= 4 && q < 4 + XP) load_x(q - 4);
            if (q >= 9 && q < 9 + ZP) load_z(q - 9);
#pragma unroll
            for (int t = 0; t < 9; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[q & 1][t], bq[q & 1], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        bcur = bnxt;
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
constexpr int kTH = 4, kTW = 8;          // pipelined kernel: tiles of 1 x 4 x 8 voxels
int wgrad64_splits(int N, int D, int H, int W) {
    const long long ntiles = (long long)N * D * ((H + kTH - 1) / kTH) * ((W + kTW - 1) / kTW);
    // 3 tap groups x S workgroups; two workgroups fit a CU -> aim for ~512 resident, at least 8 tiles each
    long long S = 170;                 // 3*170 = 510 workgroups ~ 2 per CU
    if (ntiles / 8 < S) S = ntiles / 8 > 0 ? ntiles / 8 : 1;
    return (int)S;
}
}  // namespace

// dw = sum over S of partial[S][27][64][64] (shared with the Winograd kernel, whose workgroups write the same layout)
int fdn_wgrad64_reduce_launch(const float* partial, float* dw, int S, hipStream_t s) {
    hipLaunchKernelGGL(wgrad64_reduce_kernel, dim3(27 * 1024 / 64), dim3(256), 0, s, partial, dw, S);
    FDN_CHECK_LAUNCH("wgrad64_reduce_kernel");
    return FDN_OK;
}

size_t fdn_wgrad64_workspace_bytes(int N, int D, int H, int W) {
    return (size_t)wgrad64_splits(N, D, H, W) * 27 * 4096 * sizeof(float);
}

int fdn_wgrad64_launch(const float* x, const float* dz, float* dw, void* ws, size_t ws_bytes, int N, int D, int H,
                       int W, hipStream_t s) {
    Wgrad64Args a;
    a.x = x; a.dz = dz; a.partial = (float*)ws;
    a.N = N; a.D = D; a.H = H; a.W = W;
    a.ntd = D; a.nth = (H + kTH - 1) / kTH; a.ntw = (W + kTW - 1) / kTW;
    a.ntiles = N * a.ntd * a.nth * a.ntw;
    a.S = wgrad64_splits(N, D, H, W);
    FDN_REQUIRE((long long)N * D * H * W * 256 < (1ll << 32), "wgrad64: x of %dx%dx%dx%dx64 floats exceeds the 32-bit buffer addressing", N, D, H, W);
    a.bytes = (unsigned)((long long)N * D * H * W * 256);
    const size_t lds = (size_t)3 * ((kTH + 2) * (kTW + 2) + kTH * kTW) * 256;
    if (int rc = fdn_func_max_lds((const void*)wgrad64_pipe_kernel<kTH, kTW>, (int)lds, "wgrad64")) return rc;
    hipLaunchKernelGGL((wgrad64_pipe_kernel<kTH, kTW>), dim3(a.S, 3), dim3(256), lds, s, a);
    FDN_CHECK_LAUNCH("wgrad64_pipe_kernel");
    hipLaunchKernelGGL(wgrad64_reduce_kernel, dim3(27 * 1024 / 64), dim3(256), 0, s, (const float*)ws, dw, a.S);
    FDN_CHECK_LAUNCH("wgrad64_reduce_kernel");
    return FDN_OK;
}

// conv3d 3x3x3, 64 -> 64 channels, fp32, NDHWC: implicit GEMM on v_mfma_f32_32x32x2_f32.
//
// Replaces tf.pad(SYMMETRIC) + Conv3D(valid) + BiasAdd + ReLU/LeakyReLU (+ the resnet_block add)
// of src/Network/SR4DFlowNet.py:93-120, and -- in "zero" boundary mode on the padded output grid --
// Conv3DBackpropInputV2 of the same layers.
//
// Work decomposition (one workgroup = 4 waves = one output box tile of up to 128*MT voxels x 64 cout):
//   M = output voxels of the tile, N = 64 cout, K = 27 taps x 64 cin.
//   * The input box + 1-voxel halo is staged ONCE per cin-half into LDS (128 B per voxel row,
//     16-B chunks XOR-swizzled by row so the A-fragment ds_read_b128 spreads over all banks);
//     the boundary rule (edge clamp for forward == SYMMETRIC p=1, zero for dgrad) is applied while
//     staging, so the K loop is branch-free.
//   * Every wave owns 32*MT voxel rows x all 64 cout: MT x 2 accumulators of 32x32 (16 VGPRs each).
//   * A fragments: one ds_read_b128 per (M-tile, 8 cin) -- lane (i,kh) gets cin 8g+4kh+{0..3} of voxel i;
//     MFMA step s contracts the cin pair {8g+s, 8g+4+s}.  The K order is a permutation of cin, matched
//     by the packed weight stream, so no data movement is needed to form fragments.
//   * B fragments (weights) are NOT staged: the packed stream [half][tap][g][kh][cout][s] is read
//     straight from L1/L2 with one global_load_dwordx4 per (N-tile, 8 cin) and software-prefetched one
//     tap ahead.  442 KB of weights are shared by every workgroup on the chip, so they stay cache-resident.
//   * 2 workgroups per CU (<= 80 KB LDS, <= 256 VGPRs): one stages while the other issues MFMAs.
//   fp32 MFMA rate is 64 cyc / instruction / SIMD: per tap a wave issues 16*MT*2 MFMAs against
//   4*MT LDS reads and 8 global loads, so the matrix pipe is the only busy resource by construction.
#include "fdn_common.h"

struct Conv64Args {
    const float* x;
    const float* wp;
    const float* bias;
    const float* res;
    float* y;
    int N, ID, IH, IW, OD, OH, OW;
    int off, zero_mode;
    int td, th, tw, ntd, nth, ntw;
    int hh, hw;                 // halo dims th+2, tw+2 (hd = td+2)
    int rows;                   // hd*hh*hw
    unsigned mg_hhhw, mg_hw;    // magic divisors for halo-row decomposition
    int act;
    float alpha;
};

template <int MT>
__global__ __launch_bounds__(256, 2) void conv64_mfma_kernel(Conv64Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int* goff = (int*)(smem + p.rows * 128);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int kh = lane >> 5;

    // ---- which tile ----
    const int tiles_per_n = p.ntd * p.nth * p.ntw;
    int b = blockIdx.x;
    const int n = b / tiles_per_n;
    b -= n * tiles_per_n;
    const int tdi = b / (p.nth * p.ntw);
    b -= tdi * (p.nth * p.ntw);
    const int thi = b / p.ntw;
    const int twi = b - thi * p.ntw;
    const int p0d = tdi * p.td, p0h = thi * p.th, p0w = twi * p.tw;

    const int nv = p.td * p.th * p.tw;
    const int thtw = p.th * p.tw;

    // ---- output voxel offsets of the tile's M rows (read back in the epilogue) ----
    for (int m = tid; m < 128 * MT; m += 256) {
        int g = -1;
        if (m < nv) {
            const int md = m / thtw;
            const int r2 = m - md * thtw;
            const int mh = r2 / p.tw;
            const int mw = r2 - mh * p.tw;
            const int pd = p0d + md, ph = p0h + mh, pw = p0w + mw;
            if (pd < p.OD && ph < p.OH && pw < p.OW) g = ((n * p.OD + pd) * p.OH + ph) * p.OW + pw;
        }
        goff[m] = g;
    }

    // ---- this lane's A rows (LDS halo row of voxel m at tap (0,0,0)) ----
    int row0[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        int m = (wave * MT + mi) * 32 + li;
        m = m < nv ? m : nv - 1;
        const int md = m / thtw;
        const int r2 = m - md * thtw;
        const int mh = r2 / p.tw;
        const int mw = r2 - mh * p.tw;
        row0[mi] = (md * p.hh + mh) * p.hw + mw;
    }

    f32x16 acc[MT][2];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][nn][r] = 0.f;

    const int chunk = tid & 7;
    const int rsub = tid >> 3;
    const size_t in_n = (size_t)n * p.ID * p.IH * p.IW;

    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();  // everyone finished reading the previous half
        // ---- stage input box + halo, cin [32*half, 32*half+32) ----
        const float* xh = p.x + half * 32 + chunk * 4;
        constexpr int U = 5;
        for (int r0 = 0; r0 < p.rows; r0 += 32 * U) {
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + u * 32 + rsub;
                v[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (r < p.rows) {
                    const int zd = fdn_div20(r, p.mg_hhhw);
                    const int r2 = r - zd * p.hh * p.hw;
                    const int zh = fdn_div20(r2, p.mg_hw);
                    const int zw = r2 - zh * p.hw;
                    int qd = p0d + zd - 1 + p.off, qh = p0h + zh - 1 + p.off, qw = p0w + zw - 1 + p.off;
                    bool ok = true;
                    if (p.zero_mode) {
                        ok = (unsigned)qd < (unsigned)p.ID && (unsigned)qh < (unsigned)p.IH && (unsigned)qw < (unsigned)p.IW;
                    } else {
                        qd = min(max(qd, 0), p.ID - 1);
                        qh = min(max(qh, 0), p.IH - 1);
                        qw = min(max(qw, 0), p.IW - 1);
                    }
                    if (ok) v[u] = *(const f32x4*)(xh + (in_n + ((size_t)qd * p.IH + qh) * p.IW + qw) * 64);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r0 + u * 32 + rsub;
                if (r < p.rows) *(f32x4*)(smem + r * 128 + ((chunk ^ ((r >> 1) & 7)) << 4)) = v[u];
            }
        }
        __syncthreads();

        // ---- K loop over 27 taps x 32 cin ----
        const f32x4* bp = (const f32x4*)p.wp + (size_t)half * (27 * 4 * 128) + kh * 64 + li;
        f32x4 bcur[4][2], bnxt[4][2];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int nn = 0; nn < 2; ++nn) bcur[g][nn] = bp[g * 128 + nn * 32];

        int ta = 0, tb = 0, tc = 0;
#pragma unroll 1
        for (int tap = 0; tap < 27; ++tap) {
            const int tnext = tap < 26 ? tap + 1 : 26;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int nn = 0; nn < 2; ++nn) bnxt[g][nn] = bp[(tnext * 4 + g) * 128 + nn * 32];

            const int tapoff = (ta * p.hh + tb) * p.hw + tc;
            f32x4 av[MT][4];
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) {
                const int row = row0[mi] + tapoff;
                const int sw = (row >> 1) & 7;
                const char* base = smem + row * 128;
#pragma unroll
                for (int g = 0; g < 4; ++g) av[mi][g] = *(const f32x4*)(base + (((g * 2 + kh) ^ sw) << 4));
            }
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                        for (int nn = 0; nn < 2; ++nn)
                            acc[mi][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mi][g][s], bcur[g][nn][s], acc[mi][nn], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int nn = 0; nn < 2; ++nn) bcur[g][nn] = bnxt[g][nn];
            if (++tc == 3) { tc = 0; if (++tb == 3) { tb = 0; ++ta; } }
        }
    }

    // ---- epilogue: bias + residual + activation, 128-B row segments per half-wave ----
    float bv[2] = {0.f, 0.f};
    if (p.bias) { bv[0] = p.bias[li]; bv[1] = p.bias[32 + li]; }
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int mrow = (wave * MT + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            const int g = goff[mrow];
            if (g >= 0) {
                const size_t o = (size_t)g * 64 + li;
#pragma unroll
                for (int nn = 0; nn < 2; ++nn) {
                    float z = acc[mi][nn][r] + bv[nn];
                    if (p.res) z += p.res[o + nn * 32];
                    p.y[o + nn * 32] = fdn_act(z, p.act, p.alpha);
                }
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
// weight packing: Keras (27,64,64)[tap][cin][cout] -> operand streams [half][tap][g][kh][cout][s]
//   fwd  : cin = 32*half + 8g + 4kh + s, same tap
//   dgrad: contraction runs over cout of the layer, taps flipped: stream[..][ci][s] = w[26-tap][ci][co = 32*half+8g+4kh+s]
// --------------------------------------------------------------------------------------------
__global__ void pack_conv64_kernel(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wd) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // over 27*64*64 packed elements
    if (idx >= 27 * 64 * 64) return;
    const int s = idx & 3;
    const int j = (idx >> 2) & 63;
    const int kh = (idx >> 8) & 1;
    const int g = (idx >> 9) & 3;
    const int rest = idx >> 11;          // half*27 + tap
    const int half = rest / 27;
    const int tap = rest - half * 27;
    const int k = half * 32 + g * 8 + kh * 4 + s;
    if (wf) wf[idx] = w[(tap * 64 + k) * 64 + j];
    if (wd) wd[idx] = w[((26 - tap) * 64 + j) * 64 + k];
}

extern "C" int fdn_pack_conv64_weights(const float* w, float* wp_fwd, float* wp_dgrad, void* stream) {
    FDN_REQUIRE(w != nullptr, "fdn_pack_conv64_weights: w is NULL");
    hipLaunchKernelGGL(pack_conv64_kernel, dim3((27 * 64 * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w,
                       wp_fwd, wp_dgrad);
    FDN_CHECK_LAUNCH("fdn_pack_conv64_weights");
    return FDN_OK;
}

// --------------------------------------------------------------------------------------------
// host side: tile planning + launch
// --------------------------------------------------------------------------------------------
FdnTile fdn_plan_tile(int N, int OD, int OH, int OW, int max_vox, int max_halo_rows, int halo) {
    FdnTile best = {1, 1, 2, OD, OH, (OW + 1) / 2};
    double best_cost = 1e30;
    for (int td = 1; td <= OD && td <= max_vox; ++td)
        for (int th = 1; th <= OH && td * th <= max_vox; ++th)
            for (int tw = 1; tw <= OW && td * th * tw <= max_vox; ++tw) {
                const int rows = (td + halo) * (th + 2) * (tw + 2);
                if (rows > max_halo_rows) continue;
                const int ntd = (OD + td - 1) / td, nth = (OH + th - 1) / th, ntw = (OW + tw - 1) / tw;
                const double tiles = (double)N * ntd * nth * ntw;
                // per-tile cost: the M rows are always fully issued; staging adds a little per halo row
                const double per_tile = max_vox + 0.12 * rows;
                const double rounds = tiles <= 256 ? 1.0 : tiles / 256.0;   // balance over 256 CUs
                const double cost = (tiles <= 256 ? 1.0 : (double)((long long)((tiles + 255) / 256))) * per_tile * 0.5 +
                                    rounds * per_tile * 0.5;
                if (cost < best_cost) { best_cost = cost; best = {td, th, tw, ntd, nth, ntw}; }
            }
    return best;
}

template <int MT>
static int launch_conv64(Conv64Args& a, const FdnTile& t, hipStream_t s) {
    a.td = t.td; a.th = t.th; a.tw = t.tw; a.ntd = t.ntd; a.nth = t.nth; a.ntw = t.ntw;
    a.hh = t.th + 2; a.hw = t.tw + 2;
    a.rows = (t.td + 2) * a.hh * a.hw;
    a.mg_hhhw = fdn_magic20(a.hh * a.hw);
    a.mg_hw = fdn_magic20(a.hw);
    const size_t lds = (size_t)a.rows * 128 + 128 * MT * sizeof(int);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void*)conv64_mfma_kernel<MT>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
        if (e != hipSuccess) { fdn_set_error("conv64: hipFuncSetAttribute: %s", hipGetErrorString(e)); return FDN_ERR_HIP; }
        attr_set = true;
    }
    const long long grid = (long long)a.N * t.ntd * t.nth * t.ntw;
    hipLaunchKernelGGL(conv64_mfma_kernel<MT>, dim3((unsigned)grid), dim3(256), lds, s, a);
    FDN_CHECK_LAUNCH("conv64_mfma_kernel");
    return FDN_OK;
}

static double tile_cost(const FdnTile& t, int N, int mcap) {
    const double tiles = (double)N * t.ntd * t.nth * t.ntw;
    const double per_tile = mcap + 0.12 * (t.td + 2) * (t.th + 2) * (t.tw + 2);
    const double rounds_hi = (double)((long long)((tiles + 255) / 256));
    const double rounds = tiles <= 256 ? 1.0 : tiles / 256.0;
    return (0.5 * rounds_hi + 0.5 * rounds) * per_tile;
}

int fdn_conv64_force_mt = 0;   // test/bench hook: 0 = auto, 1 or 2 = force the M-tile count per wave

int fdn_conv64_launch(const float* x, const float* wpack, const float* bias, const float* residual, float* y, int N,
                      int ID, int IH, int IW, int OD, int OH, int OW, int off, int zero_mode, int act, float alpha,
                      hipStream_t s) {
    Conv64Args a;
    a.x = x; a.wp = wpack; a.bias = bias; a.res = residual; a.y = y;
    a.N = N; a.ID = ID; a.IH = IH; a.IW = IW; a.OD = OD; a.OH = OH; a.OW = OW;
    a.off = off; a.zero_mode = zero_mode; a.act = act; a.alpha = alpha;
    // <= 80 KB of LDS per workgroup so two fit a CU: 128 B per halo row + the row table
    const int max_rows = (81920 - 1024) / 128;
    const FdnTile t2 = fdn_plan_tile(N, OD, OH, OW, 256, max_rows, 2);
    const FdnTile t1 = fdn_plan_tile(N, OD, OH, OW, 128, max_rows, 2);
    int mt = tile_cost(t1, N, 128) < tile_cost(t2, N, 256) ? 1 : 2;
    if (fdn_conv64_force_mt == 1 || fdn_conv64_force_mt == 2) mt = fdn_conv64_force_mt;
    return mt == 1 ? launch_conv64<1>(a, t1, s) : launch_conv64<2>(a, t2, s);
}

extern "C" int fdn_debug_set_conv64_mt(int mt) { fdn_conv64_force_mt = mt; return FDN_OK; }

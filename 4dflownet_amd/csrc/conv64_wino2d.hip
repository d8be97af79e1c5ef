// conv3d 3x3x3, 64 -> 64 channels, fp32, NDHWC: 2-D Winograd -- F(HM,3) along H (HM = 4, or 2 where H is only even) on top of F(4,3)
// along W, direct taps along D -- on v_mfma_f32_16x16x4_f32.
//
// Same contract as conv64_wino.hip / conv64_mfma.hip (tf.pad(SYMMETRIC) + Conv3D + bias + activation + residual of
// src/Network/SR4DFlowNet.py:93-120 in clamp mode; Conv3DBackpropInputV2 of the same layers in zero mode with the interior
// MirrorPadGrad fused into the epilogue), with fewer multiplies than the 1-D kernel: a CELL of HM x 4 output voxels (h, w) is
// computed from its (HM+2) x 6 input patch as
//     Y = Ah^T [ sum_kd (Gh g Gw^T) .* (Bh^T x Bw) ] Aw             (Lavin & Gray)
// HM = 4: 36 products per (kd, cin, cout) for 16 voxels = 6.75 tap-equivalents per voxel; HM = 2: 24 for 8 voxels = 9 (1-D: 13.5,
// direct: 27).  On MI355X the fp32 matrix rate equals the fp32 vector rate, so fewer multiplies is the only way past the roofline.
// HM = 4 uses the interpolation points 0, +-3/4, +-3/2, inf on both axes (conv64_wino2d_kernel.h: a quarter of the fp32 error of the
// classic 0, +-1, +-2, inf, every constant exact), HM = 2 the classic points of rounds 2-4.
//
// The (HM+2) x 6 accumulator tiles a cell needs do not fit beside everything else, so the H coordinate xh runs as an OUTER,
// sequential loop ("stage"): a stage holds the six xw accumulators of its xh only (48 registers for 32 cells x 16 cout), runs the
// whole K loop of that xh (3 depth taps x 64 cin), and folds A_w^T M[xh] into the HM output rows with column xh of A_h^T
// (Y: HM x 4 voxels x 2 M-blocks x 4 = 64 or 128 registers).  A stage's input is V[xh] = sum_j B_h^T[xh][j] x[row j] -- two rows for
// F(2,3), three or four for F(4,3), two of the middle stages formed incrementally from the stage before them -- staged for all 64
// input channels at once, so the stages take the place of the cin slices of the 1-D kernel and read each input row several times
// (L1/L2 hits; what counts is the number of load instructions).
//
// Work decomposition:
//   * M = cells.  Workgroup (4 waves) = a box tile of td x ch x cw cells (<= 32 cells = 256 / 512 voxels) x 64 cout; wave w owns all
//     32 cells (two 16-cell MFMA blocks) x cout [16w, 16w+16): D[cout][cell] orientation, so a lane (cell c = lane & 15,
//     q = lane >> 4) holds cout 16w + 4q .. + 3 of one cell = 16-B stores.
//   * Staging: thread (staged cell-plane r, 16-B channel chunk) loads the row chunks of its stage (boundary rule through the staging
//     plan: clamp == SYMMETRIC p=1, or zero through the buffer range check), forms V, applies B_w^T and writes 6 xw planes of rows
//     [cell-plane][64 cin + 32-B pad]: the 288-B row stride makes the 16-row ds_read_b128 lane groups conflict-free.  A tile stages
//     (td + 2) depth planes x ch x cw cells <= 40 rows: 69 KB, 2 workgroups per CU.
//   * Weights: packed streams U[nb][xh][kd][xw][cin/16][lane][4] (fdn_pack_wino2d_one: 72 * 64 * 64 floats per layer and direction;
//     fdn_pack_wino44_one: 108 * 64 * 64, formed in double precision), read straight from L1/L2 by buffer_load_dwordx4.
//   * K loop of a stage: 3 depth taps x 6 xw x 4 cin groups; one step = 2 ds_read_b128 + 1 buffer_load_b128 feeding 8 MFMAs
//     (2 M-blocks x 4 k-steps, alternating blocks: the 40-cycle dependent latency of v_mfma_f32_16x16x4_f32 is covered);
//     operands through fragment rings refilled inside the MFMA stream, as in the 1-D kernel.
#include "fdn_common.h"
#include "conv64_pack.h"
#include <string.h>

#include "conv64_wino2d_kernel.h"
#include "conv64_wino2d_pc_kernel.h"

namespace {

FDN_HOOK_VAR(int, fdn_conv64_wino2d_dbg, 0);
FDN_HOOK_VAR(int, fdn_conv64_wino2d_tile, 0);          // test build: force the tile, td | ch << 8 | cw << 16 (0 = planner)

FDN_HOOK_VAR(int, fdn_conv64_wino2d_mb, 0);            // test build: force the M-blocks per wave (1 = half-size tiles, 2 = full; 0 = by the grid)

template <bool FUSED, int HM, int MB = 2, bool SPLIT = false, bool MASK = false>
__global__ __launch_bounds__(256, 2) void conv64_wino2d_kernel(Wino2Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    conv64_wino2d_body<FUSED, HM, kW2RDB, kW2RDA, kW2Dep, MB, SPLIT, MASK>(p, (int)blockIdx.x, smem);
}

// FDN_ALGO_WINO_BF16X3: the producer / consumer kernel (conv64_wino2d_pc_kernel.h), one persistent workgroup of 512 threads per CU
template <bool FUSED>
__global__ __launch_bounds__(kPcThreads, 1) void conv64_wino2d_pc_kernel(Wino2Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    conv64_wino2d_pc_body<FUSED>(p, smem);
}

int device_cus() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = __atomic_load_n(&cus[dev], __ATOMIC_RELAXED);
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        __atomic_store_n(&cus[dev], n, __ATOMIC_RELAXED);
    }
    return n;
}

#ifdef FDN_TEST_HOOKS
// Occupancy experiment: the same forward body at ONE workgroup per CU -- __launch_bounds__(256, 1) gives a
// wave the whole 512-register file (256 VGPR + 256 AGPR), the launch asks for more than half a CU's LDS so that a second workgroup
// cannot become resident -- with the product rings and with deeper ones.  Selected by fdn_debug_set_conv64_wino2d_variant.
FDN_HOOK_VAR(int, fdn_conv64_wino2d_variant, 0);
constexpr int kW2LdsOcc1 = 84 * 1024;
template <int RDB, int RDA, int DEP>
__global__ __launch_bounds__(256, 1) void conv64_wino2d_occ1_kernel(Wino2Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    conv64_wino2d_body<false, 2, RDB, RDA, DEP>(p, (int)blockIdx.x, smem);
}
#endif

// the F(2,3) x F(4,3) stream of one layer (72 * 4096 floats)
__global__ void pack_conv64_wino2d_kernel(const float* __restrict__ w, float* __restrict__ uf, float* __restrict__ ud) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < 72 * 64 * 64) fdn_pack_wino2d_one(w, uf, ud, idx);      // (the F(4,3) x F(4,3) streams: pack_conv64_wino44_kernel, conv64_mfma.hip)
}

struct Wino2Plan { int td, ch, cw; double cost; };
// half-size tiles below this many full-size tiles.  Measured (tools/bench_halftile.py, profiles/r6_halftile.txt; forward / fused dgrad, ms):
// 108 full tiles 0.070 / 0.084 -> 0.045 / 0.070, 216 (8 x 24^3) 0.083 / 0.110 -> 0.074 / 0.108, 324 0.130 / 0.168 -> 0.113 / 0.150,
// 432 0.139 / 0.189 -> 0.143 / 0.192 (two full workgroups on most CUs already), 1728 0.485 / 0.618 -> 0.520 / 0.668
constexpr long long kW2HalfBelow = 400;

// tile choice: every tile costs the MFMA time of 32 cells whatever its fill, plus the staging work of its rows and a fixed prologue /
// epilogue; the launch ends with the busiest CU (2 co-resident workgroups per CU share the matrix pipe, so work per CU = its tiles).
Wino2Plan wino2d_plan(int N, int ed, int ech, int ecw, int mb) {
    Wino2Plan best{1, 1, 1, 1e30};
    const int max_rows = mb == 2 ? kW2Rows : kW2RowsH;
    for (int ch = 1; ch <= ech && ch <= 4; ++ch)
        for (int cw = 1; cw <= ecw && ch * cw <= 4; ++cw)
            for (int td = 1; td <= ed && td * ch * cw <= 16 * mb; ++td) {
                const int rows = (td + 2) * ch * cw;
                if (rows > max_rows) continue;
                const double tiles = (double)N * ((ed + td - 1) / td) * ((ech + ch - 1) / ch) * ((ecw + cw - 1) / cw);
                const double per_tile = 16.0 * mb + 0.15 * rows + 2.0;
                const double rounds = 0.9 * (double)((long long)((tiles + 255) / 256)) + 0.1 * tiles / 256.0;
                const double cst = rounds * per_tile;
                if (cst < best.cost) best = {td, ch, cw, cst};
            }
    return best;
}

}  // namespace

// Is the 2-D Winograd kernel applicable to this output box of this input grid?  (H extent even, W extent a multiple of 4, a sample
// addressable with 30-bit byte offsets -- the staging plan adds a row offset and a column offset, each of which may be the
// "reads zero" marker 2^30.)
bool fdn_conv64_wino2d_ok(int ebd, int ebh, int ebw, int ID, int IH, int IW, int hm) {
    hm &= 7;                                            // (bit 3 of a launcher's hm argument: the bf16 x 3 products, FDN_ALGO_WINO_BF16X3)
    return (hm == 2 || hm == 4) && ebd > 0 && ebh >= hm && ebh % hm == 0 && ebw >= 4 && (ebw & 3) == 0 && (long long)ID * IH * IW <= (1ll << 22);
}

static_assert(sizeof(Wino2Args) <= sizeof(FdnWino2dPrepared::args), "FdnWino2dPrepared::args too small");

int fdn_conv64_wino2d_prepare(const float* x, const float* upack2, const float* bias, const float* residual, float* y,
                              const float* fskip, const float* fy, float* fout, int N, int ID, int IH, int IW, int OD, int OH,
                              int OW, int obd, int obh, int obw, int ebd, int ebh, int ebw, int off, int zero_mode, int act,
                              float alpha, int hm_arg, FdnWino2dPrepared* out, uint16_t* ymask, const uint16_t* fmask, const FdnExtraSrc* extra) {
    const int hm = hm_arg & 7;
    const bool split = (hm_arg & 8) != 0;               // F(4,3) x F(4,3) products as bf16 x 3 (upack2 = that stream)
    FDN_REQUIRE(!split || hm == 4, "conv64 (2-D winograd): the bf16 x 3 products exist for F(4,3) along H only");
    FDN_REQUIRE(fdn_conv64_wino2d_ok(ebd, ebh, ebw, ID, IH, IW, hm), "conv64 (2-D winograd, F(%d,3) along H): box %dx%dx%d of a %dx%dx%d grid is not supported",
                hm, ebd, ebh, ebw, ID, IH, IW);
    FDN_REQUIRE(!fout || (zero_mode && off == -1 && obd == 1 && obh == 1 && obw == 1), "conv64 (2-D winograd): the fused fold belongs to the inner box of a padded dgrad");
    FDN_REQUIRE(!(ymask || fmask) || (hm == 4 && !split && (fmask ? fout != nullptr : fout == nullptr)),
                "conv64 (2-D winograd): sign masks belong to the F(4,3) x F(4,3) fp32-MFMA kernels (forward writes, fused dgrad reads)");
    Wino2Args a;
    a.x = x; a.up = upack2; a.bias = bias; a.res = residual; a.y = y; a.fskip = fskip; a.fy = fy; a.fout = fout;
    a.ymask = ymask; a.fmask = fmask;
    a.x1 = a.x2 = nullptr; a.wd1 = a.wd2 = 0; a.nsrc = 1; a.wspan = 0;
    if (extra) {
        FDN_REQUIRE(fout && !split && extra->nsrc >= 1 && extra->nsrc <= 3, "conv64 (2-D winograd): further sources belong to a fused dgrad (fp32-MFMA), 1..3 in all");
        a.x1 = extra->x1; a.x2 = extra->x2; a.wd1 = extra->wd1; a.wd2 = extra->wd2; a.nsrc = extra->nsrc;
        a.wspan = extra->nsrc > 2 ? extra->wd2 : (extra->nsrc > 1 ? extra->wd1 : 0);
    }
    a.N = N; a.ID = ID; a.IH = IH; a.IW = IW; a.OD = OD; a.OH = OH; a.OW = OW;
    a.off = off; a.zero_mode = zero_mode; a.act = act; a.alpha = alpha; a.dbg = fdn_conv64_wino2d_dbg;
    a.hm = hm; a.split = split ? 1 : 0;
    a.obd = obd; a.obh = obh; a.obw = obw; a.ebd = ebd; a.ebh = ebh; a.ebw = ebw;
    const int ech = ebh / hm, ecw = ebw / 4;
    Wino2Plan pl = wino2d_plan(N, ebd, ech, ecw, 2);
    // half-size tiles (F(4,3) along H only) where the full-size ones leave CUs with one workgroup or none: kW2HalfBelow full tiles
    int mb = 2;
    if (hm == 4 && !split) {
        const long long full = (long long)N * ((ebd + pl.td - 1) / pl.td) * ((ech + pl.ch - 1) / pl.ch) * ((ecw + pl.cw - 1) / pl.cw);
        if (full < kW2HalfBelow) mb = 1;
        if (fdn_conv64_wino2d_mb) mb = fdn_conv64_wino2d_mb;
        else if (fdn_conv64_wino2d_tile)                // (test build: a forced tile decides by its own size)
            mb = (fdn_conv64_wino2d_tile & 255) * ((fdn_conv64_wino2d_tile >> 8) & 255) * ((fdn_conv64_wino2d_tile >> 16) & 255) > 16 ? 2 : 1;
        if (mb == 1) pl = wino2d_plan(N, ebd, ech, ecw, 1);
    }
    if (fdn_conv64_wino2d_tile) {
        pl.td = fdn_conv64_wino2d_tile & 255; pl.ch = (fdn_conv64_wino2d_tile >> 8) & 255; pl.cw = (fdn_conv64_wino2d_tile >> 16) & 255;
        FDN_REQUIRE(pl.td >= 1 && pl.ch >= 1 && pl.cw >= 1 && pl.td * pl.ch * pl.cw <= 16 * mb && (pl.td + 2) * pl.ch * pl.cw <= (mb == 2 ? kW2Rows : kW2RowsH),
                    "conv64 (2-D winograd): forced tile %dx%dx%d does not fit", pl.td, pl.ch, pl.cw);
    }
    a.mb = mb;
    a.td = pl.td; a.ch = pl.ch; a.cw = pl.cw;
    a.ntd = (ebd + pl.td - 1) / pl.td; a.nth = (ech + pl.ch - 1) / pl.ch; a.ntw = (ecw + pl.cw - 1) / pl.cw;
    a.cpp = pl.ch * pl.cw; a.rows = (pl.td + 2) * a.cpp; a.items = a.rows * 16;
    a.mg_cpp = fdn_magic20(a.cpp); a.mg_cw = fdn_magic20(pl.cw);
    fdn_magic40(a.ntd * a.nth * a.ntw, &a.mg_tpn_hi, &a.mg_tpn_lo);
    fdn_magic40(a.nth * a.ntw, &a.mg_thw_hi, &a.mg_thw_lo);
    fdn_magic40(a.ntw, &a.mg_ntw_hi, &a.mg_ntw_lo);
    const long long blocks = (long long)N * a.ntd * a.nth * a.ntw;
    FDN_REQUIRE(blocks < (1ll << 31), "conv64 (2-D winograd): too many tiles");
    memcpy(out->args, &a, sizeof(a));
    out->blocks = (int)blocks;
    out->lds = split ? W2Geo<2, true>::lds : (mb == 2 ? kW2Lds : W2Geo<1>::lds);
    return FDN_OK;
}

int fdn_conv64_wino2d_launch(const float* x, const float* upack2, const float* bias, const float* residual, float* y,
                             const float* fskip, const float* fy, float* fout, int N, int ID, int IH, int IW, int OD, int OH,
                             int OW, int obd, int obh, int obw, int ebd, int ebh, int ebw, int off, int zero_mode, int act,
                             float alpha, int hm_arg, hipStream_t s, uint16_t* ymask, const uint16_t* fmask) {
    const int hm = hm_arg & 7;
    FdnWino2dPrepared pr;
    if (int rc = fdn_conv64_wino2d_prepare(x, upack2, bias, residual, y, fskip, fy, fout, N, ID, IH, IW, OD, OH, OW, obd, obh, obw, ebd,
                                           ebh, ebw, off, zero_mode, act, alpha, hm_arg, &pr, ymask, fmask))
        return rc;
    Wino2Args a;
    memcpy(&a, pr.args, sizeof(a));
    const long long blocks = pr.blocks;
    if (a.split) {
        // bf16 x 3 products: one persistent producer / consumer workgroup per CU walks the tiles
        const void* fnp = fout ? (const void*)conv64_wino2d_pc_kernel<true> : (const void*)conv64_wino2d_pc_kernel<false>;
        bool sync_variant = false;
#ifdef FDN_TEST_HOOKS
        sync_variant = fdn_conv64_wino2d_variant == 10;           // test build: the barrier-synchronous bf16 x 3 kernel (SPLIT body), for A/B
        if (sync_variant) fnp = fout ? (const void*)conv64_wino2d_kernel<true, 4, 2, true> : (const void*)conv64_wino2d_kernel<false, 4, 2, true>;
#endif
        const int ldsp = sync_variant ? W2Geo<2, true>::lds : kPcLds;
        if (int rc = fdn_func_max_lds(fnp, ldsp, "conv64_wino2d_pc")) return rc;
        const int cus = device_cus();
        const unsigned grid = sync_variant ? (unsigned)blocks : (unsigned)(blocks < cus ? blocks : cus);
        void* kargs[] = {(void*)&a};
        const hipError_t e = hipLaunchKernel(fnp, dim3(grid), dim3(sync_variant ? 256 : kPcThreads), kargs, ldsp, s);
        if (e != hipSuccess) {
            fdn_set_error("conv64_wino2d_pc_kernel: launch failed: %s", hipGetErrorString(e));
            return FDN_ERR_HIP;
        }
        return FDN_OK;
    }
    FDN_REQUIRE(!a.fmask, "conv64 (2-D winograd): the mask-reading fused dgrad is the one-launch form (inner box + shell)");
    FDN_REQUIRE(a.nsrc == 1, "conv64 (2-D winograd): the multi-source fused dgrad is the one-launch form (inner box + shell)");
    const void* fn = a.ymask ? (a.mb == 1 ? (const void*)conv64_wino2d_kernel<false, 4, 1, false, true> : (const void*)conv64_wino2d_kernel<false, 4, 2, false, true>)
                   : fout ? (hm == 4 ? (a.mb == 1 ? (const void*)conv64_wino2d_kernel<true, 4, 1> : (const void*)conv64_wino2d_kernel<true, 4>) : (const void*)conv64_wino2d_kernel<true, 2>)
                          : (hm == 4 ? (a.mb == 1 ? (const void*)conv64_wino2d_kernel<false, 4, 1> : (const void*)conv64_wino2d_kernel<false, 4>) : (const void*)conv64_wino2d_kernel<false, 2>);
    int lds = pr.lds;
#ifdef FDN_TEST_HOOKS
    if (fdn_conv64_wino2d_variant && !fout && hm == 2) {
        switch (fdn_conv64_wino2d_variant) {
            case 1: fn = (const void*)conv64_wino2d_occ1_kernel<6, 3, 1>; break;
            case 2: fn = (const void*)conv64_wino2d_occ1_kernel<12, 6, 1>; break;
            case 3: fn = (const void*)conv64_wino2d_occ1_kernel<12, 6, 3>; break;
            case 4: fn = (const void*)conv64_wino2d_occ1_kernel<8, 4, 2>; break;
            case 5: fn = (const void*)conv64_wino2d_occ1_kernel<24, 6, 3>; break;
            default: FDN_REQUIRE(false, "conv64 (2-D winograd): unknown variant %d", fdn_conv64_wino2d_variant);
        }
        lds = kW2LdsOcc1;
    }
#endif
    if (int rc = fdn_func_max_lds(fn, lds, "conv64_wino2d")) return rc;
    void* kargs[] = {(void*)&a};
    const hipError_t e = hipLaunchKernel(fn, dim3((unsigned)blocks), dim3(256), kargs, lds, s);
    if (e != hipSuccess) {
        fdn_set_error("conv64_wino2d_kernel: launch failed: %s", hipGetErrorString(e));
        return FDN_ERR_HIP;
    }
    return FDN_OK;
}

int fdn_pack_conv64_wino2d_launch(const float* w, float* uf, float* ud, hipStream_t s) {
    hipLaunchKernelGGL(pack_conv64_wino2d_kernel, dim3((72 * 64 * 64 + 255) / 256), dim3(256), 0, s, w, uf, ud);
    FDN_CHECK_LAUNCH("pack_conv64_wino2d_kernel");
    return FDN_OK;
}

#ifdef FDN_TEST_HOOKS
extern "C" int fdn_debug_set_conv64_wino2d_dbg(int bits) { fdn_conv64_wino2d_dbg = bits; return FDN_OK; }
extern "C" int fdn_debug_set_conv64_wino2d_tile(int packed) { fdn_conv64_wino2d_tile = packed; return FDN_OK; }
extern "C" int fdn_debug_set_conv64_wino2d_variant(int v) { fdn_conv64_wino2d_variant = v; return FDN_OK; }
extern "C" int fdn_debug_set_conv64_wino2d_mb(int mb) { fdn_conv64_wino2d_mb = mb; return FDN_OK; }
#endif

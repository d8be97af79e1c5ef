// conv3d 3x3x3, 64 -> 64 channels, fp32, NDHWC: Winograd F(4,3) along W + direct taps along D,H, on v_mfma_f32_32x32x2_f32.
//
// Same contract as conv64_mfma.hip (tf.pad(SYMMETRIC) + Conv3D + bias + activation + residual of
// src/Network/SR4DFlowNet.py:93-120 in clamp mode; Conv3DBackpropInputV2 of the same layers in zero mode with the
// interior MirrorPadGrad fused into the epilogue), but with HALF the multiplies: on MI355X the fp32 matrix rate equals
// the fp32 vector rate (157.3 TFLOP/s both), so the only way past the fp32 MFMA roofline is to execute fewer FLOPs.
//
// A run of 4 consecutive output voxels along W (a "group") is computed from 6 input voxels x[4p-1 .. 4p+4] as
//     Y = A^T [ (G g) .* (B^T x) ]                      (Lavin & Gray, F(4,3); interpolation points 0, +-1, +-2, inf)
// per (depth tap a, height tap b, cin): 6 multiplies instead of 12.  In implicit-GEMM terms the 27-tap K loop of the direct
// kernel becomes 9 (a,b) taps x 6 Winograd coordinates xi, each a [groups x 64 cin] x [64 cin x 64 cout] GEMM into its OWN
// accumulator M_xi; the output transform A^T runs once, in the epilogue.  MFMA work per voxel: 9*6/4 = 13.5 tap-equivalents
// instead of 27.  fp32 error: ~3x the direct kernel's (5e-7 vs 1.7e-7 relative, tools/wino_numerics.py) -- three orders of
// magnitude inside the 1e-3 parity tolerance.
//
// Work decomposition:
//   * M = groups.  One workgroup (4 waves) = a box tile of td x th x tg groups (<= 64 groups = 256 voxels) x 64 cout;
//     wave (wm, wn) owns 32 groups x 32 cout x 6 xi = 6 accumulator tiles of 32x32 (96 VGPRs).
//   * Input transform while staging: for every (halo line, group, 16-B channel chunk) one thread loads the 6 input chunks
//     (boundary rule applied: edge clamp == SYMMETRIC p=1, or zero for dgrad, through the buffer range check), forms the 6
//     transformed chunks V_xi = B^T x (14 VALU per float) and writes them to LDS as 6 planes [xi][line][group] of rows with
//     64/CS channels (+16-B pad: conflict-free ds_read_b128, see conv64_mfma.hip).  CS = 4 slices of 16 cin: 48 KB of LDS for an
//     8x8x1 tile, 2 workgroups per CU; while one transforms, the other keeps the matrix pipe busy.
//   * Weights: the packed stream U = G g per (a,b) tap, [cin/32][tap*6+xi][k-group][lane-half][cout row][4], read straight from
//     L1/L2 (885 KB per layer and direction, shared by every workgroup), cout rows permuted so a lane's 16 accumulator
//     registers are 16 consecutive channels (as in the direct kernel).
//   * K loop: one step = one k-group (8 cin) of one (tap, xi): 1 ds_read_b128 + 1 buffer_load_b128 feed 4 MFMAs.  Fragments
//     live in a ring of 4 slots; the slot consumed by step s-1 is refilled with the operands of step s+3 right after the first
//     MFMA of step s (scalar address arithmetic, one v_add per LDS read) -- the same in-stream pipeline as the direct kernel.
//   * Epilogue: Y = A^T M per lane (12 VALU per output element quartet), then bias / residual / activation, or the fused
//     MirrorPadGrad + skip + act' of the dgrad mode, and 16-B stores: a lane owns 4 voxels x 16 consecutive channels.
#include "fdn_common.h"
#include "conv64_pack.h"
#include <string.h>

#include "conv64_wino_kernel.h"
#include "conv64_wino2d_kernel.h"

namespace {

FDN_HOOK_VAR(int, fdn_conv64_wino_dbg, 0);
FDN_HOOK_VAR(int, fdn_conv64_wino_tile, 0);            // test build: force the main region's tile, td | th << 8 | tg << 16 (0 = planner)

template <int CS, bool GEN>
__global__ __launch_bounds__(256, 2) void conv64_wino_kernel(WinoArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    conv64_wino_body<CS, GEN>(p, (int)blockIdx.x, smem);
}

// Fused dgrad in ONE launch (round 4): workgroups [0, n2d) run the 2-D Winograd body on the inner box (fused-fold epilogue), the rest
// the 1-D body on the shell faces.  The faces are a few hundred short tiles; as a launch of their own they cost a launch gap and an
// almost empty chip (0.069 ms at (8,48^3), 0.027 ms at (8,24^3)), here they are dispatched last and fill the inner launch's tail
// (its last round of workgroup slots is a quarter empty at 48^3; at 24^3 the 432 inner tiles leave 80 of the 512 slots free).
template <int HM, int MB = 2, bool MASK = false>
__global__ __launch_bounds__(256, 2) void conv64_wino2d_shell_kernel(Wino2Args p2, WinoArgs p1, int n2d) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < n2d) conv64_wino2d_body<true, HM, kW2RDB, kW2RDA, kW2Dep, MB, false, MASK>(p2, (int)blockIdx.x, smem);
    else conv64_wino_body<kWinoCS, true>(p1, (int)blockIdx.x - n2d, smem);
}

__global__ void pack_conv64_wino_kernel(const float* __restrict__ w, float* __restrict__ uf, float* __restrict__ ud) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < 54 * 64 * 64) fdn_pack_wino_one(w, uf, ud, idx);
}

struct WinoPlan { int td, th, tg; double cost; };

// tile choice: every tile costs the MFMA time of 64 groups (scaled by the region's share of the 9 (kd,kh) taps) whatever its
// fill, plus the transform work of its halo lines and a fixed prologue / epilogue; the launch ends with the busiest CU
// (2 co-resident workgroups per CU share the matrix pipe, so work per CU = its tiles).
// tail = a secondary region of a multi-region launch (the shell faces of a fused dgrad): its tiles fill the slots the main region
// leaves free, so what counts is their total work (tiles x per-tile time), not rounds over a chip of their own -- without this the
// faces were cut into 4x more tiles than needed (13- and 25-group tiles that still pay the K loop of 64 groups).
WinoPlan wino_plan(int N, const FdnWinoBox& bx, bool tail) {
    const int ebg = bx.ew / 4, da = bx.ta1 - bx.ta0, db = bx.tb1 - bx.tb0;
    const double tapfrac = (da + 1) * (db + 1) / 9.0;
    WinoPlan best{1, 1, 1, 1e30};
    for (int td = 1; td <= bx.ed && td <= 64; ++td)
        for (int th = 1; th <= bx.eh && td * th <= 64; ++th)
            for (int tg = 1; tg <= ebg && td * th * tg <= 64; ++tg) {
                const int ltg = (td + da) * (th + db) * tg;
                if (ltg > kWinoMaxLtg || ltg * (64 / kWinoCS / 4) > kWinoUA * 256) continue;
                const double tiles = (double)N * ((bx.ed + td - 1) / td) * ((bx.eh + th - 1) / th) * ((ebg + tg - 1) / tg);
                const double per_tile = 64.0 * tapfrac * (bx.wface ? 1.0 / 3 : 1.0) + (tail ? 0.04 : 0.12) * ltg + 4.0;    // tail tiles: 0.04 / 0.12 / 0.30 measured 0.859 / 0.863 / 0.865 ms at (8,48^3)
                const double rounds = 0.9 * (double)((long long)((tiles + 255) / 256)) + 0.1 * tiles / 256.0;
                const double c = tail ? tiles * per_tile : rounds * per_tile;
                if (c < best.cost) best = {td, th, tg, c};
            }
    return best;
}

}  // namespace

// Is the Winograd kernel applicable to this output box?  (W extent a multiple of 4; everything else falls to the direct kernel.)
bool fdn_conv64_wino_ok(int ebd, int ebh, int ebw) { return ebd > 0 && ebh > 0 && ebw >= 4 && (ebw & 3) == 0; }

int fdn_conv64_wino_launch_boxes(const float* x, const float* upack, const float* bias, const float* residual, float* y,
                                 const float* fskip, const float* fy, float* fout, int N, int ID, int IH, int IW, int OD, int OH,
                                 int OW, const FdnWinoBox* boxes, int nbox, int off, int zero_mode, int act, float alpha,
                                 hipStream_t s, const FdnWino2dPrepared* inner, const FdnExtraSrc* extra) {
    FDN_REQUIRE((long long)ID * IH * IW < (1ll << 24), "conv64 (winograd): a sample of %dx%dx%d voxels exceeds the 32-bit row addressing", ID, IH, IW);
    FDN_REQUIRE(nbox >= 1 && nbox <= 6, "conv64 (winograd): %d regions", nbox);
    constexpr int CS = kWinoCS, LROW = 256 / CS + 16, CH = 256 / CS / 16;
    WinoArgs a;
    a.x = x; a.up = upack; a.bias = bias; a.res = residual; a.y = y; a.fskip = fskip; a.fy = fy; a.fout = fout;
    a.x1 = a.x2 = nullptr; a.wd1 = a.wd2 = 0; a.nsrc = 1; a.wspan = 0;
    if (extra) {
        FDN_REQUIRE(inner && fout && extra->nsrc >= 1 && extra->nsrc <= 3, "conv64 (winograd): further sources belong to the one-launch fused dgrad, 1..3 in all");
        a.x1 = extra->x1; a.x2 = extra->x2; a.wd1 = extra->wd1; a.wd2 = extra->wd2; a.nsrc = extra->nsrc;
        a.wspan = extra->nsrc > 2 ? extra->wd2 : (extra->nsrc > 1 ? extra->wd1 : 0);
    }
    a.N = N; a.ID = ID; a.IH = IH; a.IW = IW; a.OD = OD; a.OH = OH; a.OW = OW;
    a.off = off; a.zero_mode = zero_mode; a.act = act; a.alpha = alpha; a.dbg = fdn_conv64_wino_dbg;
    a.nreg = 0;
    long long blocks = 0;
    int max_ltg = 0;
    // a launch that starts with a shell face carries no main region (round 4: the inner box of a fused dgrad runs on the 2-D kernel):
    // all of its regions share the chip, so all are planned by total work
    const bool shell_only = boxes[0].wface || boxes[0].ta0 != 0 || boxes[0].ta1 != 2 || boxes[0].tb0 != 0 || boxes[0].tb1 != 2;
    for (int i = 0; i < nbox; ++i) {
        const FdnWinoBox& bx = boxes[i];
        if (bx.ed <= 0 || bx.eh <= 0 || bx.ew <= 0) continue;
        FDN_REQUIRE(fdn_conv64_wino_ok(bx.ed, bx.eh, bx.ew), "conv64 (winograd): W extent %d is not a multiple of 4", bx.ew);
        FDN_REQUIRE(!bx.wface || (zero_mode && fout && bx.ew == 4 && bx.ow == 0 && bx.ta0 == 0 && bx.ta1 == 2 && bx.tb0 == 0 && bx.tb1 == 2),
                    "conv64 (winograd): a w-face region belongs to a fused dgrad launch");
        WinoPlan pl = wino_plan(N, bx, bx.wface != 0);
        if (fdn_conv64_wino_tile && a.nreg == 0) {
            pl.td = fdn_conv64_wino_tile & 255; pl.th = (fdn_conv64_wino_tile >> 8) & 255; pl.tg = (fdn_conv64_wino_tile >> 16) & 255;
        }
        // a secondary region too small to fill the chip on its own (< 1024 tiles = two rounds of workgroup slots): plan it by total work instead (see wino_plan)
        if ((a.nreg > 0 || shell_only) && !(fdn_conv64_wino_dbg & 256) &&
            (long long)N * ((bx.ed + pl.td - 1) / pl.td) * ((bx.eh + pl.th - 1) / pl.th) * ((bx.ew / 4 + pl.tg - 1) / pl.tg) < 1024)
            pl = wino_plan(N, bx, true);
        WinoRegion& r = a.reg[a.nreg++];
        r.first_block = (int)blocks;
        r.wface = bx.wface;
        r.obd = bx.od; r.obh = bx.oh; r.obw = bx.ow; r.ebd = bx.ed; r.ebh = bx.eh; r.ebw = bx.ew;
        r.ta0 = bx.ta0; r.ta1 = bx.ta1; r.tb0 = bx.tb0; r.tb1 = bx.tb1;
        r.td = pl.td; r.th = pl.th; r.tg = pl.tg;
        const int ebg = bx.ew / 4;
        r.ntd = (bx.ed + pl.td - 1) / pl.td; r.nth = (bx.eh + pl.th - 1) / pl.th; r.ntg = (ebg + pl.tg - 1) / pl.tg;
        r.hh = pl.th + (bx.tb1 - bx.tb0); r.lines = (pl.td + (bx.ta1 - bx.ta0)) * r.hh; r.ltg = r.lines * pl.tg; r.items = r.ltg * CH;
        r.mg_tg = fdn_magic20(pl.tg); r.mg_thtg = fdn_magic20(pl.th * pl.tg);
        r.mg_itg = fdn_magic20(pl.tg); r.mg_ihh = fdn_magic20(r.hh);
        fdn_magic40(r.ntd * r.nth * r.ntg, &r.mg_tpn_hi, &r.mg_tpn_lo);
        fdn_magic40(r.nth * r.ntg, &r.mg_thg_hi, &r.mg_thg_lo);
        fdn_magic40(r.ntg, &r.mg_ntg_hi, &r.mg_ntg_lo);
        blocks += (long long)N * r.ntd * r.nth * r.ntg;
        if (r.ltg > max_ltg) max_ltg = r.ltg;
    }
    if (a.nreg == 0) return FDN_OK;
    FDN_REQUIRE(blocks < (1ll << 31), "conv64 (winograd): too many tiles");
    size_t lds = (size_t)6 * (kWinoMaxLtg * LROW + 64) + 192 * 4;      // fixed plane stride (see the kernel); max_ltg <= kWinoMaxLtg by the planner
    (void)max_ltg;
    if (fdn_conv64_wino_dbg & 64) lds = 82 * 1024;             // ablation: only ONE workgroup fits a CU
    const int lds_max = 84 * 1024;
    if (int rc = fdn_func_max_lds((const void*)conv64_wino_kernel<CS, true>, lds_max, "conv64_wino")) return rc;
    if (int rc = fdn_func_max_lds((const void*)conv64_wino_kernel<CS, false>, lds_max, "conv64_wino")) return rc;
    if (inner) {
        // the fused dgrad as ONE launch: the prepared 2-D launch of the inner box in front, these (shell) regions behind it
        static_assert(sizeof(Wino2Args) <= sizeof(FdnWino2dPrepared::args), "FdnWino2dPrepared::args too small");
        FDN_REQUIRE(inner->blocks > 0 && blocks + inner->blocks < (1ll << 31), "conv64 (winograd): too many tiles");
        Wino2Args a2;
        memcpy(&a2, inner->args, sizeof(a2));
        if ((size_t)inner->lds > lds) lds = (size_t)inner->lds;
        FDN_REQUIRE(!a2.split, "conv64 (winograd): the bf16 x 3 inner box is a launch of its own");
        const void* fn = a2.fmask ? (a2.mb == 1 ? (const void*)conv64_wino2d_shell_kernel<4, 1, true> : (const void*)conv64_wino2d_shell_kernel<4, 2, true>)
                       : a2.hm == 4 ? (a2.mb == 1 ? (const void*)conv64_wino2d_shell_kernel<4, 1> : (const void*)conv64_wino2d_shell_kernel<4>)
                                    : (const void*)conv64_wino2d_shell_kernel<2>;
        if (int rc = fdn_func_max_lds(fn, lds_max, "conv64_wino2d_shell")) return rc;
        int n2d = inner->blocks;
        void* kargs[] = {(void*)&a2, (void*)&a, (void*)&n2d};
        const hipError_t e = hipLaunchKernel(fn, dim3((unsigned)(blocks + inner->blocks)), dim3(256), kargs, lds, s);
        if (e != hipSuccess) {
            fdn_set_error("conv64_wino2d_shell_kernel: launch failed: %s", hipGetErrorString(e));
            return FDN_ERR_HIP;
        }
        return FDN_OK;
    }
    const WinoRegion& r0 = a.reg[0];
    const bool simple = a.nreg == 1 && r0.ta0 == 0 && r0.ta1 == 2 && r0.tb0 == 0 && r0.tb1 == 2;
    if (simple) hipLaunchKernelGGL((conv64_wino_kernel<CS, false>), dim3((unsigned)blocks), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((conv64_wino_kernel<CS, true>), dim3((unsigned)blocks), dim3(256), lds, s, a);
    FDN_CHECK_LAUNCH("conv64_wino_kernel");
    return FDN_OK;
}

int fdn_conv64_wino_launch(const float* x, const float* upack, const float* bias, const float* residual, float* y,
                           const float* fskip, const float* fy, float* fout, int N, int ID, int IH, int IW, int OD, int OH,
                           int OW, int obd, int obh, int obw, int ebd, int ebh, int ebw, int off, int zero_mode, int act,
                           float alpha, hipStream_t s) {
    const FdnWinoBox bx{obd, obh, obw, ebd, ebh, ebw, 0, 2, 0, 2};
    return fdn_conv64_wino_launch_boxes(x, upack, bias, residual, y, fskip, fy, fout, N, ID, IH, IW, OD, OH, OW, &bx, 1, off,
                                        zero_mode, act, alpha, s);
}

#ifdef FDN_TEST_HOOKS
extern "C" int fdn_debug_set_conv64_wino_dbg(int bits) { fdn_conv64_wino_dbg = bits; return FDN_OK; }
extern "C" int fdn_debug_set_conv64_wino_tile(int packed) { fdn_conv64_wino_tile = packed; return FDN_OK; }
#endif

int fdn_pack_conv64_wino_launch(const float* w, float* uf, float* ud, hipStream_t s) {
    hipLaunchKernelGGL(pack_conv64_wino_kernel, dim3((54 * 64 * 64 + 255) / 256), dim3(256), 0, s, w, uf, ud);
    FDN_CHECK_LAUNCH("pack_conv64_wino_kernel");
    return FDN_OK;
}

// Device side of conv64_wino.hip (1-D Winograd F(4,3) along W): argument structs, constants and the kernel body.  Kept in a header
// because two launches run it: conv64_wino_kernel (conv64_wino.hip) and the tail workgroups of conv64_wino2d_shell_kernel, the
// fused-dgrad launch whose first workgroups run the 2-D body of conv64_wino2d_kernel.h.  See conv64_wino.hip for the design notes.
#pragma once
#include "fdn_common.h"
#include "conv64_pack.h"

namespace {

// One launch covers up to 5 REGIONS of the output grid: a region is an output box (W extent a multiple of 4) with the (kd, kh)
// tap ranges that can be non-zero for it, tiled with its own tile shape; workgroups [first_block, next region's) belong to it.
//   forward     : 1 region, all taps.
//   fused dgrad : the inner D x H x W box of the padded grid with all taps + the two d faces (1 depth tap) and the two h faces
//                 (1 height tap) of the shell, each W-inner: they cost a third of an inner tile, are dispatched last and fill
//                 the launch's tail.
//   w faces     : (round 3) the two w faces of the shell have a single W tap each, and in the Winograd domain that tap is ONE
//                 coordinate: the left face (padded w = 0) needs g[2] = U_5, the right face (w = OW-1) needs g[0] = 4 U_0.  A
//                 w-face region therefore reuses the staging code unchanged -- a "group" is a (d,h) position, its six loads are
//                 x[IW-1], four out-of-range rows, x[0], so that B^T x gives V_0 = 4 x[IW-1], V_5 = x[0], V_1..4 = 0 -- runs
//                 the K loop over xi in {0, 5} only (a third of a tile's MFMAs), and stores M_5 / M_0 without an output
//                 transform.  It replaces the separate direct-kernel launch per dgrad (24 us avg, 0.72 ms per cfg2 step).
struct WinoRegion {
    int first_block;
    int wface;                            // 1 = the w-face pair (see above)
    int obd, obh, obw, ebd, ebh, ebw;
    int ta0, ta1, tb0, tb1;
    int td, th, tg, ntd, nth, ntg;        // tile in (d, h, groups) and tile counts
    int hh, lines, ltg, items;            // th + (tb1-tb0), (td + (ta1-ta0))*hh, lines*tg, lines*tg*CH
    unsigned mg_tg, mg_thtg, mg_itg, mg_ihh;
    unsigned mg_tpn_hi, mg_tpn_lo, mg_thg_hi, mg_thg_lo, mg_ntg_hi, mg_ntg_lo;
};

struct WinoArgs {
    const float* x;
    const float* up;        // Winograd-domain operand stream (fdn_pack_conv64_weights, second part of the pack)
    const float* bias;
    const float* res;
    float* y;
    const float* fskip;     // fused fold (dgrad mode): see conv64_args.h
    const float* fy;
    float* fout;
    // multi-source fused dgrad (conv64_wino2d_kernel.h, Wino2Args): the slice loop runs over CS x nsrc slices
    const float* x1;
    const float* x2;
    int wd1, wd2, nsrc, wspan;
    int N, ID, IH, IW, OD, OH, OW;
    int off, zero_mode, act;
    float alpha;
    int dbg;                // ablation bits (test build only): 1 = weight stream stride 0, 4 = no staging, 8 = no epilogue,
                            // 16 = no transform arithmetic, 32 = epilogue arithmetic without the stores, 128 = no XCD remap, 256 = plan the shell faces like stand-alone launches
    int nreg;
    WinoRegion reg[6];
};


constexpr int kWinoCS = 4;                 // cin slices
constexpr int kWinoMaxLtg = 160;           // LDS: 6 planes x (160 rows x 80 B + 64) + tables = 76.1 KB -> 2 workgroups per CU
constexpr int kWinoRDB = 6;                // weight-fragment ring depth (prefetch distance 5 K steps)
constexpr int kWinoRDA = 3;                // voxel-fragment ring depth (LDS, distance 2)
constexpr int kWinoUA = 3;                 // transform items per thread (<= 768 items = 192 (line, group) pairs x 4 chunks)

// GEN = false: a single region with all 9 (kd,kh) taps (every forward launch) -- tap ranges are compile-time constants.
// (a __device__ body + thin __global__ wrappers: conv64_wino.hip also runs it as the tail of the 2-D kernel's fused-dgrad launch)
template <int CS, bool GEN>
__device__ __forceinline__ void conv64_wino_body(const WinoArgs& p, const int block_id, char* const smem) {
    constexpr int ROWB = 256 / CS, LROW = ROWB + 16, CH = ROWB / 16, KG = 8 / CS;
    constexpr int SPT = 6 * KG;            // K steps per (a,b) tap
    // fragment rings: the weight fragments come from L2 (a wave that has its SIMD to itself -- its co-resident partner staging or
    // storing -- runs the K loop at twice the shared rate, so the prefetch distance must cover the L2 latency at THAT rate), the
    // voxel fragments from LDS
    constexpr int RDB = kWinoRDB, RDA = kWinoRDA;
    constexpr int UA = kWinoUA;
    static_assert(SPT % RDB == 0 && SPT % RDA == 0, "ring slots must be compile-time");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int kh = lane >> 5;
    const int wave_m = wave & 1;
    const int wave_n = wave >> 1;
    // ---- which region, which tile (scalar multiply-shift divisions, host-made magics) ----
    int ri = 0;
    if (GEN) {
        while (ri + 1 < p.nreg && block_id >= p.reg[ri + 1].first_block) ++ri;
        ri = __builtin_amdgcn_readfirstlane(ri);
    }
    const WinoRegion R = p.reg[ri];
    const int ta0 = GEN ? R.ta0 : 0, ta1 = GEN ? R.ta1 : 2, tb0 = GEN ? R.tb0 : 0, tb1 = GEN ? R.tb1 : 2;
    // bytes per xi plane: a compile-time stride, so the plane / k-group part of every LDS address is an instruction immediate (2 instead
    // of 7 VALU adds per tap and wave: forward launch at (8,48^3) 0.771 -> 0.756 ms).  The + 64 matters: with a stride that is a multiple
    // of the 256-B bank row the six planes alias and the fused dgrad launch runs 2.7 % slower (0.908 vs 0.884 ms); 32 / 128 / 176 are
    // within 1 % of 64.
    constexpr int planeb = kWinoMaxLtg * LROW + 64;
    int* mtab = (int*)(smem + 6 * planeb);                 // [0,64): output index of the group's first voxel; [64,128): fused
                                                           // index (interior d,h) or -1; [128,192): iw of the first voxel
    const int tiles_per_n = R.ntd * R.nth * R.ntg;
    int b = block_id - R.first_block;
    if (!(FDN_DBG_BITS(p) & 128)) {
        // XCD-aware tile order: workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2); ids with the same residue
        // take one contiguous eighth of the region's tile list, so the tiles that share halo lines (w and h neighbours are a few ids
        // apart) run on the same XCD at about the same time and part of the 2.3x halo over-read is served by that L2 instead of HBM
        // (forward launch at (8,48^3): HBM reads 779 -> 379 MB against 226 MB of input, 0.778 -> 0.771 ms; starting every XCD at a
        // different phase of its run: no change).
        const int T = p.N * tiles_per_n, q = T >> 3, r = T & 7, xcd = b & 7;
        b = xcd * q + min(xcd, r) + (b >> 3);
    }
    const int n = fdn_udiv40(b, R.mg_tpn_hi, R.mg_tpn_lo);
    b -= n * tiles_per_n;
    const int tdi = fdn_udiv40(b, R.mg_thg_hi, R.mg_thg_lo);
    b -= tdi * (R.nth * R.ntg);
    const int thi = fdn_udiv40(b, R.mg_ntg_hi, R.mg_ntg_lo);
    const int p0d = R.obd + tdi * R.td, p0h = R.obh + thi * R.th, p0w = R.obw + (b - thi * R.ntg) * R.tg * 4;
    const int ng = R.td * R.th * R.tg;
    const int thtg = R.th * R.tg;

    if (tid < 64) {
        int g = -1, gf = -1, iw0 = 0;
        if (tid < ng) {
            const int md = fdn_div20(tid, R.mg_thtg);
            const int r2 = tid - md * thtg;
            const int mh = fdn_div20(r2, R.mg_tg);
            const int pd = p0d + md, ph = p0h + mh, pw = p0w + 4 * (r2 - mh * R.tg);
            if (pd < R.obd + R.ebd && ph < R.obh + R.ebh && pw < R.obw + R.ebw) {
                g = ((n * p.OD + pd) * p.OH + ph) * p.OW + pw;
                if (p.fout && !(GEN && R.wface)) {
                    const int id = pd - 1, ih = ph - 1;
                    iw0 = pw - 1;
                    if (id >= 1 && id <= p.ID - 2 && ih >= 1 && ih <= p.IH - 2) gf = ((n * p.ID + id) * p.IH + ih) * p.IW + iw0;
                }
            }
        }
        mtab[tid] = g; mtab[64 + tid] = gf; mtab[128 + tid] = iw0;
    }

    // ---- this lane's A row: (line of its group at tap (0,0)) * tg + group-in-line, plane xi = 0 ----
    int abase;
    {
        int m = wave_m * 32 + li;
        m = m < ng ? m : ng - 1;
        const int md = fdn_div20(m, R.mg_thtg);
        const int r2 = m - md * thtg;
        const int mh = fdn_div20(r2, R.mg_tg);
        abase = ((md * R.hh + mh) * R.tg + (r2 - mh * R.tg)) * LROW + kh * 16;
    }
    const int cofs = wave_n * 32 + kh * 16;                // first of this lane's 16 consecutive output channels

    f32x16 acc[6];
#pragma unroll
    for (int xi = 0; xi < 6; ++xi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[xi][r] = 0.f;

    // ---- transform plan, once per tile: for each of this thread's (line, group, chunk) items the byte offsets (from the
    // sample's first voxel) of the 6 input rows with the boundary rule applied (0xffffffff = reads zero) and the LDS offset ----
    // staged box origin in input coordinates: output p reads input p + tap - 1 + off, the first staged (kd,kh) tap is (ta0,tb0)
    const int q0d = p0d - 1 + p.off + ta0, q0h = p0h - 1 + p.off + tb0, q0w = p0w - 1 + p.off;
    unsigned soff[UA][6];
    int vrow[UA];
#pragma unroll
    for (int u = 0; u < UA; ++u) {
        const int i = u * 256 + tid;
        const bool oki = i < R.items;
        const int chunk = i & (CH - 1);
        const int r = i / CH;
        const int line = fdn_div20(r, R.mg_itg);
        const int pg = r - line * R.tg;
        const int zd = fdn_div20(line, R.mg_ihh);
        int qd = q0d + zd, qh = q0h + (line - zd * R.hh);
        const int qw0 = q0w + 4 * pg;
        bool okl = oki;
        if (p.zero_mode) okl = okl && (unsigned)qd < (unsigned)p.ID && (unsigned)qh < (unsigned)p.IH;
        else { qd = min(max(qd, 0), p.ID - 1); qh = min(max(qh, 0), p.IH - 1); }
        const int lbase = (qd * p.IH + qh) * p.IW;
#pragma unroll
        for (int nn = 0; nn < 6; ++nn) {
            int qw = qw0 + nn;
            bool ok = okl;
            if (GEN && R.wface) { qw = nn == 0 ? p.IW - 1 : 0; ok = ok && (nn == 0 || nn == 5); }     // V_0 = 4 x[IW-1], V_5 = x[0]
            else if (p.zero_mode) ok = ok && (unsigned)qw < (unsigned)p.IW;
            else qw = min(max(qw, 0), p.IW - 1);
            soff[u][nn] = ok ? (unsigned)(lbase + qw) * 256u + (unsigned)(chunk * 16) : 0xffffffffu;
        }
        vrow[u] = oki ? r * LROW + chunk * 16 : -1;
    }
    const size_t in_n = (size_t)n * p.ID * p.IH * p.IW;
    const unsigned sample_bytes = (unsigned)(p.ID * p.IH * p.IW) * 256u;

    // weight stream: unit (2048 B) index = half*216 + (tap*6 + xi)*4 + k-group-in-half
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.up, 0, 54 * 64 * 64 * 4 + (GEN ? p.wspan : 0), 0x00020000);
    // GEN: the slice index runs over CS x nsrc (multi-source fused dgrad): slice sl_ of source sl_ / CS
    const int nsl = GEN ? CS * p.nsrc : CS;
    auto src_x = [&](int sl_) { return (!GEN || sl_ < CS) ? p.x : (sl_ < 2 * CS ? p.x1 : p.x2); };
    auto src_wd = [&](int sl_) { return (!GEN || sl_ < CS) ? 0 : (sl_ < 2 * CS ? p.wd1 : p.wd2); };
    const int wvoff = (kh * 64 + wave_n * 32 + li) * 16;
    f32x4 A[RDA], B[RDB];
    const int bmul = (FDN_DBG_BITS(p) & 1) ? 0 : 2048;
    auto wsoff = [&](int sl_, int tap, int jj) -> int {      // jj = xi*KG + g within the tap
        const int sc = GEN ? sl_ & (CS - 1) : sl_;           // slice within its source
        return ((((sc * KG) >> 2) * 216) + tap * 24 + ((sc * KG) & 3) + (jj / KG) * 4 + (jj % KG)) * bmul + src_wd(sl_);
    };
    auto ldb = [&](int slot, int so) {
        B[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, so, 0));
    };
    auto lda = [&](int slot, int tapb, int jj) {             // tapb: byte offset of the (a,b) tap's line
        A[slot] = *(const f32x4*)(smem + abase + tapb + (jj / KG) * planeb + (jj % KG) * 32);
    };

    if (GEN && R.wface) {
        // ---- w-face tile (see WinoRegion): its own slice loop, so that nothing of the main path's fragment rings is live here.
        // Staging: two loads per item (x[IW-1] -> plane 0 scaled by 4, x[0] -> plane 5), planes 1..4 are never read.
        // K loop: 9 (kd,kh) taps x xi in {0, 5} x KG k-groups = a third of a tile's MFMAs; weight fragments are refilled in place
        // for the next tap three steps (12 MFMAs) ahead, voxel fragments one step ahead.
        constexpr int NE = 2 * KG;
        auto jj_of = [](int e) { return (e < KG ? 0 : 5) * KG + (e % KG); };
#pragma unroll 1
        for (int sl = 0; sl < nsl; ++sl) {
            if (sl) __syncthreads();
            {
                const int sc = sl & (CS - 1);
                const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
                    (void*)(src_x(sl) + in_n * 64 + sc * (64 / CS)), 0, sample_bytes - sc * (256 / CS), 0x00020000);
                f32x4 xr[UA], xl[UA];
#pragma unroll
                for (int u = 0; u < UA; ++u) {
                    if (u * 256 >= R.items) break;
                    xr[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, soff[u][0], 0, 0));
                    xl[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, soff[u][5], 0, 0));
                }
#pragma unroll
                for (int u = 0; u < UA; ++u) {
                    if (u * 256 >= R.items) break;
                    if (vrow[u] < 0) continue;
                    *(f32x4*)(smem + vrow[u]) = 4.f * xr[u];                     // V_0 = 4 x0 - 5 x2 + x4 with x2 = x4 = 0
                    *(f32x4*)(smem + vrow[u] + 5 * planeb) = xl[u];              // V_5 = 4 x1 - 5 x3 + x5 with x1 = x3 = 0
                }
            }
            __syncthreads();
            f32x4 Bw[NE], Aw[2];
            auto ldbw = [&](int e, int tap) {
                Bw[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, wsoff(sl, tap, jj_of(e)), 0));
            };
            auto ldaw = [&](int slot, int tapb, int e) {
                Aw[slot] = *(const f32x4*)(smem + abase + tapb + (jj_of(e) / KG) * planeb + (jj_of(e) % KG) * 32);
            };
#pragma unroll
            for (int e = 0; e < NE; ++e) ldbw(e, 0);
            ldaw(0, 0, 0);
#pragma unroll 1
            for (int tap9 = 0; tap9 < 9; ++tap9) {
                const int tn = tap9 < 8 ? tap9 + 1 : 8;          // the last tap reloads itself (harmless)
                const int ta_ = tap9 / 3, tn_a = tn / 3;
                const int tapb = (ta_ * R.hh + (tap9 - 3 * ta_)) * R.tg * LROW;
                const int tapb_n = (tn_a * R.hh + (tn - 3 * tn_a)) * R.tg * LROW;
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    if (e + 1 < NE) ldaw((e + 1) & 1, tapb, e + 1);
                    else ldaw((e + 1) & 1, tapb_n, 0);
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        if (e < KG) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Bw[e][s4], Aw[e & 1][s4], acc[0], 0, 0, 0);
                        else acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(Bw[e][s4], Aw[e & 1][s4], acc[5], 0, 0, 0);
                    }
                    ldbw(e, tn);
                }
            }
        }
        // padded scratch only (shell positions are finished by the border fold): left face at w = 0, right face at w = OW-1
        const int g0w = mtab[wave_m * 32 + li];
        if (g0w < 0) return;
        float* yl = p.y + (size_t)g0w * 64 + cofs;
        float* yr = yl + (size_t)(p.OW - 1) * 64;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *(f32x4*)(yl + q * 4) = (f32x4){acc[5][q * 4], acc[5][q * 4 + 1], acc[5][q * 4 + 2], acc[5][q * 4 + 3]};
            *(f32x4*)(yr + q * 4) = (f32x4){acc[0][q * 4], acc[0][q * 4 + 1], acc[0][q * 4 + 2], acc[0][q * 4 + 3]};
        }
        return;
    }

#pragma unroll 1
    for (int sl = 0; sl < nsl; ++sl) {
        if (sl) __syncthreads();                             // everyone finished reading the previous slice
        // ---- stage + transform cin [sl*64/CS, (sl+1)*64/CS): all loads in flight before the first use ----
        {
            const int sc = GEN ? sl & (CS - 1) : sl;
            const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
                (void*)(src_x(sl) + in_n * 64 + sc * (64 / CS)), 0, sample_bytes - sc * (256 / CS), 0x00020000);
            f32x4 xv[UA][6];
            const int items_eff = (FDN_DBG_BITS(p) & 4) ? 0 : R.items;
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                if (u * 256 >= items_eff) break;
#pragma unroll
                for (int nn = 0; nn < 6; ++nn)
                    xv[u][nn] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, soff[u][nn], 0, 0));
            }
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                if (u * 256 >= items_eff) break;
                if (vrow[u] < 0) continue;
                if (FDN_DBG_BITS(p) & 16) {                            // ablation: raw rows, no transform arithmetic
                    char* vq = smem + vrow[u];
#pragma unroll
                    for (int nn = 0; nn < 6; ++nn) *(f32x4*)(vq + nn * planeb) = xv[u][nn];
                    continue;
                }
                // B^T of F(4,3): rows (4,0,-5,0,1,0) (0,-4,-4,1,1,0) (0,4,-4,-1,1,0) (0,-2,-1,2,1,0) (0,2,-1,-2,1,0) (0,4,0,-5,0,1)
                const f32x4 x0 = xv[u][0], x1 = xv[u][1], x2 = xv[u][2], x3 = xv[u][3], x4 = xv[u][4], x5 = xv[u][5];
                const f32x4 t1 = x4 - 4.f * x2, t2 = x3 - 4.f * x1;
                const f32x4 t3 = x4 - x2, t4 = 2.f * (x3 - x1);
                char* vp = smem + vrow[u];
                *(f32x4*)(vp) = 4.f * x0 - 5.f * x2 + x4;
                *(f32x4*)(vp + planeb) = t1 + t2;
                *(f32x4*)(vp + 2 * planeb) = t1 - t2;
                *(f32x4*)(vp + 3 * planeb) = t3 + t4;
                *(f32x4*)(vp + 4 * planeb) = t3 - t4;
                *(f32x4*)(vp + 5 * planeb) = 4.f * x1 - 5.f * x3 + x5;
            }
        }
        __syncthreads();

        // ---- K loop: (kd,kh) taps of the region x 6 xi x KG k-groups ----
        {
            const int tap_first = ta0 * 3 + tb0;
            const int ntap9 = (ta1 - ta0 + 1) * (tb1 - tb0 + 1);
            if (sl == 0) {
#pragma unroll
                for (int j = 0; j < RDB - 1; ++j) ldb(j, wsoff(0, tap_first, j));
            }
#pragma unroll
            for (int j = 0; j < RDA - 1; ++j) lda(j, 0, j);
            const int sln = sl + 1 < nsl ? sl + 1 : sl;      // harmless reload after the last slice
            int ta = ta0, tb = tb0, tapb = 0;
#pragma unroll 1
            for (int it = 0; it < ntap9; ++it) {
                int na = ta, nb = tb + 1;
                if (nb > tb1) { nb = tb0; ++na; }
                const bool last = it + 1 == ntap9;
                const int tapb_n = last ? tapb : ((na - ta0) * R.hh + (nb - tb0)) * R.tg * LROW;
                const int tap_c = ta * 3 + tb;
                const int tap_n = last ? tap_first : na * 3 + nb;
                const int sl_n = last ? sln : sl;
#pragma unroll
                for (int j = 0; j < SPT; ++j) {
                    const int sb = j % RDB, sa = j % RDA;
                    const int xi = j / KG;
                    acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(B[sb][0], A[sa][0], acc[xi], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    {
                        // the steps whose operands go into the slots step j-1 just freed
                        const int jb = j + RDB - 1, ja = j + RDA - 1;
                        if (jb < SPT) ldb(jb % RDB, wsoff(sl, tap_c, jb));
                        else ldb(jb % RDB, wsoff(sl_n, tap_n, jb - SPT));
                        if (ja < SPT) lda(ja % RDA, tapb, ja);
                        else lda(ja % RDA, tapb_n, ja - SPT);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s = 1; s < 4; ++s)
                        acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(B[sb][s], A[sa][s], acc[xi], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                ta = na; tb = nb; tapb = tapb_n;
            }
        }
    }

    if (FDN_DBG_BITS(p) & 8) return;
    // ---- epilogue: Y = A^T M, A^T = (1,1,1,1,1,0) (0,1,-1,2,-2,0) (0,1,1,4,4,0) (0,1,-1,8,-8,1); lane = one group x 16 cout ----
    const int m = wave_m * 32 + li;
    const int g0 = mtab[m];
    if (g0 < 0) return;
    const int gf0 = mtab[64 + m];
    const int iw0 = mtab[128 + m];
    const float slope = p.act == FDN_ACT_RELU ? 0.f : (p.act == FDN_ACT_LEAKY ? p.alpha : 1.f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x4 z[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = q * 4 + e;
                const float s12 = acc[1][r] + acc[2][r], d12 = acc[1][r] - acc[2][r];
                const float s34 = acc[3][r] + acc[4][r], d34 = acc[3][r] - acc[4][r];
                float v;
                if (i == 0) v = acc[0][r] + s12 + s34;
                else if (i == 1) v = d12 + 2.f * d34;
                else if (i == 2) v = s12 + 4.f * s34;
                else v = d12 + 8.f * d34 + acc[5][r];
                z[q][e] = v;
            }
        if (p.fout) {
            const int iw = iw0 + i;
            if (gf0 >= 0 && iw >= 1 && iw <= p.IW - 2) {
                // strictly inside the volume: exactly one contribution -> finish dz_prev = (dgrad + skip) * act'(y) here
                const size_t o = (size_t)(gf0 + i) * 64 + cofs;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 sk = p.fskip ? *(const f32x4*)(p.fskip + o + q * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
                    const f32x4 ym = p.fy ? *(const f32x4*)(p.fy + o + q * 4) : (f32x4){1.f, 1.f, 1.f, 1.f};
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (z[q][e] + sk[e]) * (ym[e] > 0.f ? 1.f : slope);
                    *(f32x4*)(p.fout + o + q * 4) = v;
                }
            } else {
                const size_t o = (size_t)(g0 + i) * 64 + cofs;     // surface voxel: padded scratch, finished by the border fold
#pragma unroll
                for (int q = 0; q < 4; ++q) *(f32x4*)(p.y + o + q * 4) = z[q];
            }
        } else {
            const size_t o = (size_t)(g0 + i) * 64 + cofs;
            if (p.res) {
#pragma unroll
                for (int q = 0; q < 4; ++q) z[q] += *(const f32x4*)(p.res + o + q * 4);
            }
            if (p.bias) {
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r >> 2][r & 3] += p.bias[cofs + r];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = z[q][e];
                    z[q][e] = fmaxf(t, slope * t);             // relu / leaky / none: slope in [0,1]
                }
                if (!(FDN_DBG_BITS(p) & 32) || z[q][0] == 12345.678f) *(f32x4*)(p.y + o + q * 4) = z[q];
            }
        }
    }
}

}  // namespace

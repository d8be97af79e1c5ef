// conv3d 3x3x3, 64 -> 64 channels, fp32, NDHWC: implicit GEMM on v_mfma_f32_32x32x2_f32.
//
// Replaces tf.pad(SYMMETRIC) + Conv3D(valid) + BiasAdd + ReLU/LeakyReLU (+ the resnet_block add)
// of src/Network/SR4DFlowNet.py:93-120, and -- in "zero" boundary mode on the padded output grid --
// Conv3DBackpropInputV2 of the same layers (+ MirrorPadGrad for interior voxels, see the epilogue).
//
// Work decomposition: M = output voxels of a box tile, N = 64 cout, K = 27 taps x 64 cin.
//   * One workgroup (4 waves) per box tile.  The tile's input box + 1-voxel halo is staged in CS slices of
//     64/CS input channels into LDS (256/CS bytes per voxel row + a 16-B pad: with a row stride of 144 B / 80 B the 16 rows
//     a ds_read_b128 lane group touches fall on 16 different 4-bank groups, so no XOR swizzle -- and no per-read vector
//     address arithmetic -- is needed: the A-fragment address is lane base + tap offset (one v_add) + an immediate).  The boundary rule (edge clamp for forward == SYMMETRIC p=1, zero for
//     dgrad) is applied while staging, so the K loop is branch-free.
//   * Latency hiding is by co-resident workgroups, not by software pipelining across phases: 2 workgroups per CU with
//     CS=2 (<= 80 KB LDS each; the variant that measures best on large grids), up to 4-6 with CS=4 / the small layouts;
//     while one stages or stores, the others keep the matrix pipe busy.  Within a phase, ALL staging loads of a slice are
//     issued before the first LDS write (one memory round trip).  Measured on MI355X: persistent register-prefetching,
//     producer/consumer wave specialisation, start staggering and s_setprio were all slower or null -- see DESIGN.md.
//   * One launch covers up to 7 REGIONS (conv64_args.h): forward = 1 region; fused dgrad = the inner D^3 box + six
//     9-tap shell slabs of the padded grid.
//   * Wave layouts <MT,NW>:  <2,1>: tile 256 voxels, wave = 64 voxels x 64 cout;  <1,1>: 128 voxels, wave =
//     32 x 64;  <1,2>: 64 voxels, wave = 32 voxels x 32 cout (fine load balance on small grids).
//     Accumulators: MT x (2/NW) tiles of 32x32 (16 VGPRs each).
//   * A fragments: one ds_read_b128 per (M-tile, 8 cin) -- lane (i,kh) gets cin 8g+4kh+{0..3} of voxel i;
//     MFMA step s contracts the cin pair {8g+s, 8g+4+s}.  The K order is a permutation of cin, matched by the
//     packed weight stream, so no data movement is needed to form fragments.
//   * B fragments (weights) are NOT staged: the packed stream [cin/32][tap][g][kh][cout][s] is read straight from
//     L1/L2 with one global_load_dwordx4 per (N-tile, 8 cin).  442 KB of weights are shared by every workgroup
//     on the chip, so they stay cache-resident.
//   fp32 MFMA rate is 64 cyc / instruction / SIMD: per 8 cin a wave issues 4*MT*(2/NW) MFMAs against MT LDS reads and
//   2/NW global loads, so the matrix pipe is the only busy resource by construction.
#include "fdn_common.h"
#include "conv64_args.h"
#include "conv64_pack.h"

// test/bench hooks: compile-time constants in the product library, settable through fdn_debug_* in the test build only
FDN_HOOK_VAR(int, fdn_conv64_force_layout, 0);   // 0 = auto, 1..6 = index into the variant table below
FDN_HOOK_VAR(int, fdn_conv64_dbg, 0);            // ablation bits, see Conv64Args::dbg
FDN_HOOK_VAR(int, fdn_conv64_split_dgrad, 0);    // fused dgrad, 2-D Winograd path: 1 = inner box and shell faces as two launches (until round 4's last change)
FDN_HOOK_VAR(int, fdn_conv64_wface_direct, 0);   // fused dgrad, Winograd path: 1 = the w faces as a separate direct-kernel launch (round 2), 0 = a region of the Winograd launch
FDN_HOOK_VAR(int, fdn_conv64_shell_slabs, 1);    // fused dgrad: 1 = inner box + 6 shell slabs, 0 = one launch over the padded grid

template <int MT, int NW, int CS>
struct Conv64Cfg {
    static constexpr int WM = 4 / NW;                 // wave rows
    static constexpr int NT = 2 / NW;                 // 32-wide cout tiles per wave
    static constexpr int MCAP = WM * MT * 32;         // voxels per tile
    static constexpr int ROWB = 256 / CS;             // bytes per staged voxel row
    static constexpr int LROW = ROWB + 16;            // LDS row stride: one 16-B pad per row (see the kernel header)
    static constexpr int CH = ROWB / 16;              // 16-B chunks per row
    static constexpr int KG = 8 / CS;                 // k-groups (8 cin) per staged slice
    // workgroups per CU the variant is sized for (VGPR cap via launch bounds, LDS via MAXROWS)
    static constexpr int WG_PER_CU = MCAP == 256 ? (CS == 4 ? 3 : 2) : (MCAP == 128 ? (CS == 4 ? 4 : 3) : (CS == 4 ? 6 : 4));   // <1,1,4>: 4, not the 5 its LDS would allow -- a 96-VGPR cap made it spill 32 registers
    static constexpr int MAXROWS = ((160 * 1024 / WG_PER_CU) - MCAP * 4 - 256) / LROW > 1000
                                       ? 1000 : ((160 * 1024 / WG_PER_CU) - MCAP * 4 - 256) / LROW;
    static constexpr int LDS_BYTES = MAXROWS * LROW + MCAP * 4;
};

// GEN = false: a single region with all 27 taps (every forward launch) -- tap ranges are compile-time constants.
template <int MT, int NW, int CS, bool GEN>
__global__ __launch_bounds__(256, (Conv64Cfg<MT, NW, CS>::WG_PER_CU > 8 ? 8 : Conv64Cfg<MT, NW, CS>::WG_PER_CU))
void conv64_mfma_kernel(Conv64Args p) {
    using C = Conv64Cfg<MT, NW, CS>;
    constexpr int NT = C::NT, LROW = C::LROW, CH = C::CH, KG = C::KG;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int kh = lane >> 5;
    const int wave_m = wave % C::WM;
    const int wave_n = wave / C::WM;

    // ---- which region, which tile ----
    int ri = 0;
    if (GEN) {
        while (ri + 1 < p.nreg && (int)blockIdx.x >= p.reg[ri + 1].first_block) ++ri;
        ri = __builtin_amdgcn_readfirstlane(ri);  // provably wave-uniform -> the region fields load into SGPRs
    }
    const Conv64Region R = p.reg[ri];
    const int ta0 = GEN ? R.ta0 : 0, ta1 = GEN ? R.ta1 : 2, tb0 = GEN ? R.tb0 : 0, tb1 = GEN ? R.tb1 : 2;
    const int tc0 = GEN ? R.tc0 : 0, tc1 = GEN ? R.tc1 : 2;
    // tile coordinates by multiply-shift with host-made magics: runtime integer division costs ~20 VALU instructions, and
    // on this part every one of them takes its cycles from the fp32 MFMA stream of the SIMD
    const int tiles_per_n = R.ntd * R.nth * R.ntw;
    int b = (int)blockIdx.x - R.first_block;
    const int n = fdn_udiv40(b, R.mg_tpn_hi, R.mg_tpn_lo);
    b -= n * tiles_per_n;
    const int tdi = fdn_udiv40(b, R.mg_thw_hi, R.mg_thw_lo);
    b -= tdi * (R.nth * R.ntw);
    const int thi = fdn_udiv40(b, R.mg_ntw_hi, R.mg_ntw_lo);
    int* mtab = (int*)(smem + R.rows * LROW);          // output voxel index of each tile row, -1 if unused
    const int p0d = R.obd + tdi * R.td, p0h = R.obh + thi * R.th, p0w = R.obw + (b - thi * R.ntw) * R.tw;

    const int nv = R.td * R.th * R.tw;
    const int thtw = R.th * R.tw;

    // ---- output voxel of each tile row: padded-grid voxel index (bit 31 clear), or -1 ----
    for (int m = tid; m < C::MCAP; m += 256) {
        int g = -1;
        if (m < nv) {
            const int md = fdn_div20(m, R.mg_thtw);
            const int r2 = m - md * thtw;
            const int mh = fdn_div20(r2, R.mg_tw);
            const int pd = p0d + md, ph = p0h + mh, pw = p0w + (r2 - mh * R.tw);
            if (pd < R.obd + R.ebd && ph < R.obh + R.ebh && pw < R.obw + R.ebw) {
                g = ((n * p.OD + pd) * p.OH + ph) * p.OW + pw;
                if (p.fout) {
                    // fused fold: rows strictly inside the input volume are redirected to dz_prev (tag bit 30)
                    const int id = pd - 1, ih = ph - 1, iw = pw - 1;
                    if (id >= 1 && id <= p.ID - 2 && ih >= 1 && ih <= p.IH - 2 && iw >= 1 && iw <= p.IW - 2)
                        g = (((n * p.ID + id) * p.IH + ih) * p.IW + iw) | (1 << 30);
                }
            }
        }
        mtab[m] = g;
    }

    // ---- this lane's A rows (LDS halo row of voxel m at tap (0,0,0)) ----
    int row0[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        int m = (wave_m * MT + mi) * 32 + li;
        m = m < nv ? m : nv - 1;
        const int md = fdn_div20(m, R.mg_thtw);
        const int r2 = m - md * thtw;
        const int mh = fdn_div20(r2, R.mg_tw);
        row0[mi] = (md * R.hh + mh) * R.hw + (r2 - mh * R.tw);
    }
    int abase[MT];                                      // LDS byte offset of this lane's A fragment at tap (0,0,0), k-group 0
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) abase[mi] = row0[mi] * LROW + kh * 16;

    // The MFMAs run as D[cout][voxel] (the weights are the row operand) and the packed stream permutes the cout rows so that
    // the 16 accumulator registers of lane (li, kh) are 16 CONSECUTIVE output channels of voxel li: cout = 32 nn + 16 kh + r
    // (+ 32 NT wave_n).
    const int cofs = wave_n * (NT * 32) + kh * 16;
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int nn = 0; nn < NT; ++nn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][nn][r] = 0.f;

    const int chunk = tid % CH;
    const int rsub = tid / CH;
    constexpr int RPP = 256 / CH;                      // rows staged per pass
    const int rows_eff = (p.dbg & 4) ? 0 : R.rows;
    const int bstride = (p.dbg & 1) ? 0 : 128;
    const size_t in_n = (size_t)n * p.ID * p.IH * p.IW;
    // staged box origin in input coordinates: output p reads input p + tap - 1 + off, first staged tap is (ta0,tb0,tc0)
    const int q0d = p0d - 1 + p.off + ta0, q0h = p0h - 1 + p.off + tb0, q0w = p0w - 1 + p.off + tc0;
    const int ntap = (ta1 - ta0 + 1) * (tb1 - tb0 + 1) * (tc1 - tc0 + 1);

    // fragment ring + its loaders (see the K loop)
    f32x4 A[KG][MT], B[KG][NT];
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, 27 * 64 * 64 * 4, 0x00020000);
    const int wvoff = (kh * 64 + wave_n * (NT * 32) + li) * 16;
    const int tap_first = (ta0 * 3 + tb0) * 3 + tc0;
    // byte offset of (slice, tap, local k-group) in the packed stream [cin/32][tap][g][kh][cout][s]
    auto wsoff = [&](int sl_, int tap, int g) -> int {
        return ((((sl_ * KG) >> 2) * 27 * 4 + ((sl_ * KG) & 3)) * 128 + (tap * 4 + g) * bstride) * 16;
    };
    auto ldb = [&](int g, int so) {
#pragma unroll
        for (int nn = 0; nn < NT; ++nn)
            B[g][nn] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff + nn * 512, so, 0));
    };
    auto lda = [&](int g, int tapoff_) {
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            A[g][mi] = *(const f32x4*)(smem + abase[mi] + tapoff_ * LROW + g * 32);
        }
    };

    // ---- staging plan, once per tile: byte offset (from the sample's first voxel) of each halo row this thread stages, with the
    // boundary rule applied -- edge clamp, or 0xffffffff for rows the dgrad mode reads as zero (a buffer load past
    // num_records returns 0) and for rows past the box.  The slices then cost one buffer load per row and no vector ALU
    // work: on this part every VALU instruction of every wave takes its cycles from the fp32 MFMA stream. ----
    constexpr int UA = (C::MAXROWS + RPP - 1) / RPP;
    unsigned soff[UA];
#pragma unroll
    for (int u = 0; u < UA; ++u) {
        const int r = u * RPP + rsub;
        const int zd = fdn_div20(r, R.mg_hhhw);
        const int r2 = r - zd * R.hh * R.hw;
        const int zh = fdn_div20(r2, R.mg_hw);
        int qd = q0d + zd, qh = q0h + zh, qw = q0w + (r2 - zh * R.hw);
        bool ok = r < rows_eff;
        if (p.zero_mode) {
            ok = ok && (unsigned)qd < (unsigned)p.ID && (unsigned)qh < (unsigned)p.IH && (unsigned)qw < (unsigned)p.IW;
        } else {
            qd = min(max(qd, 0), p.ID - 1);
            qh = min(max(qh, 0), p.IH - 1);
            qw = min(max(qw, 0), p.IW - 1);
        }
        soff[u] = ok ? (unsigned)((qd * p.IH + qh) * p.IW + qw) * 256u + (unsigned)(chunk * 16) : 0xffffffffu;
    }
    const int nfull = rows_eff / RPP;                       // passes in which every thread has a row
    const bool tail = rsub < rows_eff - nfull * RPP;        // this thread has a row in the last, partial pass
    const unsigned sample_bytes = (unsigned)(p.ID * p.IH * p.IW) * 256u;

#pragma unroll 1
    for (int sl = 0; sl < CS; ++sl) {
        if (sl) __syncthreads();  // everyone finished reading the previous slice
        // ---- stage input box + halo, cin [sl*64/CS, (sl+1)*64/CS) ----
        const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(p.x + in_n * 64 + sl * (64 / CS)), 0, sample_bytes - sl * (256 / CS), 0x00020000);
        // all loads of the slice are issued before the first LDS write (one memory round trip)
        constexpr int U = UA;
#pragma unroll
        for (int u0 = 0; u0 < UA; u0 += U) {
            if (u0 * RPP >= rows_eff) break;
            f32x4 v[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (u0 + u < UA) v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, soff[u0 + u], 0, 0));
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (u0 + u < UA && (u0 + u < nfull || (u0 + u == nfull && tail)))
                    *(f32x4*)(smem + ((u0 + u) * RPP + rsub) * LROW + (chunk << 4)) = v[u];
        }
        __syncthreads();

        // ---- K loop over the taps x KG k-groups (8 cin) of this slice ----
        // One step = one k-group of one tap: MT A fragments (LDS) + NT B fragments (weight stream) feed 4*MT*NT MFMAs.  The
        // fragments live in a ring of KG slots indexed by the k-group: right after the first MFMAs of step g have issued,
        // slot g-1 -- consumed by the previous step -- is refilled for its next use KG-1 steps later, so every load and LDS
        // read has ~KG-1 steps to land and no wave ever waits; no register copies, no waits at tap boundaries, and the
        // refill addresses are scalar (+ one v_add per tap): an fp32 MFMA does not overlap with any other vector instruction
        // of its SIMD (machine model), so what the loop saves is instructions, not exposed latency.
        // (Measured: with the side work batched at the tap boundary -- vmcnt(0), 32 v_mov, 64-bit address arithmetic -- the
        // K loop lost ~7 %.)
        {
            int ta = ta0, tb = tb0, tc = tc0, tapoff = 0;
            if (sl == 0) {
#pragma unroll
                for (int g = 0; g < KG - 1; ++g) ldb(g, wsoff(0, tap_first, g));
            }
#pragma unroll
            for (int g = 0; g < KG - 1; ++g) lda(g, 0);
            const int sln = sl + 1 < CS ? sl + 1 : sl;            // harmless reload after the last slice
            // one tap = KG steps; (tapoff, tap) of this tap and of the next one (for the refills) come in as scalars
            auto tap_body = [&](int tapoff, int tapoff_n, int tap_cur, int tap_nxt, int sl_nxt) {
#pragma unroll
                for (int g = 0; g < KG; ++g) {
#pragma unroll
                    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                        for (int nn = 0; nn < NT; ++nn)
                            acc[mi][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(B[g][nn][0], A[g][mi][0], acc[mi][nn], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (g == 0) {
                        ldb(KG - 1, wsoff(sl, tap_cur, KG - 1));
                        lda(KG - 1, tapoff);
                    } else {
                        ldb(g - 1, wsoff(sl_nxt, tap_nxt, g - 1));
                        lda(g - 1, tapoff_n);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int s = 1; s < 4; ++s)
#pragma unroll
                        for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                            for (int nn = 0; nn < NT; ++nn)
                                acc[mi][nn] = __builtin_amdgcn_mfma_f32_32x32x2f32(B[g][nn][s], A[g][mi][s], acc[mi][nn], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            {
#pragma unroll 1
                for (int it = 0; it < ntap; ++it) {
                    int na = ta, nb = tb, nc = tc;
                    if (++nc > tc1) { nc = tc0; if (++nb > tb1) { nb = tb0; ++na; } }
                    const bool last = it + 1 == ntap;
                    const int tapoff_n = last ? tapoff : ((na - ta0) * R.hh + (nb - tb0)) * R.hw + (nc - tc0);
                    tap_body(tapoff, tapoff_n, (ta * 3 + tb) * 3 + tc, last ? tap_first : (na * 3 + nb) * 3 + nc, last ? sln : sl);
                    ta = na; tb = nb; tc = nc; tapoff = tapoff_n;
                }
            }
        }
    }
    if (p.dbg & 8) return;

    // ---- epilogue: every lane owns ONE output voxel per M tile (see the accumulator layout above): one index lookup, 16-B
    // vector loads / stores, and the activation as max(z, slope*z) for slope in [0,1] (relu / leaky / none).  Everything
    // here is paid in MFMA cycles (machine model), so it is kept to ~3 VALU instructions per element. ----
    const float slope = p.act == FDN_ACT_RELU ? 0.f : (p.act == FDN_ACT_LEAKY ? p.alpha : 1.f);
    const bool act_max = slope <= 1.f;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const int g = mtab[(wave_m * MT + mi) * 32 + li];
        if (g < 0) continue;
        if (p.fout) {
            // dgrad with fused fold: tagged voxels (strictly inside the volume, exactly one contribution) are finished
            // here into dz_prev; everything else goes to the padded scratch for the border fold.
            const size_t o = (size_t)(g & ~(1 << 30)) * 64 + cofs;
            if (g & (1 << 30)) {
                f32x4 sk[NT][4], ym[NT][4];
#pragma unroll
                for (int nn = 0; nn < NT; ++nn)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        sk[nn][q] = p.fskip ? *(const f32x4*)(p.fskip + o + nn * 32 + q * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
                        ym[nn][q] = p.fy ? *(const f32x4*)(p.fy + o + nn * 32 + q * 4) : (f32x4){1.f, 1.f, 1.f, 1.f};
                    }
#pragma unroll
                for (int nn = 0; nn < NT; ++nn)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            v[e] = (acc[mi][nn][q * 4 + e] + sk[nn][q][e]) * (ym[nn][q][e] > 0.f ? 1.f : slope);
                        *(f32x4*)(p.fout + o + nn * 32 + q * 4) = v;
                    }
            } else {
#pragma unroll
                for (int nn = 0; nn < NT; ++nn)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[mi][nn][q * 4 + e];
                        *(f32x4*)(p.y + o + nn * 32 + q * 4) = v;
                    }
            }
        } else {
            const size_t o = (size_t)g * 64 + cofs;
            f32x4 z[NT][4];
            if (p.res) {
#pragma unroll
                for (int nn = 0; nn < NT; ++nn)
#pragma unroll
                    for (int q = 0; q < 4; ++q) z[nn][q] = *(const f32x4*)(p.res + o + nn * 32 + q * 4);
#pragma unroll
                for (int nn = 0; nn < NT; ++nn)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) z[nn][q][e] += acc[mi][nn][q * 4 + e];
            } else {
#pragma unroll
                for (int nn = 0; nn < NT; ++nn)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int e = 0; e < 4; ++e) z[nn][q][e] = acc[mi][nn][q * 4 + e];
            }
            if (p.bias) {             // 4 of the 30 layers; dword loads: the bias lives in the flat parameter buffer, 4-B aligned
#pragma unroll
                for (int nn = 0; nn < NT; ++nn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[nn][r >> 2][r & 3] += p.bias[cofs + nn * 32 + r];
            }
#pragma unroll
            for (int nn = 0; nn < NT; ++nn)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = z[nn][q][e];
                        z[nn][q][e] = act_max ? fmaxf(t, slope * t) : (t > 0.f ? t : slope * t);
                    }
                    *(f32x4*)(p.y + o + nn * 32 + q * 4) = z[nn][q];
                }
        }
    }
}

// --------------------------------------------------------------------------------------------
// border fold: finishes the voxels on the volume surface after fused-fold dgrad launches.
//   dz_prev[i] = (sum_s sum_{P: clamp(P)=i} dxpad_s[P] + skip[i]) * act'(y[i])   for i with some i_d in {0, D-1}
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fold_halo_border_kernel(const float* __restrict__ s0, const float* __restrict__ s1,
                                                                const float* __restrict__ s2, int nsrc,
                                                                const float* __restrict__ skip, const float* __restrict__ yprev,
                                                                int act, float alpha, float* __restrict__ out, int N, int D,
                                                                int H, int W) {
    // surface voxels enumerated as: two full d-faces, then two h-faces without the d-faces, then two w-faces without both
    const int ID = D > 2 ? D - 2 : 0, IH = H > 2 ? H - 2 : 0;
    const int nd_faces = (D > 1 ? 2 : 1) * H * W;
    const int nh_faces = ID * (H > 1 ? 2 : 1) * W;
    const int nw_faces = ID * IH * (W > 1 ? 2 : 1);
    const int per_n = nd_faces + nh_faces + nw_faces;
    const int64_t total = (int64_t)N * per_n * 16;
    const int PH = H + 2, PW = W + 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i & 15);
        int s = (int)((i >> 4) % per_n);
        const int n = (int)((i >> 4) / per_n);
        int d, h, w;
        if (s < nd_faces) {
            d = (s / (H * W)) ? D - 1 : 0; s %= H * W; h = s / W; w = s % W;
        } else if ((s -= nd_faces) < nh_faces) {
            const int f = s / (ID * W); s %= ID * W; h = f ? H - 1 : 0; d = 1 + s / W; w = s % W;
        } else {
            s -= nh_faces;
            const int f = s / (ID * IH); s %= ID * IH; w = f ? W - 1 : 0; d = 1 + s / IH; h = 1 + s % IH;
        }
        int pd[3], ph[3], pw[3];
        int nd = 0, nh = 0, nw = 0;
        pd[nd++] = d + 1; if (d == 0) pd[nd++] = 0; if (d == D - 1) pd[nd++] = D + 1;
        ph[nh++] = h + 1; if (h == 0) ph[nh++] = 0; if (h == H - 1) ph[nh++] = H + 1;
        pw[nw++] = w + 1; if (w == 0) pw[nw++] = 0; if (w == W - 1) pw[nw++] = W + 1;
        // all loads of a voxel are requested before the first use: the (up to 2 x 2 x 2) padded positions as predicated loads of a
        // fully unrolled nest (a face voxel has 2, an edge 4, a corner 8), skip and the mask up front -- with runtime loop bounds
        // they came back one round trip at a time
        const int64_t o = ((((int64_t)n * D + d) * H + h) * W + w) * 16 + c4;
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
        const f32x4 sk = skip ? ((const f32x4*)skip)[o] : zero4;
        f32x4 y = {1.f, 1.f, 1.f, 1.f};
        if (yprev) y = ((const f32x4*)yprev)[o];
        f32x4 acc = zero4;
        if (D == 1 || H == 1 || W == 1) {          // an axis of extent 1 folds both of its padded neighbours onto the voxel: general nest
            for (int a = 0; a < nd; ++a)
                for (int b = 0; b < nh; ++b)
                    for (int c = 0; c < nw; ++c) {
                        const int64_t off = (((((int64_t)n * (D + 2) + pd[a]) * PH + ph[b]) * PW + pw[c]) * 16 + c4);
                        acc += ((const f32x4*)s0)[off];
                        if (nsrc > 1) acc += ((const f32x4*)s1)[off];
                        if (nsrc > 2) acc += ((const f32x4*)s2)[off];
                    }
        } else
        for (int sidx = 0; sidx < nsrc; ++sidx) {
            const f32x4* sp = (const f32x4*)(sidx == 0 ? s0 : (sidx == 1 ? s1 : s2));
            f32x4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int a = k >> 2, b = (k >> 1) & 1, c = k & 1;
                v[k] = zero4;
                if (a < nd && b < nh && c < nw)
                    v[k] = sp[((((int64_t)n * (D + 2) + pd[a]) * PH + ph[b]) * PW + pw[c]) * 16 + c4];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) acc += v[k];
        }
        acc += sk;
        if (yprev) {
            acc.x *= fdn_act_grad(y.x, act, alpha); acc.y *= fdn_act_grad(y.y, act, alpha);
            acc.z *= fdn_act_grad(y.z, act, alpha); acc.w *= fdn_act_grad(y.w, act, alpha);
        }
        ((f32x4*)out)[o] = acc;
    }
}

// --------------------------------------------------------------------------------------------
// weight packing (layouts: conv64_pack.h)
// --------------------------------------------------------------------------------------------
__global__ void pack_conv64_kernel(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wd) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // over 27*64*64 packed elements
    if (idx < 27 * 64 * 64) fdn_pack_direct_one(w, wf, wd, idx);
}

// every 64->64 layer of the network in ONE launch: blockIdx.y = layer, packs[layer][fwd|dgrad][direct | winograd]
// sf / sd: the streams to write of the forward / dgrad pack (bit 0 direct, 1 1-D Winograd, 2 F(2,3)xF(4,3), 3 F(4,3)xF(4,3)); a block
// lies inside one stream (every stream is a multiple of 256 elements), so an unwanted stream costs an early exit
__global__ void pack_conv64_batch_kernel(const float* __restrict__ w_base, const int64_t* __restrict__ w_offsets, float* __restrict__ packs,
                                         int sf, int sd) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;   // over the 153*64*64 packed elements of the first three streams
    if (idx >= 153 * 64 * 64) return;
    const float* w = w_base + w_offsets[blockIdx.y];
    float* pf = packs + (size_t)blockIdx.y * 2 * FDN_CONV64_PACK_FLOATS;
    float* pd = pf + FDN_CONV64_PACK_FLOATS;
    const int st = idx < 27 * 64 * 64 ? 0 : idx < 81 * 64 * 64 ? 1 : 2;
    if (!(((sf | sd) >> st) & 1)) return;
    if (!((sf >> st) & 1)) pf = nullptr;
    if (!((sd >> st) & 1)) pd = nullptr;
    if (st == 0) fdn_pack_direct_one(w, pf, pd, idx);
    else if (st == 1) fdn_pack_wino_one(w, pf ? pf + 27 * 64 * 64 : nullptr, pd ? pd + 27 * 64 * 64 : nullptr, idx - 27 * 64 * 64);
    else fdn_pack_wino2d_one(w, pf ? pf + 81 * 64 * 64 : nullptr, pd ? pd + 81 * 64 * 64 : nullptr, idx - 81 * 64 * 64);
}

// The two F(4,3) x F(4,3) streams (fp32: pack + 153*4096 floats; bf16 x 3: pack + 261*4096), for one layer (w_offsets == nullptr: w_base is
// the kernel, packs = forward pack, packs_d = dgrad pack, either may be null) or for blockIdx.y = layer of the flat parameter buffer
// (packs_d == nullptr: packs = [layer][fwd | dgrad][FDN_CONV64_PACK_FLOATS]).  Thread = (direction, kd, cin, cout): all 36 coordinates from
// one load of its 9 weights, in the fp32 stream's element order (consecutive lanes write consecutive floats).  sf / sd: stream masks.
__global__ void pack_conv64_wino44_kernel(const float* __restrict__ w_base, const int64_t* __restrict__ w_offsets, float* __restrict__ packs,
                                          float* __restrict__ packs_d, int sf, int sd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= 2 * 3 * 64 * 64) return;
    const bool dg = c >= 3 * 64 * 64;
    const int s = dg ? sd : sf;
    if (!(s & 24)) return;
    const float* w = w_offsets ? w_base + w_offsets[blockIdx.y] : w_base;
    float* pk = w_offsets ? packs + ((size_t)blockIdx.y * 2 + (dg ? 1 : 0)) * FDN_CONV64_PACK_FLOATS : (dg ? packs_d : packs);
    if (!pk) return;
    const int e = c - (dg ? 3 * 64 * 64 : 0);
    const int o = e & 1023, nb = (e >> 10) & 3, kd = e >> 12;
    fdn_pack_wino44_column(w, (s & 8) ? pk + 153 * 64 * 64 : nullptr, (s & 16) ? (uint16_t*)(pk + 261 * 64 * 64) : nullptr, kd,
                           (o >> 6) * 4 + (o & 3), nb * 16 + ((o >> 2) & 15), dg);
}

int fdn_pack_conv64_wino44_launch(const float* w, float* wp_fwd, float* wp_dgrad, hipStream_t s) {
    hipLaunchKernelGGL(pack_conv64_wino44_kernel, dim3((2 * 3 * 64 * 64 + 255) / 256, 1), dim3(256), 0, s, w, (const int64_t*)nullptr, wp_fwd, wp_dgrad, 24, 24);
    FDN_CHECK_LAUNCH("pack_conv64_wino44_kernel");
    return FDN_OK;
}

// pack = [direct stream, 27*64*64 floats | Winograd F(4,3) stream, 54*64*64 | 2-D F(2,3)xF(4,3) stream, 72*64*64 | 2-D F(4,3)xF(4,3)
// stream, 108*64*64 | the same as three bf16 pieces per value, 162*64*64 float-sized slots]  (FDN_CONV64_PACK_FLOATS in fdn.h)
constexpr int kDirectPackFloats = 27 * 64 * 64;
constexpr int kWino1PackFloats = 54 * 64 * 64;
constexpr int kWino2PackFloats = 72 * 64 * 64;
constexpr int kWino44PackFloats = 108 * 64 * 64;
static_assert(kDirectPackFloats + kWino1PackFloats + kWino2PackFloats + kWino44PackFloats + 162 * 64 * 64 == FDN_CONV64_PACK_FLOATS, "pack layout");

extern "C" int fdn_pack_conv64_weights(const float* w, float* wp_fwd, float* wp_dgrad, void* stream) {
    FDN_REQUIRE(w != nullptr, "fdn_pack_conv64_weights: w is NULL");
    hipLaunchKernelGGL(pack_conv64_kernel, dim3((27 * 64 * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w,
                       wp_fwd, wp_dgrad);
    FDN_CHECK_LAUNCH("fdn_pack_conv64_weights");
    if (int rc = fdn_pack_conv64_wino_launch(w, wp_fwd ? wp_fwd + kDirectPackFloats : nullptr,
                                             wp_dgrad ? wp_dgrad + kDirectPackFloats : nullptr, (hipStream_t)stream))
        return rc;
    if (int rc = fdn_pack_conv64_wino2d_launch(w, wp_fwd ? wp_fwd + kDirectPackFloats + kWino1PackFloats : nullptr,
                                               wp_dgrad ? wp_dgrad + kDirectPackFloats + kWino1PackFloats : nullptr, (hipStream_t)stream))
        return rc;
    return fdn_pack_conv64_wino44_launch(w, wp_fwd, wp_dgrad, (hipStream_t)stream);
}

extern "C" int fdn_pack_conv64_weights_batch_streams(const float* w_base, const int64_t* w_offsets, int n_layers, float* packs,
                                                     int streams_fwd, int streams_dgrad, void* stream) {
    FDN_REQUIRE(w_base && w_offsets && packs && n_layers > 0, "fdn_pack_conv64_weights_batch: NULL argument or n_layers<=0");
    FDN_REQUIRE(!(streams_fwd & ~FDN_PACK_STREAM_ALL) && !(streams_dgrad & ~FDN_PACK_STREAM_ALL), "fdn_pack_conv64_weights_batch_streams: bad stream mask %d / %d",
                streams_fwd, streams_dgrad);
    if (!(streams_fwd | streams_dgrad)) return FDN_OK;
    if ((streams_fwd | streams_dgrad) & 7) {
        hipLaunchKernelGGL(pack_conv64_batch_kernel, dim3((153 * 64 * 64 + 255) / 256, (unsigned)n_layers), dim3(256), 0, (hipStream_t)stream,
                           w_base, w_offsets, packs, streams_fwd, streams_dgrad);
        FDN_CHECK_LAUNCH("fdn_pack_conv64_weights_batch");
    }
    if ((streams_fwd | streams_dgrad) & 24) {
        hipLaunchKernelGGL(pack_conv64_wino44_kernel, dim3((2 * 3 * 64 * 64 + 255) / 256, (unsigned)n_layers), dim3(256), 0, (hipStream_t)stream,
                           w_base, w_offsets, packs, (float*)nullptr, streams_fwd, streams_dgrad);
        FDN_CHECK_LAUNCH("pack_conv64_wino44_kernel");
    }
    return FDN_OK;
}

extern "C" int fdn_pack_conv64_weights_batch(const float* w_base, const int64_t* w_offsets, int n_layers, float* packs, void* stream) {
    return fdn_pack_conv64_weights_batch_streams(w_base, w_offsets, n_layers, packs, FDN_PACK_STREAM_ALL, FDN_PACK_STREAM_ALL, stream);
}

extern "C" int fdn_conv64_pack_streams(int N, int D, int H, int W, int algo, int role) {
    FDN_REQUIRE(algo >= FDN_ALGO_AUTO && algo <= FDN_ALGO_LAST, "fdn_conv64_pack_streams: bad algo %d", algo);
    FDN_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && D <= 1020 && H <= 1020 && W <= 1020, "fdn_conv64_pack_streams: bad dims");
    FDN_REQUIRE(role >= FDN_ROLE_FWD && role <= FDN_ROLE_DGRAD_FUSED, "fdn_conv64_pack_streams: bad role %d", role);
    unsigned m = 0;
    float* const some = reinterpret_cast<float*>(sizeof(float));      // a fused fold is requested by a non-NULL dz_prev; nothing is dereferenced
    int rc;
    if (role == FDN_ROLE_FWD)
        rc = fdn_conv64_launch_ex(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, N, D, H, W, D, H, W, 0, 0, FDN_ACT_NONE, 0.f,
                                  nullptr, 3, algo, &m);
    else
        rc = fdn_conv64_launch_ex(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, role == FDN_ROLE_DGRAD_FUSED ? some : nullptr, N, D, H, W,
                                  D + 2, H + 2, W + 2, -1, 1, FDN_ACT_NONE, 0.f, nullptr, 3, algo, &m);
    return rc ? rc : (int)m;
}

// 1: the forward and the fused dgrad of a 64->64 layer on this grid both run the plain F(4,3) x F(4,3) fp32-MFMA kernels, which write / read
// sign masks (fdn_conv64_fwd_mask / fdn_conv64_dgrad_fused_mask); 0: they do not (the caller keeps y); < 0: error.  The launcher's own selection.
extern "C" int fdn_conv64_mask_ok(int N, int D, int H, int W, int algo) {
    const int f = fdn_conv64_pack_streams(N, D, H, W, algo, FDN_ROLE_FWD);
    if (f < 0) return f;
    const int d = fdn_conv64_pack_streams(N, D, H, W, algo, FDN_ROLE_DGRAD_FUSED);
    if (d < 0) return d;
    return f == FDN_PACK_STREAM_WINO_H4 && d == (FDN_PACK_STREAM_WINO_H4 | FDN_PACK_STREAM_WINO_W) ? 1 : 0;
}

// --------------------------------------------------------------------------------------------
// host side: tile planning + launch
// --------------------------------------------------------------------------------------------
namespace {

// an output box of the (OD,OH,OW) grid and the tap range that is non-zero for it
struct Box { int od, oh, ow, ed, eh, ew, ta0, ta1, tb0, tb1, tc0, tc1; };
struct Plan { FdnTile t; double cost; };

// estimated time of a launch in units of "one M row through all 27x64 K steps on one CU".  Fitted to tools/bench_kernels.py
// on MI355X after the K loop got its fragment ring: per-tile staging/epilogue is hidden by the co-resident workgroups, a
// launch costs (tiles on the busiest CU) x (rows per tile) x f + a fixed latency that grows with the tile (first staging and
// last epilogue of a CU have nothing to overlap with); f = relative MFMA efficiency of the wave layout.
double plan_cost(const FdnTile& t, int N, int mcap, int cs, const Box& bx, double f) {
    const double tiles = (double)N * t.ntd * t.nth * t.ntw;
    const double rows = (double)(t.td + bx.ta1 - bx.ta0) * (t.th + bx.tb1 - bx.tb0) * (t.tw + bx.tc1 - bx.tc0);
    const double tapfrac = (bx.ta1 - bx.ta0 + 1) * (bx.tb1 - bx.tb0 + 1) * (bx.tc1 - bx.tc0 + 1) / 27.0;
    const double per_tile = mcap * tapfrac * f + 0.005 * rows + 0.5 * cs;
    const double per_cu_max = (double)((long long)((tiles + 255) / 256));
    const double per_cu_avg = tiles / 256.0;
    // the launch ends when the busiest CU does; the average only breaks ties
    return (0.95 * per_cu_max + 0.05 * per_cu_avg) * per_tile + 20.0 + 0.18 * mcap;
}

Plan best_plan(int N, const Box& bx, int mcap, int max_rows, int cs, double f) {
    Plan best{{1, 1, 1, bx.ed, bx.eh, bx.ew}, 1e30};
    const int da = bx.ta1 - bx.ta0, db = bx.tb1 - bx.tb0, dc = bx.tc1 - bx.tc0;
    for (int td = 1; td <= bx.ed && td <= mcap; ++td)
        for (int th = 1; th <= bx.eh && td * th <= mcap; ++th)
            for (int tw = 1; tw <= bx.ew && td * th * tw <= mcap; ++tw) {
                if ((td + da) * (th + db) * (tw + dc) > max_rows) continue;
                FdnTile t{td, th, tw, (bx.ed + td - 1) / td, (bx.eh + th - 1) / th, (bx.ew + tw - 1) / tw};
                const double c = plan_cost(t, N, mcap, cs, bx, f);
                if (c < best.cost) best = {t, c};
            }
    return best;
}

template <int MT, int NW, int CS>
Plan plan_for(int N, const Box& bx) {
    using C = Conv64Cfg<MT, NW, CS>;
    // measured at 48^3 N=8 (same work per CU): <1,1> 1.463 ms, <1,2> 1.506, <2,1> 1.542 per 3456..3584 row units; cs4 +2 %
    const double f = (MT == 2 ? 1.0 : (NW == 1 ? 0.985 : 1.015)) + (CS == 4 ? 0.02 : 0.0);
    return best_plan(N, bx, C::MCAP, C::MAXROWS, CS, f);
}

template <int MT, int NW, int CS>
int launch_conv64(Conv64Args& a, const Box* boxes, int nbox, hipStream_t s) {
    using C = Conv64Cfg<MT, NW, CS>;
    int first = 0, max_rows = 0;
    a.nreg = 0;
    for (int i = 0; i < nbox; ++i) {
        const Box& bx = boxes[i];
        if (bx.ed <= 0 || bx.eh <= 0 || bx.ew <= 0) continue;
        const FdnTile t = plan_for<MT, NW, CS>(a.N, bx).t;
        Conv64Region& r = a.reg[a.nreg++];
        r.first_block = first;
        r.obd = bx.od; r.obh = bx.oh; r.obw = bx.ow; r.ebd = bx.ed; r.ebh = bx.eh; r.ebw = bx.ew;
        r.ta0 = bx.ta0; r.ta1 = bx.ta1; r.tb0 = bx.tb0; r.tb1 = bx.tb1; r.tc0 = bx.tc0; r.tc1 = bx.tc1;
        r.td = t.td; r.th = t.th; r.tw = t.tw; r.ntd = t.ntd; r.nth = t.nth; r.ntw = t.ntw;
        r.hh = t.th + (bx.tb1 - bx.tb0); r.hw = t.tw + (bx.tc1 - bx.tc0);
        r.rows = (t.td + (bx.ta1 - bx.ta0)) * r.hh * r.hw;
        r.mg_hhhw = fdn_magic20(r.hh * r.hw);
        r.mg_hw = fdn_magic20(r.hw);
        r.mg_thtw = fdn_magic20(t.th * t.tw);
        r.mg_tw = fdn_magic20(t.tw);
        fdn_magic40(t.ntd * t.nth * t.ntw, &r.mg_tpn_hi, &r.mg_tpn_lo);
        fdn_magic40(t.nth * t.ntw, &r.mg_thw_hi, &r.mg_thw_lo);
        fdn_magic40(t.ntw, &r.mg_ntw_hi, &r.mg_ntw_lo);
        first += a.N * t.ntd * t.nth * t.ntw;
        if (r.rows > max_rows) max_rows = r.rows;
    }
    if (a.nreg == 0) return FDN_OK;
    if (int rc = fdn_func_max_lds((const void*)conv64_mfma_kernel<MT, NW, CS, true>, C::LDS_BYTES, "conv64")) return rc;
    if (int rc = fdn_func_max_lds((const void*)conv64_mfma_kernel<MT, NW, CS, false>, C::LDS_BYTES, "conv64")) return rc;
    const size_t lds = (size_t)max_rows * C::LROW + C::MCAP * 4;
    const Conv64Region& r0 = a.reg[0];
    const bool simple = a.nreg == 1 && r0.ta0 == 0 && r0.ta1 == 2 && r0.tb0 == 0 && r0.tb1 == 2 && r0.tc0 == 0 && r0.tc1 == 2;
    if (simple) hipLaunchKernelGGL((conv64_mfma_kernel<MT, NW, CS, false>), dim3((unsigned)first), dim3(256), lds, s, a);
    else hipLaunchKernelGGL((conv64_mfma_kernel<MT, NW, CS, true>), dim3((unsigned)first), dim3(256), lds, s, a);
    FDN_CHECK_LAUNCH("conv64_mfma_kernel");
    return FDN_OK;
}

// pick the variant from the first (dominant) box, then launch all boxes as regions of ONE launch
int launch_boxes(Conv64Args& a, const Box* boxes, int nbox, hipStream_t s) {
    const Box& bx = boxes[0];
    // variant table: 1:<2,1,2> 2:<2,1,4> 3:<1,1,2> 4:<1,1,4> 5:<1,2,2> 6:<1,2,4>
    const double c[7] = {1e30, plan_for<2, 1, 2>(a.N, bx).cost, plan_for<2, 1, 4>(a.N, bx).cost, plan_for<1, 1, 2>(a.N, bx).cost,
                         1e30, plan_for<1, 2, 2>(a.N, bx).cost, 1e30};
    int v = fdn_conv64_force_layout;
    if (v < 1 || v > 6) {
        // auto: cheapest of the cs2 variants (tools/bench_kernels.py: <1,1,cs2> wins at 48^3 N=8 -- 6912 tiles = 27 per CU
        // exactly -- <1,2,cs2> at 24^3; the cs4 variants are never ahead)
        v = 1;
        if (c[3] < c[v]) v = 3;
        if (c[5] < c[v]) v = 5;
    }
    switch (v) {
        case 1: return launch_conv64<2, 1, 2>(a, boxes, nbox, s);
        case 3: return launch_conv64<1, 1, 2>(a, boxes, nbox, s);
#ifdef FDN_TEST_HOOKS                                   // the cs4 layouts are never the planner's choice: test build only (forced layouts)
        case 2: return launch_conv64<2, 1, 4>(a, boxes, nbox, s);
        case 4: return launch_conv64<1, 1, 4>(a, boxes, nbox, s);
        case 6: return launch_conv64<1, 2, 4>(a, boxes, nbox, s);
#endif
        default: return launch_conv64<1, 2, 2>(a, boxes, nbox, s);
    }
}

}  // namespace

FdnTile fdn_plan_tile(int N, int OD, int OH, int OW, int max_vox, int max_halo_rows, int) {
    return best_plan(N, Box{0, 0, 0, OD, OH, OW, 0, 2, 0, 2, 0, 2}, max_vox, max_halo_rows, 2, 1.0).t;
}

int fdn_conv64_launch_ex(const float* x, const float* wpack, const float* bias, const float* residual, float* y,
                         const float* fskip, const float* fy, float* fout, int N, int ID, int IH, int IW, int OD, int OH,
                         int OW, int off, int zero_mode, int act, float alpha, hipStream_t s, int parts, int algo, unsigned* probe,
                         uint16_t* ymask, const uint16_t* fmask, const FdnExtraSrc* extra) {
    // ymask / fmask (sign masks, conv64_wino2d_kernel.h): only the plain F(4,3) x F(4,3) fp32-MFMA paths write / read them -- the forward
    // over the whole grid, the fused dgrad as ONE launch; every other path refuses (the caller asks fdn_conv64_mask_ok first)
    // extra (further sources of a fused dgrad, fdn_conv64_dgrad_fused_multi): the same rule -- the one-launch F(4,3) x F(4,3) / F(2,3) x F(4,3) form only
    auto no_mask = [&](const char* what) {
        if (extra && extra->nsrc > 1) {
            fdn_set_error("conv64: a multi-source fused dgrad is not supported on this path (%s): ask fdn_conv64_mask_ok", what);
            return FDN_ERR_UNSUPPORTED;
        }
        if (!ymask && !fmask) return FDN_OK;
        fdn_set_error("conv64: sign masks are not supported on this path (%s): ask fdn_conv64_mask_ok", what);
        return FDN_ERR_UNSUPPORTED;
    };
    // probe != nullptr: launch nothing, OR into *probe the streams of the pack this call would read (bit 0 direct, 1 1-D Winograd,
    // 2 F(2,3)xF(4,3), 3 F(4,3)xF(4,3), 4 the same as bf16 x 3) -- fdn_conv64_pack_streams; the selection below is the only statement of the rule.
    // staged rows are addressed with 32-bit byte offsets from the sample's first voxel (256 B per voxel)
    FDN_REQUIRE((long long)ID * IH * IW < (1ll << 24), "conv64: a sample of %dx%dx%d voxels exceeds the 32-bit row addressing", ID, IH, IW);
    Conv64Args a;
    a.x = x; a.wp = wpack; a.bias = bias; a.res = residual; a.y = y;
    a.fskip = fskip; a.fy = fy; a.fout = fout;
    a.N = N; a.ID = ID; a.IH = IH; a.IW = IW; a.OD = OD; a.OH = OH; a.OW = OW;
    a.off = off; a.zero_mode = zero_mode; a.act = act; a.alpha = alpha; a.dbg = fdn_conv64_dbg;
    const Box full{0, 0, 0, OD, OH, OW, 0, 2, 0, 2, 0, 2};
    // Winograd F(4,3) along W (conv64_wino.hip) whenever the W extent is a multiple of 4: half the MFMA work.  FDN_ALGO_DIRECT
    // (per call) or a forced direct layout (test build) selects the direct kernel below.
    const bool wino = algo != FDN_ALGO_DIRECT && (fdn_conv64_force_layout == 0 || fdn_conv64_force_layout == 7);
    const float* upack = wpack + kDirectPackFloats;
    const float* upack2 = upack + kWino1PackFloats;
    // 2-D Winograd (conv64_wino2d.hip): F(4,3) along H on top of F(4,3) along W (6.75 of the 27 tap-equivalents) when H is a multiple
    // of 4, F(2,3) along H (9 tap-equivalents; FDN_ALGO_WINO_H2 forces it) when H is even
    const bool wino2 = wino && algo != FDN_ALGO_WINO_W && fdn_conv64_force_layout == 0;
    auto hm_for = [&](int eh, int ew) {                           // output rows per cell of the 2-D kernel for an (eh, ew) box, 0 = not applicable
        if (!wino2) return 0;
        if (algo != FDN_ALGO_WINO_H2 && fdn_conv64_wino2d_ok(1, eh, ew, ID, IH, IW, 4)) return 4;
        return fdn_conv64_wino2d_ok(1, eh, ew, ID, IH, IW, 2) ? 2 : 0;
    };
    // FDN_ALGO_WINO_BF16X3: F(4,3) x F(4,3) grids read the bf16 x 3 stream and run the SPLIT kernel (hm is passed on as 4 | 8)
    const bool split = algo == FDN_ALGO_WINO_BF16X3;
    auto upack_hm = [&](int hm) { return hm == 4 ? (split ? upack2 + kWino2PackFloats + kWino44PackFloats : upack2 + kWino2PackFloats) : upack2; };
    auto stream_bit = [&](int hm) { return hm == 4 ? (split ? 16u : 8u) : 4u; };
    auto hm_arg = [&](int hm) { return hm == 4 && split ? 12 : hm; };
    // Grids off the multiple-of-4 raster (patch sizes 10, 18, 22 are legal: README.md:83 of the reference): the F(4,3) x F(4,3) kernel takes
    // the largest (h, w)-aligned box -- eh4 x ew4 of eh x ew, when that is most of the grid -- and the direct kernel the one or two
    // remainder strips beside it (w in [ew4, ew) over all h; h in [eh4, eh) over the aligned w range) as regions of ONE more launch.
    // Round 6 (tools/bench_small_grids.py, profiles/r6_small_grids.txt): at 8 x 22^3 the all-direct forward costs 0.186 ms, 2.4x the 24^3 grid's.
    auto split_box = [&](int eh, int ew, int& eh4, int& ew4) {
        eh4 = eh & ~3; ew4 = ew & ~3;
        // (W % 4 == 0 grids have Winograd paths of their own; below ~24 K voxels the second launch costs more than the strips' multiplies:
        // 8 x 10^3: forward 0.037 ms all direct, 0.071 split)
        return wino2 && algo != FDN_ALGO_WINO_H2 && ew4 != ew && eh4 >= 4 && ew4 >= 4 && 2 * eh4 * ew4 >= eh * ew && (long long)N * ID * IH * IW >= 24576 &&
               fdn_conv64_wino2d_ok(1, eh4, ew4, ID, IH, IW, 4);
    };
    if (!(fout && zero_mode && off == -1 && fdn_conv64_shell_slabs)) {
        if (const int hm = fout ? 0 : hm_for(OH, OW)) {           // (a fused fold outside the slab path is the 1-D / direct kernels' business)
            if (probe) { *probe |= stream_bit(hm); return FDN_OK; }
            if (hm != 4 || split) if (int rc = no_mask("not the fp32 F(4,3) x F(4,3) forward")) return rc;
            return fdn_conv64_wino2d_launch(x, upack_hm(hm), bias, residual, y, nullptr, nullptr, nullptr, N, ID, IH, IW, OD, OH, OW, 0, 0, 0,
                                            OD, OH, OW, off, zero_mode, act, alpha, hm_arg(hm), s, ymask, nullptr);
        }
        if (!probe) if (int rc = no_mask("forward off the F(4,3) x F(4,3) kernel")) return rc;
        int oh4, ow4;
        if (!fout && split_box(OH, OW, oh4, ow4)) {
            if (probe) { *probe |= stream_bit(4) | 1u; return FDN_OK; }
            if (int rc = no_mask("aligned box + strips")) return rc;
            if (int rc = fdn_conv64_wino2d_launch(x, upack_hm(4), bias, residual, y, nullptr, nullptr, nullptr, N, ID, IH, IW, OD, OH, OW, 0, 0, 0,
                                                  OD, oh4, ow4, off, zero_mode, act, alpha, hm_arg(4), s))
                return rc;
            const Box strips[2] = {{0, 0, ow4, OD, OH, OW - ow4, 0, 2, 0, 2, 0, 2}, {0, oh4, 0, OD, OH - oh4, ow4, 0, 2, 0, 2, 0, 2}};
            const int first = OW > ow4 ? 0 : 1, count = (OW > ow4 ? 1 : 0) + (OH > oh4 ? 1 : 0);
            return launch_boxes(a, strips + first, count, s);
        }
        if (wino && fdn_conv64_wino_ok(OD, OH, OW)) {
            if (probe) { *probe |= 2u; return FDN_OK; }
            return fdn_conv64_wino_launch(x, upack, bias, residual, y, fskip, fy, fout, N, ID, IH, IW, OD, OH, OW, 0, 0, 0, OD, OH,
                                          OW, off, zero_mode, act, alpha, s);
        }
        if (probe) { *probe |= 1u; return FDN_OK; }
        return launch_boxes(a, &full, 1, s);
    }
    // Fused dgrad on the padded grid (OD = ID+2): padded index p <-> position P = p-1 reads dz[p - 2 + tap], zero outside.
    // The inner box p in [1,ID]^3 needs all 27 taps.  Every shell position has at least one coordinate at 0 or ID+1, where
    // only tap 2 (resp. tap 0) of that dimension can reach a real voxel: six disjoint 1-voxel slabs with 9 taps each
    // (and a 1-deep staging box in the slab's normal direction) replace the (ID+2)^3 - ID^3 extra positions of a plain
    // padded launch at a third of their MFMA work.  All seven regions go out as ONE launch.
    const Box boxes[7] = {
        {1, 1, 1, ID, IH, IW, 0, 2, 0, 2, 0, 2},
        {0, 0, 0, 1, OH, OW, 2, 2, 0, 2, 0, 2},       {ID + 1, 0, 0, 1, OH, OW, 0, 0, 0, 2, 0, 2},     // d faces, full (h,w)
        {1, 0, 0, ID, 1, OW, 0, 2, 2, 2, 0, 2},       {1, IH + 1, 0, ID, 1, OW, 0, 2, 0, 0, 0, 2},     // h faces, d inner
        {1, 1, 0, ID, IH, 1, 0, 2, 0, 2, 2, 2},       {1, 1, IW + 1, ID, IH, 1, 0, 2, 0, 2, 0, 0}};    // w faces, d,h inner
    // parts: bit 0 = the inner box (finishes the interior of dz_prev), bit 1 = the six shell slabs (padded scratch only).  The two
    // write disjoint positions, so a caller may issue them on different streams and join before the border fold.
    if (wino && fdn_conv64_wino_ok(ID, IH, IW)) {
        // ONE Winograd launch: the inner box (fused-fold epilogue) + the d and h faces of the shell restricted to the inner W
        // range (one depth resp. height tap each: a third of the work per tile, dispatched last, they fill the tail) + the pair of
        // w faces over the full (d,h) range as a region of its own (a single W tap = one Winograd coordinate per face: K loop over
        // xi in {0, 5}, no output transform; conv64_wino.hip).  Until round 3 the w faces were a separate launch of the direct kernel.
        const FdnWinoBox wb[6] = {
            {1, 1, 1, ID, IH, IW, 0, 2, 0, 2, 0},
            {0, 0, 1, 1, OH, IW, 2, 2, 0, 2, 0}, {ID + 1, 0, 1, 1, OH, IW, 0, 0, 0, 2, 0},        // d faces, full h
            {1, 0, 1, ID, 1, IW, 0, 2, 2, 2, 0}, {1, IH + 1, 1, ID, 1, IW, 0, 2, 0, 0, 0},       // h faces, d inner
            {0, 0, 0, OD, OH, 4, 0, 2, 0, 2, 1}};                                                // w faces, full (d,h)
        const Box wfaces[2] = {{0, 0, 0, OD, OH, 1, 0, 2, 0, 2, 2, 2}, {0, 0, IW + 1, OD, OH, 1, 0, 2, 0, 2, 0, 0}};
        const bool wface_direct = fdn_conv64_wface_direct != 0;       // test build: the round-2 path (direct-kernel slab launch)
        int first = (parts & 1) ? 0 : 1;
        int count = ((parts & 1) ? 1 : 0) + ((parts & 2) ? (wface_direct ? 4 : 5) : 0);
        if (const int hm = (parts & 1) ? hm_for(IH, IW) : 0) {
            // round 4: the inner box (all 27 taps, fused-fold epilogue; 95 % / 91 % of the positions at 48^3 / 24^3) on the 2-D Winograd
            // body, the shell faces on the 1-D body (their single depth / height / width tap has nothing to transform along that axis)
            if ((parts & 2) && !wface_direct && !fdn_conv64_split_dgrad && !(split && hm == 4)) {      // (bf16 x 3: the inner box is a persistent launch of its own)
                // both parts: ONE launch, the shell faces behind the inner box's workgroups (conv64_wino2d_shell_kernel, conv64_wino.hip)
                if (probe) { *probe |= stream_bit(hm) | 2u; return FDN_OK; }
                if (hm != 4) if (int rc = no_mask("not the fp32 F(4,3) x F(4,3) fused dgrad")) return rc;
                FdnWino2dPrepared inner;
                if (int rc = fdn_conv64_wino2d_prepare(x, upack_hm(hm), bias, residual, y, fskip, fy, fout, N, ID, IH, IW, OD, OH, OW, 1, 1, 1, ID,
                                                       IH, IW, off, zero_mode, act, alpha, hm_arg(hm), &inner, nullptr, fmask, extra))
                    return rc;
                return fdn_conv64_wino_launch_boxes(x, upack, bias, residual, y, fskip, fy, fout, N, ID, IH, IW, OD, OH, OW, wb + 1, count - 1,
                                                    off, zero_mode, act, alpha, s, &inner, extra);
            }
            if (!probe) if (int rc = no_mask("fused dgrad issued in parts")) return rc;
            if (probe) *probe |= stream_bit(hm);
            else if (int rc = fdn_conv64_wino2d_launch(x, upack_hm(hm), bias, residual, y, fskip, fy, fout, N, ID, IH, IW, OD, OH, OW, 1, 1, 1, ID, IH,
                                                       IW, off, zero_mode, act, alpha, hm_arg(hm), s))
                return rc;
            first = 1; count -= 1;
            if (count == 0) return FDN_OK;
        }
        if (!probe) if (int rc = no_mask("1-D Winograd fused dgrad")) return rc;
        if (probe) { *probe |= 2u | (((parts & 2) && wface_direct) ? 1u : 0u); return FDN_OK; }
        if (int rc = fdn_conv64_wino_launch_boxes(x, upack, bias, residual, y, fskip, fy, fout, N, ID, IH, IW, OD, OH, OW, wb + first,
                                                  count, off, zero_mode, act, alpha, s))
            return rc;
        return ((parts & 2) && wface_direct) ? launch_boxes(a, wfaces, 2, s) : FDN_OK;
    }
    if (!probe) if (int rc = no_mask("fused dgrad off the Winograd kernels")) return rc;
    int ih4, iw4;
    if (split_box(IH, IW, ih4, iw4)) {
        // the same split of the inner box [1, ID] x [1, IH] x [1, IW] (fused-fold epilogue in both kernels), the six shell slabs behind it
        if (probe) { *probe |= stream_bit(4) | 1u; return FDN_OK; }
        if (parts & 1) {
            if (int rc = fdn_conv64_wino2d_launch(x, upack_hm(4), bias, residual, y, fskip, fy, fout, N, ID, IH, IW, OD, OH, OW, 1, 1, 1, ID, ih4,
                                                  iw4, off, zero_mode, act, alpha, hm_arg(4), s))
                return rc;
            const Box strips[2] = {{1, 1, 1 + iw4, ID, IH, IW - iw4, 0, 2, 0, 2, 0, 2}, {1, 1 + ih4, 1, ID, IH - ih4, iw4, 0, 2, 0, 2, 0, 2}};
            const int first = IW > iw4 ? 0 : 1, count = (IW > iw4 ? 1 : 0) + (IH > ih4 ? 1 : 0);
            if (int rc = launch_boxes(a, strips + first, count, s)) return rc;
        }
        return (parts & 2) ? launch_boxes(a, boxes + 1, 6, s) : FDN_OK;
    }
    if (probe) { *probe |= 1u; return FDN_OK; }
    if (parts == 3) return launch_boxes(a, boxes, 7, s);
    if (parts & 1) return launch_boxes(a, boxes, 1, s);
    return (parts & 2) ? launch_boxes(a, boxes + 1, 6, s) : FDN_OK;
}

int fdn_conv64_launch(const float* x, const float* wpack, const float* bias, const float* residual, float* y, int N,
                      int ID, int IH, int IW, int OD, int OH, int OW, int off, int zero_mode, int act, float alpha,
                      hipStream_t s, int algo) {
    return fdn_conv64_launch_ex(x, wpack, bias, residual, y, nullptr, nullptr, nullptr, N, ID, IH, IW, OD, OH, OW, off,
                                zero_mode, act, alpha, s, 3, algo, nullptr);
}

int fdn_fold_halo_border_launch(const float* s0, const float* s1, const float* s2, int nsrc, const float* skip,
                                const float* yprev, int act, float alpha, float* out, int N, int D, int H, int W,
                                hipStream_t s) {
    const int ID = D > 2 ? D - 2 : 0, IH = H > 2 ? H - 2 : 0;
    const int64_t per_n = (int64_t)(D > 1 ? 2 : 1) * H * W + (int64_t)ID * (H > 1 ? 2 : 1) * W + (int64_t)ID * IH * (W > 1 ? 2 : 1);
    const int64_t total = (int64_t)N * per_n * 16;
    int64_t nb = (total + 255) / 256;
    if (nb < 1) nb = 1;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(fold_halo_border_kernel, dim3((unsigned)nb), dim3(256), 0, s, s0, s1, s2, nsrc, skip, yprev, act, alpha,
                       out, N, D, H, W);
    FDN_CHECK_LAUNCH("fold_halo_border_kernel");
    return FDN_OK;
}

#ifdef FDN_TEST_HOOKS
extern "C" int fdn_debug_set_conv64_mt(int layout) { fdn_conv64_force_layout = layout; return FDN_OK; }
extern "C" int fdn_debug_set_conv64_dbg(int bits) { fdn_conv64_dbg = bits; return FDN_OK; }
extern "C" int fdn_debug_set_conv64_shell_slabs(int on) { fdn_conv64_shell_slabs = on; return FDN_OK; }
extern "C" int fdn_debug_set_conv64_wface_direct(int on) { fdn_conv64_wface_direct = on; return FDN_OK; }
extern "C" int fdn_debug_set_conv64_split_dgrad(int on) { fdn_conv64_split_dgrad = on; return FDN_OK; }
#endif

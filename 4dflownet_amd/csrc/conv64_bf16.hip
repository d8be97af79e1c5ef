// conv3d 3x3x3, 64 -> 64 channels, bf16 activations / fp32 accumulation, NDHWC: implicit GEMM on
// v_mfma_f32_32x32x16_bf16 (BASELINE.json configs[3]: the bf16 variant of SR4DFlowNet.py:93-120 and its dgrad).
//
// At bf16 MFMA rate (32 cycles per 32x32x16, 16x the fp32 rate) an operand fragment (1 KB) has to feed several MFMAs or
// the LDS / L1 pipes become the bound, and the layer only has N = 64 output channels.  Hence, differently from the
// fp32 kernel (conv64_mfma.hip):
//   * MFMA orientation is C[cout][voxel] = W[cout][k] * X[k][voxel]: A = weights, B = voxels.  With the cout rows of the
//     packed weight stream permuted (sigma below) a lane ends up holding 16 CONSECUTIVE cout of ONE voxel -> the epilogue is
//     two 16-B stores (and two 16-B loads for the residual / skip / mask) per lane and 32x32 block, no row loop.
//   * A wave owns 32 cout x 32 (h,w) positions x MT consecutive d planes (MT accumulator blocks).  The voxel fragment of
//     staged plane p is the operand of output planes p, p-1, p-2 with the depth taps a = 0,1,2 ("depth slide"): per
//     (b, c, 16 cin) a wave reads MT+2 voxel fragments (LDS) and 3 weight fragments (L1/L2) for 3*MT MFMAs.
//     MT = 8: 10 + 3 KB per 24 MFMAs (768 cycles) per wave -> LDS 53 B/clk/CU (21 %), L1 16 B/clk/CU (25 %).
//   * Workgroup = 4 waves = 2 cout halves x 2 halves of a (th x tw <= 64)-position plane block; tile = MT x th x tw voxels
//     (8x8x8 on the large grids), two workgroups per CU.
//   * The input box + halo is staged in 4 slices of 16 cin (32 B per voxel row = exactly one MFMA K step) into TWO LDS
//     buffers: while the K loop runs on slice s, the global loads of slice s+1 are issued (step 0) and written to the other
//     buffer (step 2) -- 8 loads of 16 B per thread, so staging costs 32 transient VGPRs and one barrier per slice and only
//     the first slice of a tile is exposed.  Weight fragments are prefetched two (b,c) steps ahead into two register sets.
//   * LDS image (Conv64Region::hs, swz_*): conflict-free ds_read_b128 for every tap shift, SQ_LDS_BANK_CONFLICT = 0.
//   * Same region table / boundary rules / fused-fold epilogue as the fp32 kernel (conv64_args.h).
// Measured on MI355X (tools/bench_bf16.py, tools/mfma_peak_bf16.hip): a pure MFMA loop sustains 1.90 PFLOP/s on this part
// (clock drops to ~1.8 GHz under bf16 matrix load; nominal 2.5 PFLOP/s at 2.4 GHz).
#include "fdn_common.h"
#include "conv64_args.h"
#include <type_traits>
#include <stdio.h>
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

FDN_HOOK_VAR(int, fdn_conv64bf_force_mt, 0);    // test/bench hook: 0 = auto, 4 / 8 = force the variant, +16 = full-depth tiles only
FDN_HOOK_VAR(int, fdn_conv64bf_mode2, 1);       // test/bench hook: 0 = never the two-slice kernel (MODE 2)
FDN_HOOK_VAR(int, fdn_conv64bf_dbg, 0);         // ablation bits: 1 = weight stride 0, 2 = no XCD remap, 4 = staging loads from a cache-resident 32 KB, 8 = no epilogue, 16 = plan the shell slabs like stand-alone launches

template <int MT>
struct Conv64BfCfg {
    static constexpr int ROWB = 32;                       // bytes per staged voxel row and slice (16 cin bf16)
    static constexpr int MCAP = MT * 64;                  // mtab entries (plane-major, 64 per plane)
    static constexpr int NP = 8;                          // staging passes of 128 rows -> at most 1024 staged voxels
    static constexpr int MAXROWS = NP * 128;
    static constexpr int LDS_BUDGET = 160 * 1024 / 2 - 256;                   // two workgroups per CU
    static constexpr int MAXLROWS = (LDS_BUDGET - MCAP * 4) / (2 * ROWB) / 32 * 32;   // LDS rows per buffer, whole 1-KB pieces (32 rows)
    static constexpr int MAXLROWS2 = (LDS_BUDGET - MCAP * 4) / (2 * ROWB) / 16 * 16;  // MODE 2: ONE buffer of 64-B rows, whole 1-KB pieces (16 rows)
};

__device__ __forceinline__ bf16x8 ld_bf16x8(const void* p) { return __builtin_bit_cast(bf16x8, *(const u32x4*)p); }

// 16 fp32 -> 16 bf16 (round to nearest even), two 16-B stores
__device__ __forceinline__ void st_bf16x16(uint16_t* dst, const float (&z)[16]) {
    bf16x8 lo, hi;
#pragma unroll
    for (int r = 0; r < 8; ++r) { lo[r] = (__bf16)z[r]; hi[r] = (__bf16)z[8 + r]; }
    *(u32x4*)dst = __builtin_bit_cast(u32x4, lo);
    *(u32x4*)(dst + 8) = __builtin_bit_cast(u32x4, hi);
}
__device__ __forceinline__ void ld_bf16x16(const uint16_t* src, float (&z)[16]) {
    const bf16x8 lo = ld_bf16x8(src), hi = ld_bf16x8(src + 8);
#pragma unroll
    for (int r = 0; r < 8; ++r) { z[r] = (float)lo[r]; z[8 + r] = (float)hi[r]; }
}

// MODE 1 (FAST): every region of the launch has all 27 taps and td == MT (forward; the inner box of a fused dgrad): 9 unrolled
//               (b,c) steps per slice of 16 cin with register-prefetched weights and in-loop staging of the next slice (two LDS buffers).
// MODE 2 (round 4): the same regions with 8 x 8 plane blocks, staged in TWO slices of 32 cin into ONE buffer of 64-B rows.  The memory
//               system serves a 32-B piece of a bf16 row by fetching its whole 128-B line (tools/fetch_calib.hip), so four slices pull
//               every line four times; two slices pull it twice.  One buffer means a tile's staging is not hidden behind its own K loop
//               any more -- the co-resident workgroup's K loop runs meanwhile.
// MODE 0      : shell slabs / ragged tiles: rolled loop, predicated planes and taps.
// (a __device__ body + thin __global__ wrappers: conv64_bf16_fused_kernel below runs the MODE 2 body on the inner box of a fused dgrad
// and the MODE 0 body on its shell slabs in ONE launch.  block / total: this part's workgroup id and count; xcd_remap: the ids are the
// hardware's own (dealt round-robin to the XCDs), so the XCD-aware order applies.)
// MULTI: the multi-source fused dgrad (p.nsrc sources); false = one source, every source loop folds away at compile time (the single-source
// kernels are round 5's instruction for instruction: with a run-time source count the two-slice forward ran 7 % slower, r6 kernel traces)
template <int MT, int MODE, bool MULTI = false>
__device__ __forceinline__ void conv64_bf16_body(const Conv64BfArgs& p, const int block, const int total, const bool xcd_remap, char* const smem) {
    constexpr bool FAST = MODE != 0, S2 = MODE == 2;
    constexpr bool GEN = !FAST;
    using C = Conv64BfCfg<MT>;
    constexpr int ROWB = S2 ? 64 : C::ROWB, NP = C::NP;
    constexpr int SPR = ROWB / 16;                         // 16-B slots per LDS row

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int j = lane & 31;
    const int kh = lane >> 5;
    const int wn = wave & 1;      // cout half
    const int wm = wave >> 1;     // half of the plane block

    // ---- XCD-aware tile order: workgroup ids are dealt round-robin to the 8 XCDs; give each XCD one contiguous run of
    // tiles so that neighbouring tiles (shared halo voxels) meet in the same L2 ----
    int bid = block;
    if (xcd_remap && !(p.dbg & 2)) {
        const int q = total >> 3, rem = total & 7, x = bid & 7;
        bid = x * q + (x < rem ? x : rem) + (bid >> 3);
    }
    int ri = 0;
    while (ri + 1 < p.nreg && bid >= p.reg[ri + 1].first_block) ++ri;
    ri = __builtin_amdgcn_readfirstlane(ri);
    const Conv64Region R = p.reg[ri];
    const int ta0 = GEN ? R.ta0 : 0, tb0 = GEN ? R.tb0 : 0, tc0 = GEN ? R.tc0 : 0;
    const int na = GEN ? R.ta1 - R.ta0 + 1 : 3;          // depth taps: 3, or 1 on the d-face slabs
    const int nb = GEN ? R.tb1 - R.tb0 + 1 : 3, nc = GEN ? R.tc1 - R.tc0 + 1 : 3;
    const int tiles_per_n = R.ntd * R.nth * R.ntw;
    // (multiply-shift with host-made magics: a runtime integer division is ~25 VALU instructions, and everything outside the K
    // loop of this kernel is issue-bound work no co-resident wave can hide -- DESIGN.md, machine model)
    int b = bid - R.first_block;
    const int n = fdn_udiv40(b, R.mg_tpn_hi, R.mg_tpn_lo);
    b -= n * tiles_per_n;
    const int tdi = fdn_udiv40(b, R.mg_thw_hi, R.mg_thw_lo);
    b -= tdi * (R.nth * R.ntw);
    const int thi = fdn_udiv40(b, R.mg_ntw_hi, R.mg_ntw_lo);
    const int bufB = R.lrows_p * ROWB;
    int* mtab = (int*)(smem + (S2 ? 1 : 2) * bufB);
    const int p0d = R.obd + tdi * R.td, p0h = R.obh + thi * R.th, p0w = R.obw + (b - thi * R.ntw) * R.tw;
    const int prn = R.th * R.tw;                          // positions per plane block (<= 64)

    // ---- staging descriptors of this thread (the same for every slice) ----
    const int chunk = tid & 1;
    const int q0d = p0d - 1 + p.off + ta0, q0h = p0h - 1 + p.off + tb0, q0w = p0w - 1 + p.off + tc0;
    const int cstride = R.hh * R.hw;                      // staged voxels per plane
    // FAST: direct-to-LDS staging (buffer_load_dwordx4 ... lds).  The destination of such a load is M0 + lane * 16, i.e. the LDS image
    // has to be LANE-LINEAR: item i = pass * 256 + tid is LDS row i >> 1 (rows in padded order (zd, zh, zw' < hs)), 16-B slot i & 1;
    // the conflict-avoiding chunk swizzle moves to the SOURCE side (slot c receives channel chunk c ^ f).  Pad rows, rows past the
    // image and (dgrad) voxels outside the volume read past the buffer's range, which returns -- and stores -- zeros.  One
    // instruction per 1 KB, no data registers, no ds_write, no zero-select: the staging of a slice is NPL VMEM instructions per wave.
    // MODE 2: rows are dense (hs = hw) and hold four 16-B slots; slot s of a row with staged height index zh sits at physical slot
    // s ^ (zh & 3): the four 4-lane runs of a ds_read_b128 lane group lie on four consecutive zh (8 x 8 plane block), i.e. on four
    // different slots, and the four rows of a run on four different bank quads -- conflict-free for every tap shift.
    constexpr int NPL = S2 ? (C::MAXLROWS2 * 4 + 255) / 256 : (FAST ? (C::MAXLROWS * 2 + 255) / 256 : 1);
    unsigned goff[NPL];                                   // byte offset from the sample's first voxel, or kOob
    constexpr unsigned kOob = 0x80000000u;
    const int nitems = R.lrows_p * SPR;                     // buffers are padded to whole 1-KB pieces: no partial wave
    if (FAST) {
        const int hhhs = R.hh * R.hs;
#pragma unroll
        for (int u = 0; u < NPL; ++u) {
            const int i = u * 256 + tid;
            const int r = i / SPR, pslot = i % SPR;
            goff[u] = kOob;
            if (r < R.lrows) {
                const int zd = fdn_div20(r, R.mg_hhhs);
                const int r2 = r - zd * hhhs;
                const int zh = fdn_div20(r2, R.mg_hs);
                const int zw = r2 - zh * R.hs;
                const int f = ((zh >> R.swz_hs) + ((zw >> 2) & R.swz_wm)) & 1;
                int qd = q0d + zd, qh = q0h + zh, qw = q0w + zw;
                const bool inside = (unsigned)qd < (unsigned)p.ID && (unsigned)qh < (unsigned)p.IH && (unsigned)qw < (unsigned)p.IW;
                qd = min(max(qd, 0), p.ID - 1);
                qh = min(max(qh, 0), p.IH - 1);
                qw = min(max(qw, 0), p.IW - 1);
                if (zw < R.hw && (inside || !p.zero_mode)) {
                    unsigned v = (unsigned)((qd * p.IH + qh) * p.IW + qw);
                    if (p.dbg & 4) v = (unsigned)(r & 255);  // ablation: real data, but always the same 32 KB (cache hits)
                    goff[u] = v * 128u + (unsigned)((S2 ? (pslot ^ (zh & 3)) : (chunk ^ f)) << 4);
                    if (S2 && (p.dbg & 64)) goff[u] = v * 64u + (unsigned)((pslot ^ (zh & 3)) << 4);   // timing prototype: planar halves [2][N][DHW][32]
                }
            }
        }
    }
    const bool planar = S2 && (p.dbg & 64);
    const unsigned half_bytes = (unsigned)(p.N * p.ID * p.IH * p.IW) * 64u;
    // multi-source fused dgrad (MODE 2 / MODE 0 of the one-launch form): the rows / weights of source `src` (scalar selects)
    const int nsrc = MULTI ? p.nsrc : 1;
    // (loaded once, unconditionally: a conditional load per branch gets merged into ONE load through a selected ADDRESS, which forces the
    // whole argument block into scratch memory)
    const uint16_t* const xs0 = p.x; const uint16_t* const xs1 = p.x1; const uint16_t* const xs2 = p.x2;
    const uint16_t* const ws0 = p.wp; const uint16_t* const ws1 = p.wp1; const uint16_t* const ws2 = p.wp2;
    auto x_of = [&](int src) { return (!MULTI || src == 0) ? xs0 : (src == 1 ? xs1 : xs2); };
    auto wp_of = [&](int src) { return (!MULTI || src == 0) ? ws0 : (src == 1 ? ws1 : ws2); };
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    auto stage_dma = [&](char* buf, int sl, int u0, int u1, int src = 0) {
        const __amdgpu_buffer_rsrc_t xrsrc = planar
            ? __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)n * p.ID * p.IH * p.IW * 32), 0, half_bytes + (unsigned)(p.ID * p.IH * p.IW) * 64u, 0x00020000)
            : __builtin_amdgcn_make_buffer_rsrc((void*)(x_of(src) + (size_t)n * p.ID * p.IH * p.IW * 64), 0, (unsigned)(p.ID * p.IH * p.IW) * 128u, 0x00020000);
#pragma unroll
        for (int u = 0; u < NPL; ++u) {
            if (u < u0 || u >= u1) continue;
            if (u * 256 + wave_u * 64 >= nitems) continue;            // wave-uniform: the whole 1-KB piece lies past the (padded) image
            fdn_lds_dma16(xrsrc, buf + (u * 256 + wave_u * 64) * 16, goff[u], planar ? sl * (int)half_bytes : sl * ROWB);
        }
    };
    // GEN: register staging -- NP staged voxels x one 16-B chunk per thread, written to a padded / swizzled image by ds_write
    int gv[GEN ? NP : 1];                                 // input voxel index (clamped), sign bit set = store zeros
    int lo[GEN ? NP : 1];                                 // LDS byte offset inside a buffer, -1 = nothing to stage
    if (GEN) {
        const int rows_eff = R.rows;
        const int in_n = n * p.ID * p.IH * p.IW;
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int r = u * 128 + (tid >> 1);
            gv[u] = 0; lo[u] = -1;
            if (r < rows_eff) {
                const int zd = fdn_div20(r, R.mg_hhhw);
                const int r2 = r - zd * cstride;
                const int zh = fdn_div20(r2, R.mg_hw);
                const int zw = r2 - zh * R.hw;
                const int f = ((zh >> R.swz_hs) + ((zw >> 2) & R.swz_wm)) & 1;
                lo[u] = ((zd * R.hh + zh) * R.hs + zw) * ROWB + ((chunk ^ f) << 4);
                int qd = q0d + zd, qh = q0h + zh, qw = q0w + zw;
                if (GEN && R.swap_dh) { const int t_ = qd; qd = qh; qh = t_; }           // kernel axes (h, d) -> real (d, h)
                const bool inside = (unsigned)qd < (unsigned)p.ID && (unsigned)qh < (unsigned)p.IH && (unsigned)qw < (unsigned)p.IW;
                qd = min(max(qd, 0), p.ID - 1);
                qh = min(max(qh, 0), p.IH - 1);
                qw = min(max(qw, 0), p.IW - 1);
                gv[u] = in_n + (qd * p.IH + qh) * p.IW + qw;
                if (p.dbg & 4) gv[u] = r & 255;          // ablation: real data, but always the same 32 KB (cache hits)
                if (p.zero_mode && !inside) gv[u] |= (int)0x80000000;
            }
        }
    }
    const int xchunk_off = chunk * 8;
    // staged in two halves of NP/2 voxels that share four registers (half 0: loaded at step 0, written at step 3; half 1: steps 4, 7)
    constexpr int NH = NP / 2;
    u32x4 sv[GEN ? NH : 1];
    auto stage_load = [&](int vs, int half) {       // unconditional loads (clamped address), zero selection afterwards; vs = 4 * source + slice
        if (!GEN) return;
        const uint16_t* xchunk = x_of(vs >> 2) + xchunk_off;
#pragma unroll
        for (int u = 0; u < NH; ++u) sv[u] = *(const u32x4*)(xchunk + (size_t)(gv[half * NH + u] & 0x7fffffff) * 64 + (vs & 3) * 16);
    };
    auto stage_write = [&](char* buf, int half) {
        if (!GEN) return;
#pragma unroll
        for (int u = 0; u < NH; ++u) {
            if (gv[half * NH + u] < 0) sv[u] = (u32x4){0u, 0u, 0u, 0u};
            if (lo[half * NH + u] >= 0) *(u32x4*)(buf + lo[half * NH + u]) = sv[u];
        }
    };
    if (GEN) stage_load(0, 0);
    else stage_dma(smem, 0, 0, NPL);

    // ---- weight fragments: stream [slice 4][b*3+c][a][kh][cout row 64] x 16 B, two register sets, two steps ahead ----
    const int wstride = (p.dbg & 1) ? 0 : 1;
    const int wlane = kh * 64 + wn * 32 + j;
    u32x4 wq[3][3];                       // three register sets: weights THREE steps ahead (see fast_slice)
    auto load_w = [&](u32x4 (&dst)[3], int sl, int tb, int tc, int src = 0) {
        const u32x4* wbase = (const u32x4*)wp_of(src) + wlane;
        const int bc = tb * 3 + tc;
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (GEN && a >= na) continue;
            // stream index = ((slice * 9 + kh * 3 + kw) * 3 + kd); a swapped region slides along h: its "depth" tap is kh, its tb is kd
            const int idx = (GEN && R.swap_dh) ? (((sl & 3) * 9 + (ta0 + a) * 3 + tc) * 3 + tb) : (((sl & 3) * 9 + bc) * 3 + ta0 + a);
            dst[a] = wbase[(idx * 128) * wstride];
        }
    };
    // step `it` of a slice -> tap (tb0 + it / nc, tc0 + it % nc); nb * nc is 9 or 3
    load_w(wq[0], 0, tb0, tc0);
    load_w(wq[1], 0, tb0 + (nc == 1 ? 1 : 0), tc0 + (nc == 1 ? 0 : 1));
    load_w(wq[2], 0, tb0 + (nc == 1 ? 2 : 0), tc0 + (nc == 1 ? 0 : 2));

    // ---- output voxel of each (plane, position): output-grid voxel index, tagged (bit 30) if finished by the fused fold ----
    for (int m = tid; m < C::MCAP; m += 256) {
        int g = -1;
        const int md = m >> 6, pr = m & 63;
        if (md < R.td && pr < prn) {
            const int mh = fdn_div20(pr, R.mg_tw);
            int pd = p0d + md, ph = p0h + mh;
            const int pw = p0w + (pr - mh * R.tw);
            if (pd < R.obd + R.ebd && ph < R.obh + R.ebh && pw < R.obw + R.ebw) {
                if (GEN && R.swap_dh) { const int t_ = pd; pd = ph; ph = t_; }            // kernel axes (h, d) -> real (d, h)
                g = ((n * p.OD + pd) * p.OH + ph) * p.OW + pw;
                if (p.fout) {
                    const int id = pd - 1, ih = ph - 1, iw = pw - 1;
                    if (id >= 1 && id <= p.ID - 2 && ih >= 1 && ih <= p.IH - 2 && iw >= 1 && iw <= p.IW - 2)
                        g = (((n * p.ID + id) * p.IH + ih) * p.IW + iw) | (1 << 30);
                }
            }
        }
        mtab[m] = g;
    }

    // ---- this lane's voxel inside a staged plane ----
    int mh0, mw0;
    {
        int pr = wm * 32 + j;
        pr = pr < prn ? pr : prn - 1;
        mh0 = fdn_div20(pr, R.mg_tw);
        mw0 = pr - mh0 * R.tw;
    }
    const int lrow0 = mh0 * R.hs + mw0;
    const int pstrideB = R.hh * R.hs * ROWB;              // LDS bytes per staged plane
    const int npl = R.td + na - 1;                        // staged planes

    // accumulators start from the bias (lane (j,kh) holds cout [wn*32 + kh*16, +16) of its voxels): no add in the epilogue
    f32x16 acc[MT];
    {
        f32x16 bv16;
#pragma unroll
        for (int r = 0; r < 16; ++r) bv16[r] = 0.f;
        if (p.bias) {
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                const f32x4 t = *(const f32x4*)(p.bias + wn * 32 + kh * 16 + r);
                bv16[r] = t.x; bv16[r + 1] = t.y; bv16[r + 2] = t.z; bv16[r + 3] = t.w;
            }
        }
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) acc[mi] = bv16;
    }

    // one (b,c) step on buffer `buf`: FULL = every plane and depth tap present (no predicates)
    auto kstep = [&](auto fullc, const char* buf, int db, int dc, const u32x4 (&wsrc)[3], int k2 = 0) {
        constexpr bool FULL = decltype(fullc)::value;
        bf16x8 wv[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) wv[a] = __builtin_bit_cast(bf16x8, wsrc[a]);
        const int f = (((mh0 + db) >> R.swz_hs) + (((mw0 + dc) >> 2) & R.swz_wm)) & 1;
        const char* lp = buf + (lrow0 + db * R.hs + dc) * ROWB + (S2 ? (((k2 * 2 + kh) ^ ((mh0 + db) & 3)) << 4) : ((kh ^ f) << 4));
#pragma unroll
        for (int pl = 0; pl < MT + 2; ++pl) {
            if (!FULL && pl >= npl) continue;
            const bf16x8 xv = ld_bf16x8(lp + pl * pstrideB);
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const int md = pl - a;
                if (md < 0 || md >= MT) continue;
                if (!FULL && a >= na) continue;
                acc[md] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv[a], xv, acc[md], 0, 0, 0);
            }
        }
    };

    if (GEN) {
        stage_write(smem, 0);
        stage_load(0, 1); stage_write(smem, 1);
    }
    __syncthreads();
    if (S2) {
        // two slices of 32 cin = 18 steps each: the two 16-cin halves of a row one after the other, 9 (b,c) taps each -- the SAME
        // summation order as the four-slice kernel (cin group outermost), so the variants stay bit-identical; weights three steps
        // ahead across the slice boundary; the second slice replaces the first in the same buffer between two barriers
#pragma unroll 1
        for (int vs = 0; vs < 2 * nsrc; ++vs) {               // slice vs & 1 of source vs >> 1
#pragma unroll
            for (int it = 0; it < 18; ++it) {
                const int k2 = it / 9, bc = it % 9;
                kstep(std::true_type{}, smem, bc / 3, bc % 3, wq[it % 3], k2);
                // step it + 3 of the stream of 36 steps per source: 16-cin group n / 9, tap n % 9 (running into the next source's stream)
                const int nx = it + 3, vsx = vs + nx / 18, itx = nx % 18;
                if (vsx < 2 * nsrc) load_w(wq[it % 3], (vsx & 1) * 2 + itx / 9, tb0 + (itx % 9) / 3, tc0 + (itx % 9) % 3, vsx >> 1);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (vs + 1 < 2 * nsrc) {
                __syncthreads();                          // every wave is done reading this slice
                stage_dma(smem, (vs + 1) & 1, 0, NPL, (vs + 1) >> 1);
                __syncthreads();                          // (the compiler drains the LDS-DMA queue in front of the barrier)
            }
        }
    } else if (FAST) {
        // 9 unrolled steps per slice; weights two steps ahead (running into the next slice) in two register sets; the next
        // slice's voxels are loaded behind step 0's weight refill and written to the other buffer at step 7.  No branch
        // sits between a load and its use, so every wait is a counted vmcnt.  A slice has an odd number of steps, so the
        // register set of step `it` alternates from slice to slice (PAR); all four slices are straight-line code.
        auto fast_slice = [&](auto prefetchc, auto parc, int sl) {   // sl = 4 * source + slice
            constexpr bool PREFETCH = decltype(prefetchc)::value;
            (void)parc;
            const char* cur = smem + (sl & 1) * bufB;
            char* nxt = smem + ((sl + 1) & 1) * bufB;
            const int src_n = min((sl + 1) >> 2, nsrc - 1);      // source of the next slice (past the last one: a harmless re-read)
#pragma unroll
            for (int it = 0; it < 9; ++it) {
                // A wave's vector-memory operations return in order (vmcnt), so the wait for a weight fragment also waits for every
                // staging load issued before it: fragments three steps ahead and staging loads in two groups of four give those
                // loads ~70 MFMAs to come back before the first wait that covers them.
                kstep(std::true_type{}, cur, it / 3, it % 3, wq[it % 3]);
                if (PREFETCH || it < 6)
                    load_w(wq[it % 3], sl + (it + 3) / 9, tb0 + ((it + 3) % 9) / 3, tc0 + ((it + 3) % 9) % 3, it < 6 ? sl >> 2 : src_n);
                // the next slice goes straight from memory into the other buffer, in two bursts (VMEM operations return in order: a burst
                // sits in front of the weight fragments requested after it)
                if (PREFETCH && it == 0) stage_dma(nxt, (sl + 1) & 3, 0, NPL / 2, src_n);
                if (PREFETCH && it == 4) stage_dma(nxt, (sl + 1) & 3, NPL / 2, NPL, src_n);
                // keep every step's loads inside the step: under register pressure the scheduler otherwise sinks the
                // weight refills down to their first use, i.e. prefetch distance 0
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
        };
#pragma unroll 1
        for (int sl = 0; sl < 4 * nsrc; sl += 2) {
            fast_slice(std::true_type{}, std::integral_constant<int, 0>{}, sl);
            fast_slice(std::true_type{}, std::integral_constant<int, 1>{}, sl + 1);
        }
    } else {
#pragma unroll 1
        for (int vs = 0; vs < 4 * nsrc; ++vs) {              // slice vs & 3 of source vs >> 2
            const char* cur = smem + (vs & 1) * bufB;
            char* nxt = smem + ((vs + 1) & 1) * bufB;
            const bool more = vs + 1 < 4 * nsrc;
            // shell slabs / ragged tiles: rolled loop, predicated planes and taps, weights loaded in step
            if (more) { stage_load(vs + 1, 0); stage_write(nxt, 0); stage_load(vs + 1, 1); }
#pragma unroll 1
            for (int it = 0; it < nb * nc; ++it) {
                const int db = it / nc, dc = it - db * nc;
                u32x4 w3[3] = {};
                load_w(w3, vs & 3, tb0 + db, tc0 + dc, vs >> 2);
                kstep(std::false_type{}, cur, db, dc, w3);
            }
            if (more) stage_write(nxt, 1);
            __syncthreads();
        }
    }
    if (p.dbg & 8) return;

    // ---- epilogue: lane (j,kh) holds cout [wn*32 + kh*16, +16) of the voxel at position wm*32 + j of every plane ----
    const float slope = p.act == FDN_ACT_RELU ? 0.f : (p.act == FDN_ACT_LEAKY ? p.alpha : 1.f);
    const int cofs = wn * 32 + kh * 16;
    const bool act_max = slope <= 1.f;          // act(t) = max(t, slope*t) for slope in [0,1] (relu, leaky, none): 2 VALU, not 3
    // fused dgrad with a sign mask instead of y: the MT mask words of this lane (2 B each) are requested up front
    unsigned mk[MT];
    if (p.fout && p.fmask) {
#pragma unroll
        for (int md = 0; md < MT; ++md) {
            const int g = mtab[md * 64 + wm * 32 + j];
            mk[md] = (g >= 0 && (g & (1 << 30))) ? (unsigned)p.fmask[(size_t)(g & ~(1 << 30)) * 4 + (cofs >> 4)] : 0u;
        }
    }
#pragma unroll
    for (int md = 0; md < MT; ++md) {
        const int g = mtab[md * 64 + wm * 32 + j];
        if (g < 0) continue;
        float z[16];
        if (p.fout) {
            if (g & (1 << 30)) {
                const size_t o = (size_t)(g & ~(1 << 30)) * 64 + cofs;
                float sk[16], ym[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) { sk[r] = 0.f; ym[r] = 1.f; }
                if (p.fskip) ld_bf16x16(p.fskip + o, sk);
                if (p.fmask) {
                    // act' straight from the bit: sign-extend it to a lane mask and select between the bit patterns of 1 and slope (v_bfe_i32 + v_bfi_b32)
                    const unsigned oneb = 0x3f800000u, slb = __builtin_bit_cast(unsigned, slope);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned t = (unsigned)((int)(mk[md] << (31 - r)) >> 31);
                        z[r] = (acc[md][r] + sk[r]) * __builtin_bit_cast(float, (t & oneb) | (~t & slb));
                    }
                } else {
                    if (p.fy) ld_bf16x16(p.fy + o, ym);
#pragma unroll
                    for (int r = 0; r < 16; ++r) z[r] = (acc[md][r] + sk[r]) * (ym[r] > 0.f ? 1.f : slope);
                }
                st_bf16x16(p.fout + o, z);
            } else {
                float* o = p.ypad + (size_t)g * 64 + cofs;
#pragma unroll
                for (int r = 0; r < 16; r += 4)
                    *(f32x4*)(o + r) = (f32x4){acc[md][r], acc[md][r + 1], acc[md][r + 2], acc[md][r + 3]};
            }
        } else {
            const size_t o = (size_t)g * 64 + cofs;
            if (p.res) {
                float rv[16];
                ld_bf16x16(p.res + o, rv);
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = acc[md][r] + rv[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = acc[md][r];
            }
            if (act_max) {
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = fmaxf(z[r], slope * z[r]);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) z[r] = z[r] > 0.f ? z[r] : slope * z[r];
            }
            st_bf16x16(p.y + o, z);
            if (p.ymask) {                         // bit r = (the bf16 value just stored > 0): what the dgrad's act' asks of y
                unsigned bits = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) bits |= ((float)(__bf16)z[r] > 0.f ? 1u : 0u) << r;
                p.ymask[(size_t)g * 4 + (cofs >> 4)] = (uint16_t)bits;
            }
        }
    }
}

template <int MT, int MODE, bool MULTI = false>
__global__ __launch_bounds__(256, 2) void conv64_bf16_kernel(Conv64BfArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    conv64_bf16_body<MT, MODE, MULTI>(p, (int)blockIdx.x, (int)gridDim.x, true, smem);
}

// Fused dgrad as ONE launch (round 5, the fp32 path's conv64_wino2d_shell_kernel idea): workgroups [0, n2) run the MODE 2 body on the
// inner box, the rest the general body on the six shell slabs.  As a launch of their own the slabs -- 1.6 % of the inner box's work at
// 128^3 -- cost 0.126 ms on an almost empty chip; dispatched last they fill the inner launch's tail.
template <int MT, bool MULTI = false>
__global__ __launch_bounds__(256, 2) void conv64_bf16_fused_kernel(Conv64BfArgs p2, Conv64BfArgs p0, int n2) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < n2) conv64_bf16_body<MT, 2, MULTI>(p2, (int)blockIdx.x, n2, true, smem);
    else conv64_bf16_body<MT, 0, MULTI>(p0, (int)blockIdx.x - n2, (int)gridDim.x - n2, false, smem);
}

// --------------------------------------------------------------------------------------------
// border fold (bf16 tensors, fp32 scratch): surface voxels of dz_prev after a fused-fold dgrad launch
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fold_halo_border_bf16_kernel(const float* __restrict__ s0, const float* __restrict__ s1,
                                                                     const float* __restrict__ s2, int nsrc,
                                                                     const uint16_t* __restrict__ skip,
                                                                     const uint16_t* __restrict__ yprev, int act, float alpha,
                                                                     uint16_t* __restrict__ out, int N, int D, int H, int W) {
    const int ID = D > 2 ? D - 2 : 0, IH = H > 2 ? H - 2 : 0;
    const int nd_faces = (D > 1 ? 2 : 1) * H * W;
    const int nh_faces = ID * (H > 1 ? 2 : 1) * W;
    const int nw_faces = ID * IH * (W > 1 ? 2 : 1);
    const int per_n = nd_faces + nh_faces + nw_faces;
    const int64_t total = (int64_t)N * per_n * 8;
    const int PH = H + 2, PW = W + 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i & 7);
        int s = (int)((i >> 3) % per_n);
        const int n = (int)((i >> 3) / per_n);
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
        // all loads of a voxel are requested before the first use (see fold_halo_border_kernel, conv64_mfma.hip)
        const int64_t o = ((((int64_t)n * D + d) * H + h) * W + w) * 64 + c8 * 8;
        const u32x4 zero_u = {0u, 0u, 0u, 0u};
        const u32x4 skr = skip ? *(const u32x4*)(skip + o) : zero_u;
        const u32x4 ypr = yprev ? *(const u32x4*)(yprev + o) : zero_u;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
        if (D == 1 || H == 1 || W == 1) {          // an axis of extent 1 folds both padded neighbours onto the voxel: general nest
            for (int a = 0; a < nd; ++a)
                for (int b = 0; b < nh; ++b)
                    for (int c = 0; c < nw; ++c) {
                        const int64_t off = (((((int64_t)n * (D + 2) + pd[a]) * PH + ph[b]) * PW + pw[c]) * 16 + c8 * 2);
                        a0 += ((const f32x4*)s0)[off]; a1 += ((const f32x4*)s0)[off + 1];
                        if (nsrc > 1) { a0 += ((const f32x4*)s1)[off]; a1 += ((const f32x4*)s1)[off + 1]; }
                        if (nsrc > 2) { a0 += ((const f32x4*)s2)[off]; a1 += ((const f32x4*)s2)[off + 1]; }
                    }
        } else
        for (int sidx = 0; sidx < nsrc; ++sidx) {
            const f32x4* sp = (const f32x4*)(sidx == 0 ? s0 : (sidx == 1 ? s1 : s2));
            f32x4 v0[8], v1[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int a = k >> 2, b = (k >> 1) & 1, c = k & 1;
                v0[k] = v1[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (a < nd && b < nh && c < nw) {
                    const int64_t off = ((((int64_t)n * (D + 2) + pd[a]) * PH + ph[b]) * PW + pw[c]) * 16 + c8 * 2;
                    v0[k] = sp[off]; v1[k] = sp[off + 1];
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) { a0 += v0[k]; a1 += v1[k]; }
        }
        float z[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        if (skip) {
            const bf16x8 t = __builtin_bit_cast(bf16x8, skr);
#pragma unroll
            for (int r = 0; r < 8; ++r) z[r] += (float)t[r];
        }
        if (yprev) {
            const bf16x8 t = __builtin_bit_cast(bf16x8, ypr);
#pragma unroll
            for (int r = 0; r < 8; ++r) z[r] *= fdn_act_grad((float)t[r], act, alpha);
        }
        bf16x8 q;
#pragma unroll
        for (int r = 0; r < 8; ++r) q[r] = (__bf16)z[r];
        *(u32x4*)(out + o) = __builtin_bit_cast(u32x4, q);
    }
}

// --------------------------------------------------------------------------------------------
// weight packing: Keras (27,64,64)[tap][cin][cout] fp32 -> bf16 operand streams
//   [slice = cin/16][b*3+c][a][kh][row 64][8]  with cin = 16 slice + 8 kh + k, tap = (a,b,c),
//   cout = 32 (row/32) + sigma(row%32),  sigma(q) = 16 ((q>>2)&1) + (q&3) + 4 (q>>3)   (accumulator row -> lane-contiguous)
//   dgrad stream: contraction over the layer's cout, rows = the layer's cin, taps flipped.
// --------------------------------------------------------------------------------------------
// w_offsets != nullptr: the batched form -- blockIdx.y = layer, kernel at w + w_offsets[layer], its two streams at wf + layer * 2 * 27*64*64
// (forward stream first, dgrad stream second), like fdn_pack_conv64_weights_batch
__global__ void pack_conv64_bf16_kernel(const float* __restrict__ w, uint16_t* __restrict__ wf, uint16_t* __restrict__ wd,
                                        const int64_t* __restrict__ w_offsets) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 27 * 64 * 64) return;
    if (w_offsets) {
        w += w_offsets[blockIdx.y];
        wf += (size_t)blockIdx.y * 2 * (27 * 64 * 64);
        wd = wf + 27 * 64 * 64;
    }
    const int k = idx & 7;
    const int row = (idx >> 3) & 63;
    const int kh = (idx >> 9) & 1;
    int rest = idx >> 10;
    const int a = rest % 3; rest /= 3;
    const int bc = rest % 9;
    const int sl = rest / 9;
    const int tap = a * 9 + bc;
    const int kk = sl * 16 + kh * 8 + k;
    const int q = row & 31;
    const int jj = (row & 32) + 16 * ((q >> 2) & 1) + (q & 3) + 4 * (q >> 3);
    if (wf) { const __bf16 v = (__bf16)w[(tap * 64 + kk) * 64 + jj]; wf[idx] = __builtin_bit_cast(uint16_t, v); }
    if (wd) { const __bf16 v = (__bf16)w[((26 - tap) * 64 + jj) * 64 + kk]; wd[idx] = __builtin_bit_cast(uint16_t, v); }
}

extern "C" int fdn_pack_conv64_weights_bf16(const float* w, uint16_t* wp_fwd, uint16_t* wp_dgrad, void* stream) {
    FDN_REQUIRE(w != nullptr, "fdn_pack_conv64_weights_bf16: w is NULL");
    hipLaunchKernelGGL(pack_conv64_bf16_kernel, dim3((27 * 64 * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w,
                       wp_fwd, wp_dgrad, (const int64_t*)nullptr);
    FDN_CHECK_LAUNCH("fdn_pack_conv64_weights_bf16");
    return FDN_OK;
}

extern "C" int fdn_pack_conv64_weights_bf16_batch(const float* w_base, const int64_t* w_offsets, int n_layers, uint16_t* packs, void* stream) {
    FDN_REQUIRE(w_base && w_offsets && packs && n_layers > 0, "fdn_pack_conv64_weights_bf16_batch: NULL argument or n_layers<=0");
    hipLaunchKernelGGL(pack_conv64_bf16_kernel, dim3((27 * 64 * 64 + 255) / 256, (unsigned)n_layers), dim3(256), 0, (hipStream_t)stream, w_base,
                       packs, (uint16_t*)nullptr, w_offsets);
    FDN_CHECK_LAUNCH("fdn_pack_conv64_weights_bf16_batch");
    return FDN_OK;
}

// --------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------
namespace {

// swap = 1: the box is given in KERNEL order (first axis = h, second = d, taps likewise), see Conv64Region::swap_dh
struct Box { int od, oh, ow, ed, eh, ew, ta0, ta1, tb0, tb1, tc0, tc1, swap; };
struct Plan { FdnTile t; double cost; };

// tile = td planes (<= MT) of th x tw (<= 64) positions; a tile's MFMA time does not depend on how full the plane block is
// rows of tw + halo positions are padded to hs = 4 mod 8 (conflict-free b128 reads of 2-D position blocks); a column of positions (hw = 1:
// the w faces of a dgrad shell) is dense as it is -- consecutive positions are consecutive 32-B rows -- and padded 4x it would cap the tile
// at 2 planes
int lds_hs(int hw) { if (hw == 1) return 1; int hs = hw; while ((hs & 7) != 4) ++hs; return hs; }

// tail: one of several regions sharing a launch (the six shell slabs of a fused dgrad): total work counts, not rounds over a chip of its own
// mode2: 8 x 8 plane blocks only, dense LDS rows (hs = hw)
Plan best_plan(int N, const Box& bx, int mt, int max_rows, int max_lrows, bool tail = false, bool mode2 = false) {
    const bool full_only = (fdn_conv64bf_force_mt & 16) && bx.ed >= mt;
    Plan best{{1, 1, 1, bx.ed, bx.eh, bx.ew}, 1e30};
    const int da = bx.ta1 - bx.ta0, db = bx.tb1 - bx.tb0, dc = bx.tc1 - bx.tc0;
    const double ntap = (da + 1) * (db + 1) * (dc + 1);
    for (int td = full_only ? mt : 1; td <= bx.ed && td <= mt; ++td)
        for (int th = 1; th <= bx.eh && th <= 64; ++th)
            for (int tw = 1; tw <= bx.ew && th * tw <= 64; ++tw) {
                const int rows = (td + da) * (th + db) * (tw + dc);
                if (mode2 && (th != 8 || tw != 8)) continue;
                if (rows > max_rows || (td + da) * (th + db) * (mode2 ? tw + dc : lds_hs(tw + dc)) > max_lrows) continue;
                FdnTile t{td, th, tw, (bx.ed + td - 1) / td, (bx.eh + th - 1) / th, (bx.ew + tw - 1) / tw};
                const double tiles = (double)N * t.ntd * t.nth * t.ntw;
                // tiles with td < mt (or a single depth tap) run the predicated K loop: ~1.5x per MFMA
                const double slow = (td == mt && da == 2) ? 1.0 : 1.5;
                const double per_tile = slow * td * ntap * 64.0 / 27.0 + 0.05 * rows + 16.0;
                const double c = tail ? tiles * per_tile : (0.9 * (double)((long long)((tiles + 255) / 256)) + 0.1 * tiles / 256.0) * per_tile;
                if (c < best.cost) best = {t, c};
            }
    return best;
}

template <int MT, int MODE, bool MULTI = false>
int launch_regions(Conv64BfArgs& a, hipStream_t s) {
    if (!MULTI && a.nsrc > 1) return launch_regions<MT, MODE, true>(a, s);
    using C = Conv64BfCfg<MT>;
    if (a.nreg == 0) return FDN_OK;
    if (int rc = fdn_func_max_lds((const void*)conv64_bf16_kernel<MT, MODE, MULTI>, C::LDS_BUDGET, "conv64_bf16")) return rc;
    int blocks = 0, max_lrows = 0;
    for (int i = 0; i < a.nreg; ++i) {
        Conv64Region& r = a.reg[i];
        r.first_block = blocks;
        blocks += a.N * r.ntd * r.nth * r.ntw;
        if (r.lrows_p > max_lrows) max_lrows = r.lrows_p;
    }
    // every region's mtab sits behind ITS two buffers; size the allocation for the largest region
    const size_t lds = (size_t)max_lrows * 64 + C::MCAP * 4;          // two buffers of 32-B rows, or (MODE 2) one of 64-B rows
    hipLaunchKernelGGL((conv64_bf16_kernel<MT, MODE, MULTI>), dim3((unsigned)blocks), dim3(256), lds, s, a);
    FDN_CHECK_LAUNCH("conv64_bf16_kernel");
    return FDN_OK;
}

// plan every box, then launch the full-tap / full-depth regions with the FAST kernel and the rest with the general one
template <int MT>
int launch_bf16(Conv64BfArgs& a, const Box* boxes, int nbox, hipStream_t s) {
    using C = Conv64BfCfg<MT>;
    Conv64BfArgs fast = a, fast2 = a, slow = a;
    fast.nreg = fast2.nreg = slow.nreg = 0;
    for (int i = 0; i < nbox; ++i) {
        const Box& bx = boxes[i];
        if (bx.ed <= 0 || bx.eh <= 0 || bx.ew <= 0) continue;
        FdnTile t = best_plan(a.N, bx, MT, C::MAXROWS, C::MAXLROWS).t;
        // a secondary region too small to fill the chip on its own (< 1024 tiles = two rounds of workgroup slots): plan it by total work instead (see best_plan)
        if (i > 0 && !(fdn_conv64bf_dbg & 16) && (long long)a.N * t.ntd * t.nth * t.ntw < 1024)
            t = best_plan(a.N, bx, MT, C::MAXROWS, C::MAXLROWS, true).t;
        const bool is_fast = t.td == MT && bx.ta0 == 0 && bx.ta1 == 2 && bx.tb0 == 0 && bx.tb1 == 2 && bx.tc0 == 0 && bx.tc1 == 2;
        // MODE 2 (two 32-cin slices, one LDS buffer): full 8 x 8 x MT tiles whose sample fits the 31-bit buffer offsets; a grid that
        // divides into them exactly keeps the planner's tile, otherwise the 8 x 8 plan must not cost more tiles
        bool is_fast2 = false;
        if (is_fast && fdn_conv64bf_mode2 && (long long)a.ID * a.IH * a.IW * 128 < (1ll << 31)) {
            const Plan p2 = best_plan(a.N, bx, MT, C::MAXROWS, C::MAXLROWS2, false, true);
            if (p2.cost < 1e29 && p2.t.td == MT && (long long)p2.t.ntd * p2.t.nth * p2.t.ntw <= (long long)t.ntd * t.nth * t.ntw) { t = p2.t; is_fast2 = true; }
        }
        Conv64BfArgs& dst = is_fast2 ? fast2 : (is_fast ? fast : slow);
        Conv64Region& r = dst.reg[dst.nreg++];
        r.obd = bx.od; r.obh = bx.oh; r.obw = bx.ow; r.ebd = bx.ed; r.ebh = bx.eh; r.ebw = bx.ew;
        r.ta0 = bx.ta0; r.ta1 = bx.ta1; r.tb0 = bx.tb0; r.tb1 = bx.tb1; r.tc0 = bx.tc0; r.tc1 = bx.tc1;
        r.td = t.td; r.th = t.th; r.tw = t.tw; r.ntd = t.ntd; r.nth = t.nth; r.ntw = t.ntw;
        r.hh = t.th + (bx.tb1 - bx.tb0); r.hw = t.tw + (bx.tc1 - bx.tc0);
        r.rows = (t.td + (bx.ta1 - bx.ta0)) * r.hh * r.hw;
        r.hs = is_fast2 ? r.hw : lds_hs(r.hw);
        r.lrows = (t.td + (bx.ta1 - bx.ta0)) * r.hh * r.hs;
        r.lrows_p = is_fast2 ? (r.lrows + 15) & ~15 : (r.lrows + 31) & ~31;
        r.mg_hhhs = fdn_magic20(r.hh * r.hs);
        r.mg_hs = fdn_magic20(r.hs);
        r.mg_hhhw = fdn_magic20(r.hh * r.hw);
        r.mg_hw = fdn_magic20(r.hw);
        r.mg_thtw = fdn_magic20(t.th * t.tw);
        r.mg_tw = fdn_magic20(t.tw);
        fdn_magic40(t.ntd * t.nth * t.ntw, &r.mg_tpn_hi, &r.mg_tpn_lo);
        fdn_magic40(t.nth * t.ntw, &r.mg_thw_hi, &r.mg_thw_lo);
        fdn_magic40(t.ntw, &r.mg_ntw_hi, &r.mg_ntw_lo);
        // swizzle mode by tile shape: rows of >= 4 h values per 32 positions -> by zh; one long w row -> by w quad;
        // a column (tw <= 2) -> by h quad
        r.swz_hs = t.tw <= 2 ? 2 : 0;
        r.swz_wm = t.tw >= 16 ? 1 : 0;
        r.swap_dh = bx.swap;
#ifdef FDN_TEST_HOOKS                                    // the planner's choices, test build only (the product library reads no environment)
        if (getenv("FDN_DEBUG_PLAN"))
            fprintf(stderr, "conv64_bf16<MT=%d> box %d: out (%d,%d,%d)+(%d,%d,%d) taps a[%d,%d] b[%d,%d] c[%d,%d] tile %dx%dx%d x(%d,%d,%d) rows %d lrows %d hs %d %s\n",
                    MT, i, bx.od, bx.oh, bx.ow, bx.ed, bx.eh, bx.ew, bx.ta0, bx.ta1, bx.tb0, bx.tb1, bx.tc0, bx.tc1, t.td, t.th,
                    t.tw, t.ntd, t.nth, t.ntw, r.rows, r.lrows, r.hs, is_fast ? "FAST" : "general");
#endif
    }
    if (fast2.nreg > 0 && slow.nreg > 0 && fast.nreg == 0 && a.fout && !(fdn_conv64bf_dbg & 32)) {
        // the fused dgrad of a grid with 8 x 8 plane blocks: inner box + shell slabs in ONE launch (test build, bit 32: two launches)
        int n2 = 0, n0 = 0, max2 = 0, max0 = 0;
        for (int i = 0; i < fast2.nreg; ++i) {
            Conv64Region& r = fast2.reg[i];
            r.first_block = n2; n2 += a.N * r.ntd * r.nth * r.ntw;
            if (r.lrows_p > max2) max2 = r.lrows_p;
        }
        for (int i = 0; i < slow.nreg; ++i) {
            Conv64Region& r = slow.reg[i];
            r.first_block = n0; n0 += a.N * r.ntd * r.nth * r.ntw;
            if (r.lrows_p > max0) max0 = r.lrows_p;
        }
        const size_t lds = (size_t)(max2 > max0 ? max2 : max0) * 64 + C::MCAP * 4;
        if (a.nsrc > 1) {
            if (int rc = fdn_func_max_lds((const void*)conv64_bf16_fused_kernel<MT, true>, C::LDS_BUDGET, "conv64_bf16_fused")) return rc;
            hipLaunchKernelGGL((conv64_bf16_fused_kernel<MT, true>), dim3((unsigned)(n2 + n0)), dim3(256), lds, s, fast2, slow, n2);
            FDN_CHECK_LAUNCH("conv64_bf16_fused_kernel");
            return FDN_OK;
        }
        if (int rc = fdn_func_max_lds((const void*)conv64_bf16_fused_kernel<MT>, C::LDS_BUDGET, "conv64_bf16_fused")) return rc;
        hipLaunchKernelGGL((conv64_bf16_fused_kernel<MT>), dim3((unsigned)(n2 + n0)), dim3(256), lds, s, fast2, slow, n2);
        FDN_CHECK_LAUNCH("conv64_bf16_fused_kernel");
        return FDN_OK;
    }
    if (int rc = launch_regions<MT, 2>(fast2, s)) return rc;
    if (int rc = launch_regions<MT, 1>(fast, s)) return rc;
    return launch_regions<MT, 0>(slow, s);
}

int launch_boxes(Conv64BfArgs& a, const Box* boxes, int nbox, hipStream_t s) {
    int mt = fdn_conv64bf_force_mt & 15;
    if (mt != 4 && mt != 8) {
        const Plan p8 = best_plan(a.N, boxes[0], 8, Conv64BfCfg<8>::MAXROWS, Conv64BfCfg<8>::MAXLROWS);
        const Plan p4 = best_plan(a.N, boxes[0], 4, Conv64BfCfg<4>::MAXROWS, Conv64BfCfg<4>::MAXLROWS);
        mt = p8.cost <= p4.cost ? 8 : 4;
        // a grid that gives the 8-plane layout at most one workgroup per CU runs faster as twice as many 4-plane tiles (a workgroup
        // alone on a CU does not saturate the matrix pipe): (4,32^3) forward 0.032 -> 0.030 ms, fused dgrad 0.068 -> 0.061 ms
        if ((long long)a.N * p8.t.ntd * p8.t.nth * p8.t.ntw <= 256) mt = 4;
    }
    return mt == 8 ? launch_bf16<8>(a, boxes, nbox, s) : launch_bf16<4>(a, boxes, nbox, s);
}

}  // namespace

int fdn_conv64_bf16_launch(const uint16_t* x, const uint16_t* wpack, const float* bias, const uint16_t* residual, uint16_t* y,
                           float* ypad, const uint16_t* fskip, const uint16_t* fy, uint16_t* fout, int N, int ID, int IH,
                           int IW, int OD, int OH, int OW, int off, int zero_mode, int act, float alpha, hipStream_t s,
                           uint16_t* ymask, const uint16_t* fmask, const FdnExtraSrcBf* extra) {
    Conv64BfArgs a;
    a.x = x; a.wp = wpack; a.bias = bias; a.res = residual; a.y = y; a.ypad = ypad;
    a.x1 = a.x2 = a.wp1 = a.wp2 = nullptr; a.nsrc = 1;
    if (extra && extra->nsrc > 1) {
        FDN_REQUIRE(fout && zero_mode && off == -1 && extra->nsrc <= 3, "conv64 (bf16): further sources belong to a fused dgrad, 1..3 in all");
        a.x1 = extra->x1; a.x2 = extra->x2; a.wp1 = extra->wp1; a.wp2 = extra->wp2; a.nsrc = extra->nsrc;
    }
    a.fskip = fskip; a.fy = fmask ? nullptr : fy; a.fout = fout;
    a.ymask = ymask; a.fmask = fmask;
    a.N = N; a.ID = ID; a.IH = IH; a.IW = IW; a.OD = OD; a.OH = OH; a.OW = OW;
    a.off = off; a.zero_mode = zero_mode; a.act = act; a.alpha = alpha; a.dbg = fdn_conv64bf_dbg;
    const Box full{0, 0, 0, OD, OH, OW, 0, 2, 0, 2, 0, 2, 0};
    if (!(fout && zero_mode && off == -1)) return launch_boxes(a, &full, 1, s);
    // fused dgrad on the padded grid: inner box (27 taps) + six 1-voxel shell slabs (9 taps), as in conv64_mfma.hip
    // The d faces (one depth tap) are handed over in kernel order (h, d): the kernel slides its accumulator planes along the first
    // axis, which needs the 3-tap axis there -- planes = 8 h rows x (1 x <=64 w) positions instead of 1 plane x 60 positions per tile
    // (2288 -> 408 tiles at (4,128^3)).  The w faces stage a dense column of positions (lds_hs): 8 planes per tile instead of 2.
    const Box boxes[7] = {
        {1, 1, 1, ID, IH, IW, 0, 2, 0, 2, 0, 2, 0},
        {0, 0, 0, OH, 1, OW, 0, 2, 2, 2, 0, 2, 1},    {0, ID + 1, 0, OH, 1, OW, 0, 2, 0, 0, 0, 2, 1},     // d faces: (h, d = 0 | ID+1, w)
        {1, 0, 0, ID, 1, OW, 0, 2, 2, 2, 0, 2, 0},    {1, IH + 1, 0, ID, 1, OW, 0, 2, 0, 0, 0, 2, 0},
        {1, 1, 0, ID, IH, 1, 0, 2, 0, 2, 2, 2, 0},    {1, 1, IW + 1, ID, IH, 1, 0, 2, 0, 2, 0, 0, 0}};
    return launch_boxes(a, boxes, 7, s);
}

int fdn_fold_halo_border_bf16_launch(const float* s0, const float* s1, const float* s2, int nsrc, const uint16_t* skip,
                                     const uint16_t* yprev, int act, float alpha, uint16_t* out, int N, int D, int H, int W,
                                     hipStream_t s) {
    const int ID = D > 2 ? D - 2 : 0, IH = H > 2 ? H - 2 : 0;
    const int64_t per_n = (int64_t)(D > 1 ? 2 : 1) * H * W + (int64_t)ID * (H > 1 ? 2 : 1) * W + (int64_t)ID * IH * (W > 1 ? 2 : 1);
    const int64_t total = (int64_t)N * per_n * 8;
    int64_t nb = (total + 255) / 256;
    if (nb < 1) nb = 1;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(fold_halo_border_bf16_kernel, dim3((unsigned)nb), dim3(256), 0, s, s0, s1, s2, nsrc, skip, yprev, act,
                       alpha, out, N, D, H, W);
    FDN_CHECK_LAUNCH("fold_halo_border_bf16_kernel");
    return FDN_OK;
}

#ifdef FDN_TEST_HOOKS
extern "C" int fdn_debug_set_conv64_bf16_mt(int mt) { fdn_conv64bf_force_mt = mt; return FDN_OK; }
extern "C" int fdn_debug_set_conv64_bf16_dbg(int bits) { fdn_conv64bf_dbg = bits; return FDN_OK; }
extern "C" int fdn_debug_set_conv64_bf16_mode2(int on) { fdn_conv64bf_mode2 = on; return FDN_OK; }
#endif

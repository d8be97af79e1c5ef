// Device side of conv64_wino2d.hip (2-D Winograd F(HM,3) along H x F(4,3) along W, HM = 2 or 4): argument struct, constants and the
// kernel body, shared with conv64_wino.hip (conv64_wino2d_shell_kernel: the fused-dgrad launch = this body on the inner box + the 1-D
// body on the shell).  See conv64_wino2d.hip for the design notes.
#pragma once
#include "fdn_common.h"
#include "conv64_pack.h"

namespace {

struct Wino2Args {
    const float* x;
    const float* up;        // 2-D Winograd operand stream (third part of the pack)
    const float* bias;
    const float* res;
    float* y;
    const float* fskip;     // fused fold (dgrad mode): see conv64_args.h
    const float* fy;
    float* fout;
    // Sign masks (round 6, the fp32 twin of conv64_args.h's): the activation gradient needs ONE BIT of the producer's output, and at the
    // 48^3 grids an epilogue operand costs what streaming it at the HBM peak would take (0.022 of the 0.029 ms y costs a fused dgrad).
    // Planar layout [cout / 16][voxel] of uint16_t words, bit b = (y[voxel][16 (cout / 16) + b] > 0): a wave owns one plane, a cell row's
    // four voxels are four consecutive words -- ONE 8-B store (forward, MASK kernels: ymask) / load (fused dgrad: fmask instead of fy)
    // per cell row and quarter-wave instead of four 16-B loads.  8 B per voxel beside the 256-B row.
    uint16_t* ymask;
    const uint16_t* fmask;
    // Multi-source fused dgrad (round 6): dz_prev = fold(sum_s conv_T(x_s, W_s)) for up to three layers that share their INPUT (the three
    // heads' 64->64 convs read rb, SR4DFlowNet.py:39-46) -- the stage loop runs once per source over the same tile and accumulates into
    // the same output registers: one epilogue, one skip / mask read, one set of stores instead of three chained launches.  Source s > 0:
    // rows x1 / x2, weight stream at up + wd1 / wd2 bytes (the packs of one model lie in one buffer; the host orders the sources by address).
    const float* x1;
    const float* x2;
    int wd1, wd2, nsrc, wspan;          // wspan: largest wd (the weight resource covers up + wspan + one stream)
    int N, ID, IH, IW, OD, OH, OW;
    int off, zero_mode, act;
    float alpha;
    int dbg;                // ablation bits (test build only): 1 = weight stream stride 0, 4 = no staging, 8 = no epilogue, 32 = epilogue operands from cache-resident rows, 128 = no XCD remap
    int hm;                 // output rows per cell: 2 = F(2,3) along H, 4 = F(4,3) along H (selects the kernel instantiation; host side only)
    int mb;                 // 16-cell M-blocks per wave: 2 = full tiles, 1 = half-size tiles (host side only)
    int split;              // 1: `up` is the bf16 x 3 stream and the products run on the bf16 pipe (FDN_ALGO_WINO_BF16X3; host side only)
    int obd, obh, obw, ebd, ebh, ebw;      // output box (h extent a multiple of hm, w extent a multiple of 4)
    int td, ch, cw, ntd, nth, ntw;         // tile in (depth planes, cell rows, cell columns) and tile counts; a cell = hm x 4 voxels (h, w)
    int cpp, rows, items;                  // cells per plane, staged rows = (td + 2) * cpp, rows * 16
    unsigned mg_cpp, mg_cw;
    unsigned mg_tpn_hi, mg_tpn_lo, mg_thw_hi, mg_thw_lo, mg_ntw_hi, mg_ntw_lo;
};


constexpr int kW2Rows = 40;                 // staged cell-planes per tile
constexpr int kW2Row = 288;                 // bytes per LDS row: 64 cin + 32-B pad (conflict-free ds_read_b128 over 16 consecutive rows)
constexpr int kW2Plane = kW2Rows * kW2Row + 64;
constexpr int kW2Lds = 6 * kW2Plane + 96 * 4 + kW2Rows * 48;
// Half-size tiles (MB = 1: ONE 16-cell M-block per wave, <= 16 cells and <= kW2RowsH staged cell-planes per tile) for grids whose full-size
// tiles do not fill the chip: at (8,24^3) 216 tiles leave 40 CUs idle and the other 216 with a single workgroup, 432 half tiles put two
// workgroups on most CUs (a workgroup alone on its CU has nobody to cover its staging, barriers and epilogue).  Same LDS image, fewer rows.
constexpr int kW2RowsH = 24;
// SPLIT (FDN_ALGO_WINO_BF16X3): the Winograd-domain products on the bf16 matrix pipe.  Both operands are split EXACTLY into three bf16
// pieces, v = hi + mid + lo (each the round-to-nearest bf16 of what the pieces before it leave: 8 + 8 + 8 significand bits), and the six
// cross terms hi.hi, mid.hi, lo.hi, hi.mid, mid.mid, hi.lo run on v_mfma_f32_16x16x32_bf16 with fp32 accumulation (a bf16 x bf16 product is
// exact in fp32; dropped: mid.lo + lo.mid + lo.lo <= 2^-25 |u||v|, under the half ulp an fp32 multiply rounds away).  gfx950's fp32
// MFMA runs at 1/16 of the bf16 rate, so six terms cost 6/16 of the matrix time, and the bf16 pipe -- unlike the fp32 one -- lets the
// co-resident workgroup's staging run beside it.  An LDS row holds 32 cin x 3 pieces (64 B each) + 32 B pad = 224 B (conflict-free
// ds_read_b128 over 16 consecutive rows like the 288-B fp32 rows), so a stage runs as TWO cin passes over the same 54 KB of planes
// and two workgroups still share a CU.
constexpr int kW2RowS = 224;
template <int MB, bool SPLIT = false> struct W2Geo {
    static constexpr int rows = MB == 2 ? kW2Rows : kW2RowsH;
    static constexpr int row = SPLIT ? kW2RowS : kW2Row;
    static constexpr int plane = rows * row + 64;
    static constexpr int lds = 6 * plane + 96 * 4 + rows * 48;
};
constexpr int kW2UnitS = 3072;              // bytes of one (stage, pass, kd, xw) step of a wave's split weight stream: 3 pieces x 64 lanes x 16 B
constexpr int kW2UA = 3;                    // transform items per thread (<= 640 items = 40 rows x 16 chunks)
constexpr int kW2RDB = 6;                   // weight-fragment ring depth
constexpr int kW2RDA = 3;                   // cell-fragment ring depth
constexpr int kW2Dep = 1;                   // staging: items (12 x 16-B loads each) in flight per thread
// F(4,3) x F(4,3) (HM = 4) uses the interpolation points 0, +-3/4, +-3/2, inf instead of Lavin & Gray's 0, +-1, +-2, inf: the same even / odd
// structure (b = 2a), every constant exact in fp32, and a quarter of the fp32 error (transform entries up to 3.4 instead of 8; measured
// e_cond 3e-7 vs 1.3e-6 on dz-like operands -- level with F(2,3) x F(4,3) on the classic points).  a = 3/4, b = 3/2:
constexpr float kPa = 0.75f, kPb = 1.5f, kPa2 = 0.5625f, kPb2 = 2.25f, kPa3 = 0.421875f, kPb3 = 3.375f;
constexpr float kPab2 = 1.6875f, kPa2b = 0.84375f;           // a b^2, a^2 b
constexpr float kPs = 2.8125f, kPp = 1.265625f;              // a^2 + b^2, a^2 b^2
constexpr unsigned kW2Big = 0x40000000u;    // "reads zero": any sum containing it is >= 2^30 > the sample's bytes

// (a __device__ body + thin __global__ wrappers: conv64_wino.hip runs it as the head of the fused-dgrad launch that also carries the shell)
// RDB / RDA / DEP: weight-fragment ring, cell-fragment ring, staging items in flight per thread (the product values are the defaults;
// the one-workgroup-per-CU occupancy experiment of conv64_wino2d.hip instantiates deeper ones)
// HM: output rows per cell = the F(HM,3) transform along H (NS = HM + 2 sequential stages, Y = HM x 4 voxels per cell)
// MB: 16-cell M-blocks per wave (2 = the full tile; 1 = half-size tiles: each xw keeps TWO accumulators, even / odd k-steps, so that no
// MFMA waits for its predecessor's result -- the sum over cin is split differently, equal to the full tile to fp32 rounding)
// MASK: the forward also writes the sign mask of its output (p.ymask) / the fused dgrad reads p.fmask instead of the rows of p.fy
template <bool FUSED, int HM = 2, int RDB = kW2RDB, int RDA = kW2RDA, int DEP = kW2Dep, int MB = 2, bool SPLIT = false, bool MASK = false>
__device__ __forceinline__ void conv64_wino2d_body(const Wino2Args& p, const int block_id, char* const smem) {
    constexpr int UA = kW2UA, SPT = 24;
    constexpr int kW2Plane = W2Geo<MB, SPLIT>::plane;       // (shadows the full-tile constants: every row / plane offset below is the variant's own)
    constexpr int kW2Row = W2Geo<MB, SPLIT>::row;
    static_assert(MB == 1 || MB == 2, "one or two M-blocks per wave");
    static_assert(!SPLIT || (HM == 4 && MB == 2), "the bf16 x 3 products exist for full F(4,3) x F(4,3) tiles");
    constexpr int NS = HM + 2;                                // stages = Winograd coordinates along H = input rows per cell
    static_assert(HM == 2 || HM == 4, "F(2,3) or F(4,3) along H");
    static_assert(HM == 2 || DEP == 1, "the F(4,3) staging holds one item in flight");
    static_assert(SPT % RDB == 0 && SPT % RDA == 0, "ring slots must be compile-time");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;               // = cout block
    const int c = lane & 15;
    const int q = lane >> 4;
    int* mtab = (int*)(smem + 6 * kW2Plane);    // [0,32): output index of the cell's first voxel or -1; [32,64): fused-fold index or -1;
                                                // [64,96): (h | w << 16) of the cell's first voxel (fused mode)

    // ---- which tile (scalar multiply-shift divisions, host-made magics; XCD-aware order as in conv64_wino.hip) ----
    const int tiles_per_n = p.ntd * p.nth * p.ntw;
    int b = block_id;
    if (!(FDN_DBG_BITS(p) & 128)) {
        const int T = p.N * tiles_per_n, qq = T >> 3, r = T & 7, xcd = b & 7;
        b = xcd * qq + min(xcd, r) + (b >> 3);
    }
    const int n = fdn_udiv40(b, p.mg_tpn_hi, p.mg_tpn_lo);
    b -= n * tiles_per_n;
    const int tdi = fdn_udiv40(b, p.mg_thw_hi, p.mg_thw_lo);
    b -= tdi * (p.nth * p.ntw);
    const int thi = fdn_udiv40(b, p.mg_ntw_hi, p.mg_ntw_lo);
    const int p0d = p.obd + tdi * p.td, p0h = p.obh + thi * p.ch * HM, p0w = p.obw + (b - thi * p.ntw) * p.cw * 4;
    const int ng = p.td * p.cpp;

    if (tid < 16 * MB) {
        int g = -1, gf = -1, hw = 0;
        if (tid < ng) {
            const int md = fdn_div20(tid, p.mg_cpp);
            const int j = tid - md * p.cpp;
            const int mh = fdn_div20(j, p.mg_cw);
            const int pd = p0d + md, ph = p0h + HM * mh, pw = p0w + 4 * (j - mh * p.cw);
            if (pd < p.obd + p.ebd && ph < p.obh + p.ebh && pw < p.obw + p.ebw) {
                g = ((n * p.OD + pd) * p.OH + ph) * p.OW + pw;
                hw = ph | (pw << 16);
                if (FUSED) {
                    const int id = pd - 1;
                    if (id >= 1 && id <= p.ID - 2) gf = ((n * p.ID + id) * p.IH + (ph - 1)) * p.IW + (pw - 1);
                }
            }
        }
        mtab[tid] = g; mtab[32 + tid] = gf; mtab[64 + tid] = hw;
    }

    // ---- this lane's two cell rows (tap 0, plane xw = 0) ----
    int abase[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        int m = mb * 16 + c;
        m = m < ng ? m : ng - 1;
        abase[mb] = m * kW2Row + q * 16;
    }

    // ---- staging plan, once per tile, in LDS: per staged cell-plane r the byte offsets (from the sample's first voxel) of its 4 input
    // rows and 6 input columns with the boundary rule applied (kW2Big = reads zero: any sum containing it is out of the buffer's range).
    // ptab[r] = {row 0..NS-1, column 0..5, ..} (48 B); a transform item adds its channel chunk.  Keeping the plan out of the register
    // file is what lets a thread hold two items' input rows in flight (24 x 16 B) beside the 64 output accumulators.
    unsigned* ptab = (unsigned*)(smem + 6 * kW2Plane + 96 * 4);
    if (tid < p.rows) {
        const int r = tid;
        const int zd = fdn_div20(r, p.mg_cpp);
        const int j = r - zd * p.cpp;
        const int mh = fdn_div20(j, p.mg_cw);
        const int q0d = p0d - 1 + p.off, q0h = p0h - 1 + p.off + HM * mh, q0w = p0w - 1 + p.off + 4 * (j - mh * p.cw);
        int qd = q0d + zd;
        bool okd = true;
        if (p.zero_mode) okd = (unsigned)qd < (unsigned)p.ID;
        else qd = min(max(qd, 0), p.ID - 1);
#pragma unroll
        for (int jj = 0; jj < NS; ++jj) {
            int qh = q0h + jj;
            bool ok = okd;
            if (p.zero_mode) ok = ok && (unsigned)qh < (unsigned)p.IH;
            else qh = min(max(qh, 0), p.IH - 1);
            ptab[r * 12 + jj] = ok ? (unsigned)((qd * p.IH + qh) * p.IW) * 256u : kW2Big;
        }
#pragma unroll
        for (int ii = 0; ii < 6; ++ii) {
            int qw = q0w + ii;
            bool ok = true;
            if (p.zero_mode) ok = (unsigned)qw < (unsigned)p.IW;
            else qw = min(max(qw, 0), p.IW - 1);
            ptab[r * 12 + NS + ii] = ok ? (unsigned)qw * 256u : kW2Big;
        }
    }
    // LDS offsets of item u's output row (-1: past the tile's rows) / plan row; chunk offset.  F(2,3): held in registers; F(4,3): Y takes
    // 128 of the 256 registers, so the six values are recomputed at every stage (a dozen VALU per stage) instead of living through the K loops
    // (F(4,3): the thread id is re-derived from the scalar wave id and mbcnt where it is needed late -- staging items, epilogue -- so that
    // neither it nor the lane's cell / quarter indices occupy registers through the K loops; hipcc spilled exactly those three)
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    auto tid_now = [&]() { return HM == 2 ? tid : wave_s * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); };
    auto item_vrow = [&](int u) { const int i = u * 256 + tid_now(), r = i >> 4; return r < p.rows ? r * kW2Row + (i & 15) * 16 : -1; };
    auto item_prow = [&](int u) { const int r = (u * 256 + tid_now()) >> 4; return (r < p.rows ? r : 0) * 48; };
    int vrow[HM == 2 ? UA : 1], prow[HM == 2 ? UA : 1];
    if constexpr (HM == 2) {
#pragma unroll
        for (int u = 0; u < UA; ++u) {
            vrow[u] = item_vrow(u);
            prow[u] = item_prow(u);
        }
    }
    const unsigned chunkb_held = (unsigned)(tid & 15) * 16u;
    auto chunkb_now = [&]() { return HM == 2 ? chunkb_held : (unsigned)(tid_now() & 15) * 16u; };
    __syncthreads();                                          // mtab + ptab visible
    const unsigned sample_bytes = (unsigned)(p.ID * p.IH * p.IW) * 256u;
    __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.x + (size_t)n * p.ID * p.IH * p.IW * 64), 0, sample_bytes, 0x00020000);
    int wsrc_off = 0;                                         // FUSED, source s > 0: byte distance of its weight stream from source 0's

    // weight stream: unit (1024 B) index = ((nb*NS + xh)*3 + kd)*24 + xw*4 + g
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.up, 0, NS * 18 * 64 * 64 * 4 + (FUSED ? p.wspan : 0), 0x00020000);
    const int wvoff = wave * (NS * 72 * 1024) + lane * 16;
    const int bmul = (FDN_DBG_BITS(p) & 1) ? 0 : 1024;
    f32x4 A[SPLIT ? 1 : RDA][MB], B[SPLIT ? 1 : RDB];
    auto ldb = [&](int slot, int xh, int kd, int j) {
        B[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, ((xh * 3 + kd) * 24 + j) * bmul + (FUSED ? wsrc_off : 0), 0));
    };
    auto lda = [&](int slot, int tapb, int j) {             // tapb: byte offset of the depth tap's rows
        const int o = tapb + (j >> 2) * kW2Plane + (j & 3) * 64;
        A[slot][0] = *(const f32x4*)(smem + abase[0] + o);
        if constexpr (MB == 2) A[slot][1] = *(const f32x4*)(smem + abase[1] + o);
    };

    f32x4 Y[HM][4][MB];                                      // [output row][output column][M-block]: cout 16w + 4q .. + 3 of that voxel
#pragma unroll
    for (int hr = 0; hr < HM; ++hr)
#pragma unroll
        for (int wi = 0; wi < 4; ++wi)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) Y[hr][wi][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int tapstep = p.cpp * kW2Row;

    // B_w^T of F(4,3) on the six column chunks of one item, written as the six xw planes of the item's LDS row.  Points 0, +-a, +-b, inf:
    //   rows (a2b2, 0, -(a2+b2), 0, 1, 0)  (0, -+a b2, -b2, +-a, 1, 0)  (0, -+a2 b, -a2, +-b, 1, 0)  (0, a2b2, 0, -(a2+b2), 0, 1)
    // HM = 2: a = 1, b = 2 -- (4,0,-5,0,1,0) (0,-4,-4,1,1,0) (0,4,-4,-1,1,0) (0,-2,-1,2,1,0) (0,2,-1,-2,1,0) (0,4,0,-5,0,1); HM = 4: a = 3/4, b = 3/2
    auto wtransform_ = [&](char* vp, const f32x4 x0, const f32x4 x1, const f32x4 x2, const f32x4 x3, const f32x4 x4, const f32x4 x5, const bool ACC) {
        f32x4 t1, t2, t3, t4, o0, o5;
        if constexpr (HM == 2) {
            t1 = x4 - 4.f * x2; t2 = x3 - 4.f * x1;
            t3 = x4 - x2; t4 = 2.f * (x3 - x1);
            o0 = 4.f * x0 - 5.f * x2 + x4;
            o5 = 4.f * x1 - 5.f * x3 + x5;
        } else {
            t1 = x4 - kPb2 * x2; t2 = kPa * x3 - kPab2 * x1;
            t3 = x4 - kPa2 * x2; t4 = kPb * x3 - kPa2b * x1;
            o0 = kPp * x0 - kPs * x2 + x4;
            o5 = kPp * x1 - kPs * x3 + x5;
        }
        if (ACC) {                                            // add to what the previous stage left in the item's row (same thread wrote it)
            o0 += *(const f32x4*)(vp);
            const f32x4 p1 = *(const f32x4*)(vp + kW2Plane), p2 = *(const f32x4*)(vp + 2 * kW2Plane);
            const f32x4 p3 = *(const f32x4*)(vp + 3 * kW2Plane), p4 = *(const f32x4*)(vp + 4 * kW2Plane);
            o5 += *(const f32x4*)(vp + 5 * kW2Plane);
            *(f32x4*)(vp) = o0;
            *(f32x4*)(vp + kW2Plane) = p1 + (t1 + t2);
            *(f32x4*)(vp + 2 * kW2Plane) = p2 + (t1 - t2);
            *(f32x4*)(vp + 3 * kW2Plane) = p3 + (t3 + t4);
            *(f32x4*)(vp + 4 * kW2Plane) = p4 + (t3 - t4);
            *(f32x4*)(vp + 5 * kW2Plane) = o5;
            return;
        }
        *(f32x4*)(vp) = o0;
        *(f32x4*)(vp + kW2Plane) = t1 + t2;
        *(f32x4*)(vp + 2 * kW2Plane) = t1 - t2;
        *(f32x4*)(vp + 3 * kW2Plane) = t3 + t4;
        *(f32x4*)(vp + 4 * kW2Plane) = t3 - t4;
        *(f32x4*)(vp + 5 * kW2Plane) = o5;
    };
    auto wtransform = [&](char* vp, const f32x4 x0, const f32x4 x1, const f32x4 x2, const f32x4 x3, const f32x4 x4, const f32x4 x5) {
        wtransform_(vp, x0, x1, x2, x3, x4, x5, false);
    };

    // SPLIT: a stage = two cin passes (32 cin each) over the same planes, each folded into Y on its own (the fold is linear; accumulators
    // that lived through both passes would sit beside the staging rows of the second: 48 registers the staging does not have)
    constexpr int NPASS = SPLIT ? 2 : 1;
    fdn_bf16x8 US[SPLIT ? 2 : 1][3], VS[3][2];              // SPLIT operands: weight pieces of this step and the next, cell pieces of this step
    const int wvoff_s = wave * (NS * 2 * 18 * kW2UnitS) + lane * 16;
    auto ldu = [&](int slot, int step) {                    // step = ((xh * 2 + pass) * 3 + kd) * 6 + xw of the wave's stream
        const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void*)p.up, 0, 4 * NS * 2 * 18 * kW2UnitS, 0x00020000);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
            US[SPLIT ? slot : 0][pc] = __builtin_bit_cast(fdn_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(srs, wvoff_s + pc * 1024, step * ((FDN_DBG_BITS(p) & 1) ? 0 : kW2UnitS), 0));
    };
    auto ldv = [&](int pc, int o) {                         // o: byte offset of (depth tap, xw plane)
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) VS[pc][mb] = __builtin_bit_cast(fdn_bf16x8, *(const fdn_u32x4*)(smem + abase[mb < MB ? mb : 0] + o + pc * 64));
    };
    const int nsrc = (FUSED && !SPLIT) ? p.nsrc : 1;
#pragma unroll 1
    for (int src = 0; src < nsrc; ++src) {
    if (FUSED && src) {                                       // (scalar selects: no indexed kernel-argument access)
        xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)((src == 1 ? p.x1 : p.x2) + (size_t)n * p.ID * p.IH * p.IW * 64), 0, sample_bytes, 0x00020000);
        wsrc_off = src == 1 ? p.wd1 : p.wd2;
    }
#pragma unroll 1
    for (int ss = 0; ss < NS * NPASS; ++ss) {
        const int xh = SPLIT ? ss >> 1 : ss;
        if (ss || src) __syncthreads();                      // everyone finished reading the previous stage's planes
        // a wave whose 64 items of a pass all lie past the tile's last item requests nothing in that pass (640 items: waves 2, 3 of pass 2)
        const int items_eff = ((FDN_DBG_BITS(p) & 4) ? 0 : p.items) - __builtin_amdgcn_readfirstlane(wave) * 64;
        if constexpr (SPLIT) {
            // ---- stage xh, cin pass: the same V = sum_j B_h^T[xh][j] x[row j] and B_w^T as below, for 32 of the 64 input channels: items =
            // (staged cell-plane r, 16-B chunk of 4 cin) = rows x 8 <= 320, i.e. one item per thread and 64 more, which rotate over the waves
            // with the stage; every value is split into its three bf16 pieces on the way to LDS (row = [piece][32 cin]).  No incremental
            // stages here: what a stage wrote is overwritten by the next cin pass. ----
            const bool ends = xh == 0 || xh == 5;
            const int ia = xh == 0 ? 0 : 1, ib = ends ? ia + 2 : 2, ic = ends ? ia + 4 : 3;
            const float ca = ends ? kPp : (xh == 1 ? -kPab2 : (xh == 2 ? kPab2 : (xh == 3 ? -kPa2b : kPa2b)));
            const float cb = ends ? -kPs : (xh <= 2 ? -kPb2 : -kPa2);
            const float cc = ends ? 1.f : (xh == 1 ? kPa : (xh == 2 ? -kPa : (xh == 3 ? kPb : -kPb)));
            const int wave_r = (wave_s + ss) & 3;            // scalar: the wave that takes the 64 extra items changes with the stage
            const int items_s = ((FDN_DBG_BITS(p) & 4) ? 0 : p.rows * 8) - wave_r * 64;
            const int lane_n = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (u * 256 >= items_s) break;
                const int i = u * 256 + wave_r * 64 + lane_n, r = i >> 3;
                const bool valid = r < p.rows;
                const unsigned* pr = ptab + (valid ? r : 0) * 12;
                const unsigned chunkb = (unsigned)((ss & 1) * 128 + (i & 7) * 16);
                const unsigned ha = pr[ia] + chunkb, hb = pr[ib] + chunkb, hc = pr[ic] + chunkb, hd = pr[4] + chunkb;
                const unsigned wo[6] = {pr[6], pr[7], pr[8], pr[9], pr[10], pr[11]};
                f32x4 xa[6], xb[6], v[6];
#pragma unroll
                for (int ii = 0; ii < 6; ++ii) {
                    xa[ii] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ha + wo[ii], 0, 0));
                    xb[ii] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, hb + wo[ii], 0, 0));
                }
#pragma unroll
                for (int ii = 0; ii < 6; ++ii) v[ii] = ca * xa[ii] + cb * xb[ii];
                if (ends) {
#pragma unroll
                    for (int ii = 0; ii < 6; ++ii) xa[ii] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, hc + wo[ii], 0, 0));
#pragma unroll
                    for (int ii = 0; ii < 6; ++ii) v[ii] += xa[ii];
                } else {
#pragma unroll
                    for (int ii = 0; ii < 6; ++ii) {
                        xa[ii] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, hc + wo[ii], 0, 0));
                        xb[ii] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, hd + wo[ii], 0, 0));
                    }
#pragma unroll
                    for (int ii = 0; ii < 6; ++ii) v[ii] += cc * xa[ii] + xb[ii];
                }
                if (!valid) continue;
                char* vp = smem + r * kW2Row + (i & 7) * 8;
                const f32x4 t1 = v[4] - kPb2 * v[2], t2 = kPa * v[3] - kPab2 * v[1];
                const f32x4 t3 = v[4] - kPa2 * v[2], t4 = kPb * v[3] - kPa2b * v[1];
                const f32x4 o[6] = {kPp * v[0] - kPs * v[2] + v[4], t1 + t2, t1 - t2, t3 + t4, t3 - t4, kPp * v[1] - kPs * v[3] + v[5]};
#pragma unroll
                for (int xw = 0; xw < 6; ++xw) {
                    fdn_u32x2 hi, mid, lo;
                    fdn_split3(o[xw], hi, mid, lo);
                    *(fdn_u32x2*)(vp + xw * kW2Plane) = hi;
                    *(fdn_u32x2*)(vp + xw * kW2Plane + 64) = mid;
                    *(fdn_u32x2*)(vp + xw * kW2Plane + 128) = lo;
                }
            }
        } else if constexpr (HM == 2) {
            // ---- stage xh: V = x[ra] + sgn * x[rb], rows (0,2,-) (1,2,+) (1,2,-) (1,3,-); then B_w^T; all 64 cin ----
            const float sgn = xh == 1 ? 1.f : -1.f;
            f32x4 xa[DEP][6], xb[DEP][6];
            auto issue = [&](int u, int buf) {
                const unsigned* pr = (const unsigned*)((const char*)ptab + prow[u]);
                const unsigned ha = (xh == 0 ? pr[0] : pr[1]) + chunkb_held;
                const unsigned hb = (xh == 3 ? pr[3] : pr[2]) + chunkb_held;
                const unsigned wo[6] = {pr[4], pr[5], pr[6], pr[7], pr[8], pr[9]};
#pragma unroll
                for (int ii = 0; ii < 6; ++ii) {
                    xa[buf][ii] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ha + wo[ii], 0, 0));
                    xb[buf][ii] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, hb + wo[ii], 0, 0));
                }
            };
#pragma unroll
            for (int u = 0; u < DEP - 1; ++u)
                if (u * 256 < items_eff) issue(u, u);
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                if (u * 256 >= items_eff) break;
                if (u + DEP - 1 < UA && (u + DEP - 1) * 256 < items_eff) issue(u + DEP - 1, (u + DEP - 1) % DEP);
                if (vrow[u] < 0) continue;
                const int bf = u % DEP;
                wtransform(smem + vrow[u], xa[bf][0] + sgn * xb[bf][0], xa[bf][1] + sgn * xb[bf][1], xa[bf][2] + sgn * xb[bf][2],
                           xa[bf][3] + sgn * xb[bf][3], xa[bf][4] + sgn * xb[bf][4], xa[bf][5] + sgn * xb[bf][5]);
            }
        } else {
            // ---- stage xh of F(4,3) along H: V = sum_j B_h^T[xh][j] x[row j] over the cell-plane's six input rows, B_h^T (a = 3/4, b = 3/2) =
            //   xh 0: (a2b2, 0, -(a2+b2), 0, 1, 0)    1: (0, -a b2, -b2,  a, 1, 0)    2: (0,  a b2, -b2, -a, 1, 0)
            //   xh 5: (0, a2b2, 0, -(a2+b2), 0, 1)    3: (0, -a2 b, -a2,  b, 1, 0)    4: (0,  a2 b, -a2, -b, 1, 0)
            // three rows (stages 0, 5) or four; two rows (12 chunks) are in flight at a time, combined into V, then the next two
            // (or the last one): the item's 18 / 24 chunks never sit in registers together (Y holds 128 of the wave's 256).
            // Stages 2 and 4 are INCREMENTAL: V[2] = V[1] + 2 a b2 x1 - 2 a x3 and V[4] = V[3] + 2 a2 b x1 - 2 b x3 differ from the stage
            // before them in rows 1 and 3 only, and B_w^T is linear -- the item loads those two rows (12 chunks instead of 24), transforms
            // the difference and adds it to the six values it wrote itself a stage earlier (still in its LDS row: the K loop only reads).
            // A load costs the SIMD ~35 matrix-pipe cycles whatever its width (tools/mfma_ldcost.hip), an LDS read 3-6. ----
            const bool ends = xh == 0 || xh == 5;            // scalar
            const int ia = xh == 0 ? 0 : 1, ib = ends ? ia + 2 : 2, ic = ends ? ia + 4 : 3;
            const float ca = ends ? kPp : (xh == 1 ? -kPab2 : (xh == 2 ? kPab2 : (xh == 3 ? -kPa2b : kPa2b)));
            const float cb = ends ? -kPs : (xh <= 2 ? -kPb2 : -kPa2);
            const float cc = ends ? 1.f : (xh == 1 ? kPa : (xh == 2 ? -kPa : (xh == 3 ? kPb : -kPb)));
            const unsigned chunkb = chunkb_now();
            const bool incr = (xh == 2 || xh == 4) && !(FDN_DBG_BITS(p) & 16);      // (test build, bit 16: every stage from its own rows)
            if (incr) {
                const float c1 = xh == 2 ? 2.f * kPab2 : 2.f * kPa2b, c3 = xh == 2 ? -2.f * kPa : -2.f * kPb;
#pragma unroll
                for (int u = 0; u < UA; ++u) {
                    if (u * 256 >= items_eff) break;
                    const unsigned* pr = (const unsigned*)((const char*)ptab + item_prow(u));
                    const unsigned ha = pr[1] + chunkb, hb = pr[3] + chunkb;
                    const unsigned wo[6] = {pr[6], pr[7], pr[8], pr[9], pr[10], pr[11]};
                    f32x4 xa[6], xb[6];
#pragma unroll
                    for (int ii = 0; ii < 6; ++ii) {
                        xa[ii] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ha + wo[ii], 0, 0));
                        xb[ii] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, hb + wo[ii], 0, 0));
                    }
                    const int vr = item_vrow(u);
                    if (vr < 0) continue;
                    wtransform_(smem + vr, c1 * xa[0] + c3 * xb[0], c1 * xa[1] + c3 * xb[1], c1 * xa[2] + c3 * xb[2], c1 * xa[3] + c3 * xb[3],
                                c1 * xa[4] + c3 * xb[4], c1 * xa[5] + c3 * xb[5], true);
                }
            } else
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                if (u * 256 >= items_eff) break;
                const unsigned* pr = (const unsigned*)((const char*)ptab + item_prow(u));
                const unsigned ha = pr[ia] + chunkb, hb = pr[ib] + chunkb, hc = pr[ic] + chunkb, hd = pr[4] + chunkb;
                const unsigned wo[6] = {pr[6], pr[7], pr[8], pr[9], pr[10], pr[11]};
                f32x4 xa[6], xb[6], v[6];
#pragma unroll
                for (int ii = 0; ii < 6; ++ii) {
                    xa[ii] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ha + wo[ii], 0, 0));
                    xb[ii] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, hb + wo[ii], 0, 0));
                }
#pragma unroll
                for (int ii = 0; ii < 6; ++ii) v[ii] = ca * xa[ii] + cb * xb[ii];
                if (ends) {
#pragma unroll
                    for (int ii = 0; ii < 6; ++ii) xa[ii] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, hc + wo[ii], 0, 0));
#pragma unroll
                    for (int ii = 0; ii < 6; ++ii) v[ii] += xa[ii];
                } else {
#pragma unroll
                    for (int ii = 0; ii < 6; ++ii) {
                        xa[ii] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, hc + wo[ii], 0, 0));
                        xb[ii] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, hd + wo[ii], 0, 0));
                    }
#pragma unroll
                    for (int ii = 0; ii < 6; ++ii) v[ii] += cc * xa[ii] + xb[ii];
                }
                const int vr = item_vrow(u);
                if (vr < 0) continue;
                wtransform(smem + vr, v[0], v[1], v[2], v[3], v[4], v[5]);
            }
        }
        // the weight ring is primed per stage, behind the staging (its registers are free for the input rows meanwhile) and
        // ahead of the barrier (whose wait covers the L2 round trip)
        if constexpr (SPLIT) ldu(0, ss * 18);
        else {
#pragma unroll
            for (int j = 0; j < RDB - 1; ++j) ldb(j, xh, 0, j);
        }
        __syncthreads();

        // ---- K loop of the stage: 3 depth taps x (6 xw x 4 cin groups) ----
        f32x4 acc[6][2];
#pragma unroll
        for (int xi = 0; xi < 6; ++xi) {
            acc[xi][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc[xi][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        if constexpr (SPLIT) {
            // one step = (kd, xw): 32 cin x 6 terms x 2 M-blocks = 12 MFMAs of 16 cycles, fed by 3 weight pieces (16 B per lane each, L1/L2;
            // requested a step ahead into the other half of US) and 3 x 2 cell pieces (ds_read_b128; each re-requested right behind its last use)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) ldv(pc, 0);
#pragma unroll 1
            for (int kd = 0; kd < 3; ++kd) {
                const bool last = kd == 2;
                const int tapb = kd * tapstep, tapb_n = last ? tapb : tapb + tapstep;
                const int st0 = (ss * 3 + kd) * 6, st0_n = last ? st0 : st0 + 6;       // the last tap re-requests its own first step (never used)
#pragma unroll
                for (int xw = 0; xw < 6; ++xw) {
                    const int su = xw & 1;
                    const int vo_n = xw < 5 ? tapb + (xw + 1) * kW2Plane : tapb_n;
                    auto mm = [&](int pu, int pv) {
                        acc[xw][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(US[su][pu], VS[pv][0], acc[xw][0], 0, 0, 0);
                        acc[xw][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(US[su][pu], VS[pv][1], acc[xw][1], 0, 0, 0);
                    };
                    mm(0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    ldu(su ^ 1, xw < 5 ? st0 + xw + 1 : st0_n);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(1, 0);
                    mm(2, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    ldv(0, vo_n);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(0, 1);
                    mm(1, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    ldv(1, vo_n);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(0, 2);
                    __builtin_amdgcn_sched_barrier(0);
                    ldv(2, vo_n);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
#pragma unroll
        for (int j = 0; j < RDA - 1; ++j) lda(j, 0, j);
#pragma unroll 1
        for (int kd = 0; kd < 3; ++kd) {
            const bool last = kd == 2;
            const int tapb = kd * tapstep;
            const int tapb_n = last ? tapb : tapb + tapstep;
            const int kd_n = last ? kd : kd + 1;             // the last tap re-requests its own first units (never used)
#pragma unroll
            for (int j = 0; j < SPT; ++j) {
                const int sb = j % RDB, sa = j % RDA;
                const int xi = j >> 2;
                acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(B[sb][0], A[sa][0][0], acc[xi][0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                {
                    const int jb = j + RDB - 1, ja = j + RDA - 1;
                    if (jb < SPT) ldb(jb % RDB, xh, kd, jb);
                    else ldb(jb % RDB, xh, kd_n, jb - SPT);
                    if (ja < SPT) lda(ja % RDA, tapb, ja);
                    else lda(ja % RDA, tapb_n, ja - SPT);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (MB == 2) {
                    acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(B[sb][0], A[sa][1][0], acc[xi][1], 0, 0, 0);
#pragma unroll
                    for (int s = 1; s < 4; ++s) {
                        acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(B[sb][s], A[sa][0][s], acc[xi][0], 0, 0, 0);
                        acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(B[sb][s], A[sa][1][s], acc[xi][1], 0, 0, 0);
                    }
                } else {                                     // one M-block: k-steps alternate between the two accumulators of the xw
                    acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(B[sb][1], A[sa][0][1], acc[xi][1], 0, 0, 0);
                    acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(B[sb][2], A[sa][0][2], acc[xi][0], 0, 0, 0);
                    acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(B[sb][3], A[sa][0][3], acc[xi][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        }

        // ---- fold the stage: t = A_w^T M[xh] (A_w^T = (1,1,1,1,1,0) (0,a,-a,b,-b,0) (0,a2,a2,b2,b2,0) (0,a3,-a3,b3,-b3,1); HM = 2: a = 1, b = 2),
        // then the output transform along H as Y_hr += A_h^T[hr][xh] t: F(2,3) A_h^T = (1,1,1,0) (0,1,-1,-1) (coordinate 2 is staged
        // negated, see the pack), F(4,3) A_h^T = A_w^T ----
        float ch[HM];
        if constexpr (HM == 2) {
            ch[0] = xh < 3 ? 1.f : 0.f;
            ch[1] = xh == 0 ? 0.f : (xh == 1 ? 1.f : -1.f);
        } else {
            const float sg = (xh & 1) ? 1.f : -1.f;          // coordinates 1, 3 = the points +a, +b; 2, 4 = -a, -b
            const float m = xh <= 2 ? kPa : kPb;
            const bool mid = xh >= 1 && xh <= 4;
            ch[0] = xh < 5 ? 1.f : 0.f;
            ch[1] = mid ? sg * m : 0.f;
            ch[2] = mid ? m * m : 0.f;
            ch[3] = mid ? sg * m * m * m : (xh == 5 ? 1.f : 0.f);
        }
        if constexpr (MB == 1) {
#pragma unroll
            for (int xi = 0; xi < 6; ++xi) acc[xi][0] += acc[xi][1];
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            const f32x4 s12 = acc[1][mb] + acc[2][mb], d12 = acc[1][mb] - acc[2][mb];
            const f32x4 s34 = acc[3][mb] + acc[4][mb], d34 = acc[3][mb] - acc[4][mb];
            f32x4 t[4];
            t[0] = acc[0][mb] + s12 + s34;
            if constexpr (HM == 2) {
                t[1] = d12 + 2.f * d34;
                t[2] = s12 + 4.f * s34;
                t[3] = d12 + 8.f * d34 + acc[5][mb];
            } else {
                t[1] = kPa * d12 + kPb * d34;
                t[2] = kPa2 * s12 + kPb2 * s34;
                t[3] = kPa3 * d12 + kPb3 * d34 + acc[5][mb];
            }
#pragma unroll
            for (int wi = 0; wi < 4; ++wi)
#pragma unroll
                for (int hr = 0; hr < HM; ++hr) Y[hr][wi][mb] += ch[hr] * t[wi];
        }
    }
    }   // sources

    if (FDN_DBG_BITS(p) & 8) return;
    // ---- epilogue: lane = cell c of each M-block x cout 16w + 4q .. + 3; HM x 4 voxels per cell ----
    const int lane_e = HM == 2 ? lane : (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int c_e = HM == 2 ? c : (lane_e & 15);
    const int cofs = (HM == 2 ? wave : wave_s) * 16 + (HM == 2 ? q : (lane_e >> 4)) * 4;
    const float slope = p.act == FDN_ACT_RELU ? 0.f : (p.act == FDN_ACT_LEAKY ? p.alpha : 1.f);
    // The outputs leave in GROUPS of HG rows of one M-block, and a group's operand loads (skip / y, or the residual) are requested one
    // group ahead of the stores: the stores of group g may alias the loads of group g + 1 as far as the compiler can tell (skip may BE
    // the output), so left to itself it serialises a memory round trip per group at the end of every tile.  A lane reads exactly the
    // addresses it writes, so hoisting the reads is safe.  F(2,3): two groups = the two M-blocks, i.e. every load ahead of the first
    // store; F(4,3): four groups of 2 rows (forward) or eight of one row (fused dgrad: two operands per voxel) -- what fits beside Y.
    // (test build, bit 32: the epilogue's operand rows come from the first 64 KB of their tensors -- cache hits; wrong results, timing only.
    // It prices what the operands cost BEYOND their instructions: at (8,48^3) a residual costs 0.036 ms of which 0.029 is the latency of rows
    // that come from HBM (tools/abl_epilogue_ops.py).  Tried and removed: pulling those rows into the L2 one stage early with untracked
    // direct-to-LDS loads (4 per lane and operand, no registers) -- vector-memory loads retire in order, so the next counted wait of the
    // wave pays the prefetch's HBM round trip, as much as the epilogue then saves: fwd + residual 0.528 with vs 0.521 ms without.)
    const int opmask = (FDN_DBG_BITS(p) & 32) ? 255 : -1;
    constexpr int HG = HM == 2 ? 2 : (FUSED ? 1 : 2);
    constexpr int GPB = HM / HG, NG = MB * GPB;               // groups per M-block, groups
    int g0[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) g0[mb] = mtab[mb * 16 + c_e];
    if (FUSED) {
        // dgrad on the inner box of the padded grid: voxels strictly inside the volume get exactly one contribution and are finished
        // here (dz_prev = (dgrad + skip) * act'(y)); surface voxels go to the padded scratch for the border fold.  Branch-free per
        // voxel: a surface voxel (or a cell outside the box) reads row 0 of the tensor, which nobody writes in this launch, and
        // discards it; value and destination are selected afterwards.
        int fi[2][HG][4];                     // voxel index into skip / y / dz_prev, or -1 (surface voxel, cell outside the box)
        f32x4 sk[2][HG][4], ym[MASK ? 1 : 2][MASK ? 1 : HG][4];
        fdn_u32x2 mw[2][HG];                  // MASK: the cell row's four sign words of this wave's 16 channels
        const int q_f = HM == 2 ? q : (lane_e >> 4);
        const size_t mplane = (size_t)(HM == 2 ? wave : wave_s) * ((size_t)p.N * p.ID * p.IH * p.IW);
        auto fload = [&](int g, int buf) {
            const int mb = g / GPB, h0 = (g % GPB) * HG;
            const int gf0 = mtab[32 + mb * 16 + c_e];
            const int hw = mtab[64 + mb * 16 + c_e];
            const int ph = hw & 0xffff, pw = hw >> 16;                    // padded coordinates of the cell's first voxel
#pragma unroll
            for (int hr = 0; hr < HG; ++hr) {
                if constexpr (MASK) {                                    // (the row's first voxel is a valid index wherever the cell has an interior depth)
                    const int ri = (g0[mb] >= 0 && gf0 >= 0) ? gf0 + (h0 + hr) * p.IW : 0;
                    mw[buf][hr] = *(const fdn_u32x2*)(p.fmask + mplane + ri);
                }
#pragma unroll
                for (int wi = 0; wi < 4; ++wi) {
                    const int ih = ph + h0 + hr - 1, iw = pw + wi - 1;
                    const bool in = g0[mb] >= 0 && gf0 >= 0 && ih >= 1 && ih <= p.IH - 2 && iw >= 1 && iw <= p.IW - 2;
                    fi[buf][hr][wi] = in ? gf0 + (h0 + hr) * p.IW + wi : -1;
                    const size_t o = (size_t)((in ? fi[buf][hr][wi] : 0) & opmask) * 64 + cofs;
                    sk[buf][hr][wi] = p.fskip ? *(const f32x4*)(p.fskip + o) : (f32x4){0.f, 0.f, 0.f, 0.f};
                    if constexpr (!MASK) ym[buf][hr][wi] = p.fy ? *(const f32x4*)(p.fy + o) : (f32x4){1.f, 1.f, 1.f, 1.f};
                }
            }
        };
        auto fstore = [&](int g, int buf) {
            const int mb = g / GPB, h0 = (g % GPB) * HG;
            if (g0[mb] < 0) return;
#pragma unroll
            for (int hr = 0; hr < HG; ++hr)
#pragma unroll
                for (int wi = 0; wi < 4; ++wi) {
                    const f32x4 z = Y[h0 + hr][wi][mb];
                    const bool in = fi[buf][hr][wi] >= 0;
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        bool pos;
                        if constexpr (MASK) pos = (((wi < 2 ? mw[buf][hr].x : mw[buf][hr].y) >> (16 * (wi & 1) + 4 * q_f + e)) & 1u) != 0;
                        else pos = ym[buf][hr][wi][e] > 0.f;
                        v[e] = in ? (z[e] + sk[buf][hr][wi][e]) * (pos ? 1.f : slope) : z[e];
                    }
                    float* dst = in ? p.fout + (size_t)fi[buf][hr][wi] * 64 + cofs : p.y + (size_t)(g0[mb] + (h0 + hr) * p.OW + wi) * 64 + cofs;
                    *(f32x4*)dst = v;
                }
        };
        fload(0, 0);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g + 1 < NG) fload(g + 1, (g + 1) & 1);
            fstore(g, g & 1);
        }
    } else {
        f32x4 rv[2][HG][4];
        auto rload = [&](int g, int buf) {
            const int mb = g / GPB, h0 = (g % GPB) * HG;
#pragma unroll
            for (int hr = 0; hr < HG; ++hr)
#pragma unroll
                for (int wi = 0; wi < 4; ++wi)
                    rv[buf][hr][wi] = *(const f32x4*)(p.res + (size_t)(((g0[mb] >= 0 ? g0[mb] : 0) + (h0 + hr) * p.OW + wi) & opmask) * 64 + cofs);
        };
        if (p.res) rload(0, 0);
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv = *(const f32x4*)(p.bias + cofs);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (p.res && g + 1 < NG) rload(g + 1, (g + 1) & 1);
            const int mb = g / GPB, h0 = (g % GPB) * HG, buf = g & 1;
            if (g0[mb] < 0) continue;
#pragma unroll
            for (int hr = 0; hr < HG; ++hr) {
                unsigned mlo = 0, mhi = 0;                                // MASK: this lane's 4 bits of the row's four voxels, at their place in the words
#pragma unroll
                for (int wi = 0; wi < 4; ++wi) {
                    f32x4 v = Y[h0 + hr][wi][mb] + bv;
                    if (p.res) v += rv[buf][hr][wi];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], slope * v[e]);          // relu / leaky / none: slope in [0,1]
                    *(f32x4*)(p.y + (size_t)(g0[mb] + (h0 + hr) * p.OW + wi) * 64 + cofs) = v;
                    if constexpr (MASK) {
                        const unsigned nib = (v[0] > 0.f ? 1u : 0u) | (v[1] > 0.f ? 2u : 0u) | (v[2] > 0.f ? 4u : 0u) | (v[3] > 0.f ? 8u : 0u);
                        if (wi < 2) mlo |= nib << (16 * wi); else mhi |= nib << (16 * (wi - 2));
                    }
                }
                if constexpr (MASK) {
                    // the four quarter-waves (cout 4q .. 4q + 3 of the same cell: lanes c, c + 16, c + 32, c + 48) OR their nibbles into the
                    // row's four 16-bit words; quarter 0 stores them (8 B, one plane per wave)
                    const int q_m = HM == 2 ? q : (lane_e >> 4);
                    mlo <<= 4 * q_m; mhi <<= 4 * q_m;
                    mlo |= __shfl_xor(mlo, 16, 64); mhi |= __shfl_xor(mhi, 16, 64);
                    mlo |= __shfl_xor(mlo, 32, 64); mhi |= __shfl_xor(mhi, 32, 64);
                    if (q_m == 0)
                        *(fdn_u32x2*)(p.ymask + (size_t)(HM == 2 ? wave : wave_s) * ((size_t)p.N * p.OD * p.OH * p.OW) + g0[mb] + (h0 + hr) * p.OW) = (fdn_u32x2){mlo, mhi};
                }
            }
        }
    }
}

}  // namespace

// conv3d 3x3x3, 64 -> 64 channels, fp32, NDHWC: 2-D Winograd -- F(2,3) along H on top of F(4,3) along W, direct taps along D --
// on v_mfma_f32_16x16x4_f32.
//
// Same contract as conv64_wino.hip / conv64_mfma.hip (tf.pad(SYMMETRIC) + Conv3D + bias + activation + residual of
// src/Network/SR4DFlowNet.py:93-120 in clamp mode; Conv3DBackpropInputV2 of the same layers in zero mode with the interior
// MirrorPadGrad fused into the epilogue), with a third fewer multiplies than the 1-D kernel: a CELL of 2 x 4 output voxels (h, w)
// is computed from its 4 x 6 input patch as
//     Y = Ah^T [ sum_kd (Gh g Gw^T) .* (Bh^T x Bw) ] Aw             (Lavin & Gray; F(2,3): points 0, +-1, inf; F(4,3): 0, +-1, +-2, inf)
// i.e. 24 products per (kd, cin, cout) for 8 voxels: 3 * 24 / 8 = 9 tap-equivalents per voxel instead of 13.5 (1-D) or 27 (direct).
// On MI355X the fp32 matrix rate equals the fp32 vector rate, so fewer multiplies is the only way past the roofline.
//
// The 24 accumulator tiles a cell needs do not fit beside everything else (24 points x 2 M-blocks x 4 registers = 192), so the
// F(2,3) coordinate xh runs as an OUTER, sequential loop ("stage"): a stage holds the six xw accumulators of its xh only (48
// registers for 32 cells x 16 cout), runs the whole K loop of that xh (3 depth taps x 64 cin), and folds A_w^T M[xh] into the two
// output rows with the F(2,3) coefficients (Y_h0 += c0[xh] t, Y_h1 += c1[xh] t; 64 registers).  Because F(2,3)'s B^T rows have two
// non-zeros, a stage's input is V[xh] = x[ra] +- x[rb] of TWO input rows (xh 0: x0-x2, 1: x1+x2, 2: x1-x2 with the sign folded into
// the packed weights, 3: x1-x3), staged for all 64 input channels at once -- so the four stages take the place of the four cin
// slices of the 1-D kernel (same number of barriers) and read each input row twice (L1/L2 hits).
//
// Work decomposition:
//   * M = cells.  Workgroup (4 waves) = a box tile of td x ch x cw cells (<= 32 cells = 256 voxels) x 64 cout; wave w owns all
//     32 cells (two 16-cell MFMA blocks) x cout [16w, 16w+16): D[cout][cell] orientation, so a lane (cell c = lane & 15,
//     q = lane >> 4) holds cout 16w + 4q .. + 3 of one cell = 16-B stores.
//   * Staging: thread (staged cell-plane r, 16-B channel chunk) loads the 2 x 6 input chunks of its stage (boundary rule through
//     the staging plan: clamp == SYMMETRIC p=1, or zero through the buffer range check), forms x[ra] +- x[rb], applies B_w^T
//     (14 VALU per float) and writes 6 xw planes of rows [cell-plane][64 cin + 32-B pad]: the 288-B row stride makes the 16-row
//     ds_read_b128 lane groups conflict-free.  A tile stages (td + 2) depth planes x ch x cw cells <= 40 rows: 69 KB, 2 workgroups per CU.
//   * Weights: packed stream U[nb][xh][kd][xw][cin/16][lane][4] (fdn_pack_wino2d_one, 72*64*64 floats per layer and direction),
//     read straight from L1/L2 by buffer_load_dwordx4.
//   * K loop of a stage: 3 depth taps x 6 xw x 4 cin groups; one step = 2 ds_read_b128 + 1 buffer_load_b128 feeding 8 MFMAs
//     (2 M-blocks x 4 k-steps, alternating blocks: the 40-cycle dependent latency of v_mfma_f32_16x16x4_f32 is covered);
//     operands through fragment rings refilled inside the MFMA stream, as in the 1-D kernel.
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
    int N, ID, IH, IW, OD, OH, OW;
    int off, zero_mode, act;
    float alpha;
    int dbg;                // ablation bits (test build only): 1 = weight stream stride 0, 4 = no staging, 8 = no epilogue, 128 = no XCD remap
    int obd, obh, obw, ebd, ebh, ebw;      // output box (h extent even, w extent a multiple of 4)
    int td, ch, cw, ntd, nth, ntw;         // tile in (depth planes, cell rows, cell columns) and tile counts
    int cpp, rows, items;                  // cells per plane, staged rows = (td + 2) * cpp, rows * 16
    unsigned mg_cpp, mg_cw;
    unsigned mg_tpn_hi, mg_tpn_lo, mg_thw_hi, mg_thw_lo, mg_ntw_hi, mg_ntw_lo;
};

FDN_HOOK_VAR(int, fdn_conv64_wino2d_dbg, 0);
FDN_HOOK_VAR(int, fdn_conv64_wino2d_tile, 0);          // test build: force the tile, td | ch << 8 | cw << 16 (0 = planner)

constexpr int kW2Rows = 40;                 // staged cell-planes per tile
constexpr int kW2Row = 288;                 // bytes per LDS row: 64 cin + 32-B pad (conflict-free ds_read_b128 over 16 consecutive rows)
constexpr int kW2Plane = kW2Rows * kW2Row + 64;
constexpr int kW2Lds = 6 * kW2Plane + 96 * 4 + kW2Rows * 48;
constexpr int kW2UA = 3;                    // transform items per thread (<= 640 items = 40 rows x 16 chunks)
constexpr int kW2RDB = 6;                   // weight-fragment ring depth
constexpr int kW2RDA = 3;                   // cell-fragment ring depth
constexpr int kW2Dep = 1;                   // staging: items (12 x 16-B loads each) in flight per thread
constexpr unsigned kW2Big = 0x40000000u;    // "reads zero": any sum containing it is >= 2^30 > the sample's bytes

template <bool FUSED>
__global__ __launch_bounds__(256, 2) void conv64_wino2d_kernel(Wino2Args p) {
    constexpr int RDB = kW2RDB, RDA = kW2RDA, UA = kW2UA, SPT = 24;
    static_assert(SPT % RDB == 0 && SPT % RDA == 0, "ring slots must be compile-time");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;               // = cout block
    const int c = lane & 15;
    const int q = lane >> 4;
    int* mtab = (int*)(smem + 6 * kW2Plane);    // [0,32): output index of the cell's first voxel or -1; [32,64): fused-fold index or -1;
                                                // [64,96): (h | w << 16) of the cell's first voxel (fused mode)

    // ---- which tile (scalar multiply-shift divisions, host-made magics; XCD-aware order as in conv64_wino.hip) ----
    const int tiles_per_n = p.ntd * p.nth * p.ntw;
    int b = (int)blockIdx.x;
    if (!(FDN_DBG_BITS(p) & 128)) {
        const int T = p.N * tiles_per_n, qq = T >> 3, r = T & 7, xcd = b & 7;
        b = xcd * qq + min(xcd, r) + (b >> 3);
    }
    const int n = fdn_udiv40(b, p.mg_tpn_hi, p.mg_tpn_lo);
    b -= n * tiles_per_n;
    const int tdi = fdn_udiv40(b, p.mg_thw_hi, p.mg_thw_lo);
    b -= tdi * (p.nth * p.ntw);
    const int thi = fdn_udiv40(b, p.mg_ntw_hi, p.mg_ntw_lo);
    const int p0d = p.obd + tdi * p.td, p0h = p.obh + thi * p.ch * 2, p0w = p.obw + (b - thi * p.ntw) * p.cw * 4;
    const int ng = p.td * p.cpp;

    if (tid < 32) {
        int g = -1, gf = -1, hw = 0;
        if (tid < ng) {
            const int md = fdn_div20(tid, p.mg_cpp);
            const int j = tid - md * p.cpp;
            const int mh = fdn_div20(j, p.mg_cw);
            const int pd = p0d + md, ph = p0h + 2 * mh, pw = p0w + 4 * (j - mh * p.cw);
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
    int abase[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        int m = mb * 16 + c;
        m = m < ng ? m : ng - 1;
        abase[mb] = m * kW2Row + q * 16;
    }

    // ---- staging plan, once per tile, in LDS: per staged cell-plane r the byte offsets (from the sample's first voxel) of its 4 input
    // rows and 6 input columns with the boundary rule applied (kW2Big = reads zero: any sum containing it is out of the buffer's range).
    // ptab[r] = {row 0..3, column 0..5, -, -} (48 B); a transform item adds its channel chunk.  Keeping the plan out of the register
    // file is what lets a thread hold two items' input rows in flight (24 x 16 B) beside the 64 output accumulators.
    unsigned* ptab = (unsigned*)(smem + 6 * kW2Plane + 96 * 4);
    if (tid < p.rows) {
        const int r = tid;
        const int zd = fdn_div20(r, p.mg_cpp);
        const int j = r - zd * p.cpp;
        const int mh = fdn_div20(j, p.mg_cw);
        const int q0d = p0d - 1 + p.off, q0h = p0h - 1 + p.off + 2 * mh, q0w = p0w - 1 + p.off + 4 * (j - mh * p.cw);
        int qd = q0d + zd;
        bool okd = true;
        if (p.zero_mode) okd = (unsigned)qd < (unsigned)p.ID;
        else qd = min(max(qd, 0), p.ID - 1);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
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
            ptab[r * 12 + 4 + ii] = ok ? (unsigned)qw * 256u : kW2Big;
        }
    }
    int vrow[UA], prow[UA];                                   // LDS offsets of the item's output row / plan row; chunk offset
#pragma unroll
    for (int u = 0; u < UA; ++u) {
        const int i = u * 256 + tid;
        const int r = i >> 4;
        vrow[u] = r < p.rows ? r * kW2Row + (i & 15) * 16 : -1;
        prow[u] = (r < p.rows ? r : 0) * 48;
    }
    const unsigned chunkb = (unsigned)(tid & 15) * 16u;
    __syncthreads();                                          // mtab + ptab visible
    const unsigned sample_bytes = (unsigned)(p.ID * p.IH * p.IW) * 256u;
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.x + (size_t)n * p.ID * p.IH * p.IW * 64), 0, sample_bytes, 0x00020000);

    // weight stream: unit (1024 B) index = ((nb*4 + xh)*3 + kd)*24 + xw*4 + g
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.up, 0, 72 * 64 * 64 * 4, 0x00020000);
    const int wvoff = wave * (288 * 1024) + lane * 16;
    const int bmul = (FDN_DBG_BITS(p) & 1) ? 0 : 1024;
    f32x4 A[RDA][2], B[RDB];
    auto ldb = [&](int slot, int xh, int kd, int j) {
        B[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, ((xh * 3 + kd) * 24 + j) * bmul, 0));
    };
    auto lda = [&](int slot, int tapb, int j) {             // tapb: byte offset of the depth tap's rows
        const int o = tapb + (j >> 2) * kW2Plane + (j & 3) * 64;
        A[slot][0] = *(const f32x4*)(smem + abase[0] + o);
        A[slot][1] = *(const f32x4*)(smem + abase[1] + o);
    };

    f32x4 Y[2][4][2];                                        // [output row][output column][M-block]: cout 16w + 4q .. + 3 of that voxel
#pragma unroll
    for (int hr = 0; hr < 2; ++hr)
#pragma unroll
        for (int wi = 0; wi < 4; ++wi)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) Y[hr][wi][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int tapstep = p.cpp * kW2Row;

#pragma unroll 1
    for (int xh = 0; xh < 4; ++xh) {
        if (xh) __syncthreads();                             // everyone finished reading the previous stage's planes
        // ---- stage xh: V = x[ra] + sgn * x[rb], rows (0,2,-) (1,2,+) (1,2,-) (1,3,-); then B_w^T; all 64 cin ----
        {
            const float sgn = xh == 1 ? 1.f : -1.f;
            // a wave whose 64 items of a pass all lie past the tile's last item requests nothing in that pass (640 items: waves 2, 3 of pass 2)
            const int items_eff = ((FDN_DBG_BITS(p) & 4) ? 0 : p.items) - __builtin_amdgcn_readfirstlane(wave) * 64;
            constexpr int DEP = kW2Dep;                       // items in flight per thread
            f32x4 xa[DEP][6], xb[DEP][6];
            auto issue = [&](int u, int buf) {
                const unsigned* pr = (const unsigned*)((const char*)ptab + prow[u]);
                const unsigned ha = (xh == 0 ? pr[0] : pr[1]) + chunkb;
                const unsigned hb = (xh == 3 ? pr[3] : pr[2]) + chunkb;
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
                // B^T of F(4,3): rows (4,0,-5,0,1,0) (0,-4,-4,1,1,0) (0,4,-4,-1,1,0) (0,-2,-1,2,1,0) (0,2,-1,-2,1,0) (0,4,0,-5,0,1)
                const f32x4 x0 = xa[bf][0] + sgn * xb[bf][0], x1 = xa[bf][1] + sgn * xb[bf][1], x2 = xa[bf][2] + sgn * xb[bf][2];
                const f32x4 x3 = xa[bf][3] + sgn * xb[bf][3], x4 = xa[bf][4] + sgn * xb[bf][4], x5 = xa[bf][5] + sgn * xb[bf][5];
                const f32x4 t1 = x4 - 4.f * x2, t2 = x3 - 4.f * x1;
                const f32x4 t3 = x4 - x2, t4 = 2.f * (x3 - x1);
                char* vp = smem + vrow[u];
                *(f32x4*)(vp) = 4.f * x0 - 5.f * x2 + x4;
                *(f32x4*)(vp + kW2Plane) = t1 + t2;
                *(f32x4*)(vp + 2 * kW2Plane) = t1 - t2;
                *(f32x4*)(vp + 3 * kW2Plane) = t3 + t4;
                *(f32x4*)(vp + 4 * kW2Plane) = t3 - t4;
                *(f32x4*)(vp + 5 * kW2Plane) = 4.f * x1 - 5.f * x3 + x5;
            }
        }
        // the weight ring is primed per stage, behind the staging (its registers are free for the input rows meanwhile) and
        // ahead of the barrier (whose wait covers the L2 round trip)
#pragma unroll
        for (int j = 0; j < RDB - 1; ++j) ldb(j, xh, 0, j);
        __syncthreads();

        // ---- K loop of the stage: 3 depth taps x (6 xw x 4 cin groups) ----
        f32x4 acc[6][2];
#pragma unroll
        for (int xi = 0; xi < 6; ++xi) {
            acc[xi][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc[xi][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
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
                acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(B[sb][0], A[sa][1][0], acc[xi][1], 0, 0, 0);
#pragma unroll
                for (int s = 1; s < 4; ++s) {
                    acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(B[sb][s], A[sa][0][s], acc[xi][0], 0, 0, 0);
                    acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(B[sb][s], A[sa][1][s], acc[xi][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- fold the stage: t = A_w^T M[xh] (A_w^T = (1,1,1,1,1,0) (0,1,-1,2,-2,0) (0,1,1,4,4,0) (0,1,-1,8,-8,1)), then the F(2,3)
        // output transform A_h^T = (1,1,1,0) (0,1,-1,-1) as Y_h0 += c0 t, Y_h1 += c1 t ----
        const float c0 = xh < 3 ? 1.f : 0.f;
        const float c1 = xh == 0 ? 0.f : (xh == 1 ? 1.f : -1.f);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const f32x4 s12 = acc[1][mb] + acc[2][mb], d12 = acc[1][mb] - acc[2][mb];
            const f32x4 s34 = acc[3][mb] + acc[4][mb], d34 = acc[3][mb] - acc[4][mb];
            f32x4 t[4];
            t[0] = acc[0][mb] + s12 + s34;
            t[1] = d12 + 2.f * d34;
            t[2] = s12 + 4.f * s34;
            t[3] = d12 + 8.f * d34 + acc[5][mb];
#pragma unroll
            for (int wi = 0; wi < 4; ++wi) {
                Y[0][wi][mb] += c0 * t[wi];
                Y[1][wi][mb] += c1 * t[wi];
            }
        }
    }

    if (FDN_DBG_BITS(p) & 8) return;
    // ---- epilogue: lane = cell c of each M-block x cout 16w + 4q .. + 3; 8 voxels per cell ----
    const int cofs = wave * 16 + q * 4;
    const float slope = p.act == FDN_ACT_RELU ? 0.f : (p.act == FDN_ACT_LEAKY ? p.alpha : 1.f);
    // Both M-blocks' operand loads (skip / y, or the residual) are requested before the first store: the stores of block 0 may alias
    // the loads of block 1 as far as the compiler can tell (skip may BE the output), so left to itself it serialises two memory round
    // trips at the end of every tile.  A lane reads exactly the addresses it writes, so hoisting the reads is safe.
    int g0[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) g0[mb] = mtab[mb * 16 + c];
    if (FUSED) {
        // dgrad on the inner box of the padded grid: voxels strictly inside the volume get exactly one contribution and are finished
        // here (dz_prev = (dgrad + skip) * act'(y)); surface voxels go to the padded scratch for the border fold.  Branch-free per
        // voxel: a surface voxel (or a cell outside the box) reads row 0 of the tensor, which nobody writes in this launch, and
        // discards it; value and destination are selected afterwards.
        int fi[2][2][4];                      // voxel index into skip / y / dz_prev, or -1 (surface voxel, cell outside the box)
        f32x4 sk[2][2][4], ym[2][2][4];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            const int gf0 = mtab[32 + mb * 16 + c];
            const int hw = mtab[64 + mb * 16 + c];
            const int ph = hw & 0xffff, pw = hw >> 16;                    // padded coordinates of the cell's first voxel
#pragma unroll
            for (int hr = 0; hr < 2; ++hr)
#pragma unroll
                for (int wi = 0; wi < 4; ++wi) {
                    const int ih = ph + hr - 1, iw = pw + wi - 1;
                    const bool in = g0[mb] >= 0 && gf0 >= 0 && ih >= 1 && ih <= p.IH - 2 && iw >= 1 && iw <= p.IW - 2;
                    fi[mb][hr][wi] = in ? gf0 + hr * p.IW + wi : -1;
                    const size_t o = (size_t)(in ? fi[mb][hr][wi] : 0) * 64 + cofs;
                    sk[mb][hr][wi] = p.fskip ? *(const f32x4*)(p.fskip + o) : (f32x4){0.f, 0.f, 0.f, 0.f};
                    ym[mb][hr][wi] = p.fy ? *(const f32x4*)(p.fy + o) : (f32x4){1.f, 1.f, 1.f, 1.f};
                }
        }
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            if (g0[mb] < 0) continue;
#pragma unroll
            for (int hr = 0; hr < 2; ++hr)
#pragma unroll
                for (int wi = 0; wi < 4; ++wi) {
                    const f32x4 z = Y[hr][wi][mb];
                    const bool in = fi[mb][hr][wi] >= 0;
                    f32x4 v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = in ? (z[e] + sk[mb][hr][wi][e]) * (ym[mb][hr][wi][e] > 0.f ? 1.f : slope) : z[e];
                    float* dst = in ? p.fout + (size_t)fi[mb][hr][wi] * 64 + cofs : p.y + (size_t)(g0[mb] + hr * p.OW + wi) * 64 + cofs;
                    *(f32x4*)dst = v;
                }
        }
    } else {
        f32x4 rv[2][2][4];
        if (p.res) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int hr = 0; hr < 2; ++hr)
#pragma unroll
                    for (int wi = 0; wi < 4; ++wi)
                        rv[mb][hr][wi] = *(const f32x4*)(p.res + (size_t)((g0[mb] >= 0 ? g0[mb] : 0) + hr * p.OW + wi) * 64 + cofs);
        }
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) bv = *(const f32x4*)(p.bias + cofs);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
            if (g0[mb] < 0) continue;
#pragma unroll
            for (int hr = 0; hr < 2; ++hr)
#pragma unroll
                for (int wi = 0; wi < 4; ++wi) {
                    f32x4 v = Y[hr][wi][mb] + bv;
                    if (p.res) v += rv[mb][hr][wi];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], slope * v[e]);          // relu / leaky / none: slope in [0,1]
                    *(f32x4*)(p.y + (size_t)(g0[mb] + hr * p.OW + wi) * 64 + cofs) = v;
                }
        }
    }
}

__global__ void pack_conv64_wino2d_kernel(const float* __restrict__ w, float* __restrict__ uf, float* __restrict__ ud) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < 72 * 64 * 64) fdn_pack_wino2d_one(w, uf, ud, idx);
}

struct Wino2Plan { int td, ch, cw; double cost; };

// tile choice: every tile costs the MFMA time of 32 cells whatever its fill, plus the staging work of its rows and a fixed prologue /
// epilogue; the launch ends with the busiest CU (2 co-resident workgroups per CU share the matrix pipe, so work per CU = its tiles).
Wino2Plan wino2d_plan(int N, int ed, int ech, int ecw) {
    Wino2Plan best{1, 1, 1, 1e30};
    for (int ch = 1; ch <= ech && ch <= 4; ++ch)
        for (int cw = 1; cw <= ecw && ch * cw <= 4; ++cw)
            for (int td = 1; td <= ed && td * ch * cw <= 32; ++td) {
                const int rows = (td + 2) * ch * cw;
                if (rows > kW2Rows) continue;
                const double tiles = (double)N * ((ed + td - 1) / td) * ((ech + ch - 1) / ch) * ((ecw + cw - 1) / cw);
                const double per_tile = 32.0 + 0.15 * rows + 2.0;
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
bool fdn_conv64_wino2d_ok(int ebd, int ebh, int ebw, int ID, int IH, int IW) {
    return ebd > 0 && ebh >= 2 && (ebh & 1) == 0 && ebw >= 4 && (ebw & 3) == 0 && (long long)ID * IH * IW <= (1ll << 22);
}

int fdn_conv64_wino2d_launch(const float* x, const float* upack2, const float* bias, const float* residual, float* y,
                             const float* fskip, const float* fy, float* fout, int N, int ID, int IH, int IW, int OD, int OH,
                             int OW, int obd, int obh, int obw, int ebd, int ebh, int ebw, int off, int zero_mode, int act,
                             float alpha, hipStream_t s) {
    FDN_REQUIRE(fdn_conv64_wino2d_ok(ebd, ebh, ebw, ID, IH, IW), "conv64 (2-D winograd): box %dx%dx%d of a %dx%dx%d grid is not supported",
                ebd, ebh, ebw, ID, IH, IW);
    FDN_REQUIRE(!fout || (zero_mode && off == -1 && obd == 1 && obh == 1 && obw == 1), "conv64 (2-D winograd): the fused fold belongs to the inner box of a padded dgrad");
    Wino2Args a;
    a.x = x; a.up = upack2; a.bias = bias; a.res = residual; a.y = y; a.fskip = fskip; a.fy = fy; a.fout = fout;
    a.N = N; a.ID = ID; a.IH = IH; a.IW = IW; a.OD = OD; a.OH = OH; a.OW = OW;
    a.off = off; a.zero_mode = zero_mode; a.act = act; a.alpha = alpha; a.dbg = fdn_conv64_wino2d_dbg;
    a.obd = obd; a.obh = obh; a.obw = obw; a.ebd = ebd; a.ebh = ebh; a.ebw = ebw;
    const int ech = ebh / 2, ecw = ebw / 4;
    Wino2Plan pl = wino2d_plan(N, ebd, ech, ecw);
    if (fdn_conv64_wino2d_tile) {
        pl.td = fdn_conv64_wino2d_tile & 255; pl.ch = (fdn_conv64_wino2d_tile >> 8) & 255; pl.cw = (fdn_conv64_wino2d_tile >> 16) & 255;
        FDN_REQUIRE(pl.td >= 1 && pl.ch >= 1 && pl.cw >= 1 && pl.td * pl.ch * pl.cw <= 32 && (pl.td + 2) * pl.ch * pl.cw <= kW2Rows,
                    "conv64 (2-D winograd): forced tile %dx%dx%d does not fit", pl.td, pl.ch, pl.cw);
    }
    a.td = pl.td; a.ch = pl.ch; a.cw = pl.cw;
    a.ntd = (ebd + pl.td - 1) / pl.td; a.nth = (ech + pl.ch - 1) / pl.ch; a.ntw = (ecw + pl.cw - 1) / pl.cw;
    a.cpp = pl.ch * pl.cw; a.rows = (pl.td + 2) * a.cpp; a.items = a.rows * 16;
    a.mg_cpp = fdn_magic20(a.cpp); a.mg_cw = fdn_magic20(pl.cw);
    fdn_magic40(a.ntd * a.nth * a.ntw, &a.mg_tpn_hi, &a.mg_tpn_lo);
    fdn_magic40(a.nth * a.ntw, &a.mg_thw_hi, &a.mg_thw_lo);
    fdn_magic40(a.ntw, &a.mg_ntw_hi, &a.mg_ntw_lo);
    const long long blocks = (long long)N * a.ntd * a.nth * a.ntw;
    FDN_REQUIRE(blocks < (1ll << 31), "conv64 (2-D winograd): too many tiles");
    if (fout) {
        if (int rc = fdn_func_max_lds((const void*)conv64_wino2d_kernel<true>, kW2Lds, "conv64_wino2d")) return rc;
        hipLaunchKernelGGL((conv64_wino2d_kernel<true>), dim3((unsigned)blocks), dim3(256), kW2Lds, s, a);
    } else {
        if (int rc = fdn_func_max_lds((const void*)conv64_wino2d_kernel<false>, kW2Lds, "conv64_wino2d")) return rc;
        hipLaunchKernelGGL((conv64_wino2d_kernel<false>), dim3((unsigned)blocks), dim3(256), kW2Lds, s, a);
    }
    FDN_CHECK_LAUNCH("conv64_wino2d_kernel");
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
#endif

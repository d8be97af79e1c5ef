// Weight gradient of the 3x3x3 64 -> 64 conv with bf16 activations / activation gradients, fp32 result
// (bf16 variant of wgrad64_mfma.hip; Conv3DBackpropFilterV2 behind tape.gradient, TrainerController.py:223):
//   dW[a,b,c][ci][co] = sum_{n,o} x[n, clamp(o + (a,b,c) - 1)][ci] * dz[n,o][co]
//
// Same decomposition as the fp32 kernel: grid = (S splits of the tile list) x (3 depth taps a); a workgroup (4 waves =
// 2x2 quadrants of (ci,co)) keeps 9 accumulators (b,c) x 32x32 in registers across all of its tiles, partial sums go to
// workspace[S][27][64][64] and wgrad64_reduce_kernel sums over S.
//
// v_mfma_f32_32x32x16_bf16 contracts 16 voxels per instruction, and both operands are "k-strided" in the NDHWC LDS image
// (a lane needs 8 consecutive VOXELS of one channel).  gfx950's transposing LDS read delivers exactly that:
// ds_read_b64_tr_b16 hands lane c of a 16-lane group the 4 voxels (rows) of channel c out of a 4-voxel x 16-channel block.
//   * k-step = the 16 voxels (2 h-rows x 8 w) of one (d, h-pair); lanes 0-31 take h-row 0, lanes 32-63 h-row 1.
//   * dz: 2 transposing reads (w 0-3, 4-7) -> B operand, shared by the 9 taps.
//   * x : per b, 3 transposing reads of the halo row (w 0-3, 4-7, 8-11); the operands of the three c taps are 16-bit
//     funnel shifts of that 12-voxel window (c = 1: 4 x v_alignbit; c = 0, 2: register selection).
//     11 LDS reads + 12 VALU per 9 MFMAs instead of 20 reads; the b = 2 group of an h-pair is kept as the b = 0 group of the
//     next one (same rows, same lanes): 8.75 reads per 9 MFMAs on average.
//   * LDS image: 128 B per voxel, the two 64-B halves swapped on rows with bit 1 set -> any 4 consecutive rows x 64 B
//     cover all 64 banks (conflict-free transposing reads); every k-step offset is a multiple of 4 rows, so the per-lane
//     addresses are computed once and the steps are immediates.
#include "fdn_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short v4i16 __attribute__((ext_vector_type(4)));

__global__ void wgrad64_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int S);   // wgrad64_mfma.hip

struct Wgrad64BfArgs {
    const uint16_t* x;
    const uint16_t* dz;
    float* partial;
    int N, D, H, W;
    int ntd, nth, ntw, ntiles, S;
    int seg_len, nseg;       // LDS-DMA kernel: walk units are (n, th, tw) columns cut into nseg depth segments of seg_len planes
    unsigned bytes;          // size of x (= of dz) in bytes (the LDS-DMA kernel addresses both through buffer resources: < 4 GB)
    int dbg;                 // ablation bits (test build): 1 = no staging loads, 2 = no LDS operand reads, 4 = no barrier, 8 = every tile reads the same cache-resident rows
};

FDN_HOOK_VAR(int, fdn_wgrad64bf_dbg, 0);
FDN_HOOK_VAR(int, fdn_wgrad64bf_variant, 0);      // test build: 1 = the register-staged kernel (two planes per tile) for every size

namespace {
constexpr int TD = 2, TH = 8, TW = 8;
constexpr int XH = TH + 2, XW = TW + 2;
constexpr int XROWS = TD * XH * XW;      // 200
constexpr int ZROWS = TD * TH * TW;      // 128
constexpr int XPAD = XROWS + 2;          // the w 8-11 read of the last halo row runs 2 rows past the image
constexpr int LUT_ROWS = 7 * 32 + 4 * 32;   // XP*32 x rows + ZP*32 dz rows
constexpr int LDS_BYTES = (XPAD + ZROWS) * 128 + LUT_ROWS * 4;
}  // namespace

__device__ __forceinline__ u32x2 tr_read(const char* lds_addr) {
    const v4i16 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4i16 __attribute__((address_space(3)))*)lds_addr);
    return __builtin_bit_cast(u32x2, r);
}

__global__ __launch_bounds__(256, 2) void wgrad64_bf16_kernel(Wgrad64BfArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs = smem;
    char* zs = smem + XPAD * 128;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int kh = lane >> 5;
    const int mq = wave & 1, nq = wave >> 1;
    // 1-D grid of 3 S workgroups (S % 8 == 0).  Workgroup ids are dealt round-robin to the 8 XCDs: the three depth-tap
    // workgroups of one tile walk are consecutive ids of the SAME XCD (they read the same x / dz tiles at about the same
    // time -> two of the three reads hit that XCD's L2), and an XCD's S/8 walks cover one contiguous eighth of the tile list
    // (neighbouring tiles share halo rows).
    const int xcd = blockIdx.x & 7, rr = blockIdx.x >> 3;
    const int a = rr % 3;            // kernel-depth tap
    const int kk = rr / 3;           // walk index inside the XCD, 0 .. S/8 - 1
    const int split = kk * 8 + xcd;
    const int spx = p.S >> 3;        // walks per XCD
    const int tq = p.ntiles >> 3, trem = p.ntiles & 7;
    const int t_begin = xcd * tq + (xcd < trem ? xcd : trem), t_end = t_begin + tq + (xcd < trem ? 1 : 0);

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ---- per-lane addresses of the transposing reads at k-step (d 0, h-pair 0) ----
    // lane: 16-lane group g = lane >> 4 (h-row g >> 1, channel block 16 (g & 1)), p = lane & 15 supplies the 8 bytes
    // [row + (p >> 2)][16 (g & 1) + 4 (p & 3) ...] of its group's 4 x 16 block.
    const int g = lane >> 4, pq = lane & 15;
    const int chan_b = (g & 1) * 32 + (pq & 3) * 8;               // byte offset inside the wave's 64-B half
    int xoff[3][3], zoff[2];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int row = ((g >> 1) + b) * XW + 4 * q + (pq >> 2);
            xoff[b][q] = row * 128 + (((mq ^ (row >> 1)) & 1) << 6) + chan_b;
        }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = (g >> 1) * TW + 4 * q + (pq >> 2);
        zoff[q] = row * 128 + (((nq ^ (row >> 1)) & 1) << 6) + chan_b;
    }

    const int tiles_per_n = p.ntd * p.nth * p.ntw;
    const int c16 = tid & 7;    // 16-B chunk within a 128-B row
    const int rsub = tid >> 3;  // 0..31

    // tile staging is software-pipelined through registers (as in the fp32 kernel): the K loop reads only LDS
    constexpr int XP = (XROWS + 31) / 32, ZP = (ZROWS + 31) / 32;
    u32x4 xv[XP], zv[ZP];
    // Row -> element-offset table for tiles that lie completely inside the volume (no clamping): one LDS read replaces
    // the ~25 integer instructions of the row decode, which otherwise sit between the barrier and the first MFMA of a tile
    int* lut = (int*)(smem + (XPAD + ZROWS) * 128);
    for (int i = tid; i < LUT_ROWS; i += 256) {
        int v;
        if (i < XP * 32) {
            const int r = i < XROWS ? i : XROWS - 1;
            const int zd = r / (XH * XW), r2 = r - zd * (XH * XW), zh = r2 / XW;
            v = ((zd * p.H + zh) * p.W + (r2 - zh * XW)) * 64;
        } else {
            const int r = i - XP * 32;
            const int zd = r / (TH * TW), r2 = r - zd * (TH * TW), zh = r2 / TW;
            v = ((zd * p.H + zh) * p.W + (r2 - zh * TW)) * 64;
        }
        lut[i] = v;
    }
    __syncthreads();
    auto prefetch = [&](int tile) {
        int b = tile;
        const int n = b / tiles_per_n;
        b -= n * tiles_per_n;
        const int tdi = b / (p.nth * p.ntw);
        b -= tdi * (p.nth * p.ntw);
        const int thi = b / p.ntw;
        const int p0d = tdi * TD, p0h = thi * TH, p0w = (b - thi * p.ntw) * TW;
        const size_t vox_n = (size_t)n * p.D * p.H * p.W;
        if (p0d + a - 1 >= 0 && p0d + TD + a - 2 < p.D && p0h >= 1 && p0h + TH < p.H && p0w >= 1 && p0w + TW < p.W) {
            // interior tile (uniform branch): addresses = tile origin + table
            const uint16_t* xb = p.x + (vox_n + ((size_t)(p0d + a - 1) * p.H + (p0h - 1)) * p.W + (p0w - 1)) * 64 + c16 * 8;
            const uint16_t* zb = p.dz + (vox_n + ((size_t)p0d * p.H + p0h) * p.W + p0w) * 64 + c16 * 8;
#pragma unroll
            for (int u = 0; u < XP; ++u) xv[u] = *(const u32x4*)(xb + lut[u * 32 + rsub]);
#pragma unroll
            for (int u = 0; u < ZP; ++u) zv[u] = *(const u32x4*)(zb + lut[XP * 32 + u * 32 + rsub]);
            return;
        }
#pragma unroll
        for (int u = 0; u < XP; ++u) {          // x rows, edge clamp applied here
            int r = u * 32 + rsub;
            r = r < XROWS ? r : XROWS - 1;
            const int zd = r / (XH * XW);
            const int r2 = r - zd * (XH * XW);
            const int zh = r2 / XW;
            const int qd = min(max(p0d + zd + a - 1, 0), p.D - 1);
            const int qh = min(max(p0h + zh - 1, 0), p.H - 1);
            const int qw = min(max(p0w + (r2 - zh * XW) - 1, 0), p.W - 1);
            xv[u] = *(const u32x4*)(p.x + (vox_n + ((size_t)qd * p.H + qh) * p.W + qw) * 64 + c16 * 8);
        }
#pragma unroll
        for (int u = 0; u < ZP; ++u) {          // dz rows, zero outside the volume
            const int r = u * 32 + rsub;
            const int zd = r / (TH * TW);
            const int r2 = r - zd * (TH * TW);
            const int zh = r2 / TW;
            const int qd = p0d + zd, qh = p0h + zh, qw = p0w + (r2 - zh * TW);
            const bool ok = qd < p.D && qh < p.H && qw < p.W;
            const int cd = min(qd, p.D - 1), ch = min(qh, p.H - 1), cw = min(qw, p.W - 1);
            zv[u] = *(const u32x4*)(p.dz + (vox_n + ((size_t)cd * p.H + ch) * p.W + cw) * 64 + c16 * 8);
            if (!ok) zv[u] = (u32x4){0u, 0u, 0u, 0u};
        }
    };
    if (t_begin + kk < t_end) prefetch(t_begin + kk);
    for (int tile = t_begin + kk; tile < t_end; tile += spx) {
        __syncthreads();   // previous tile fully consumed
#pragma unroll
        for (int u = 0; u < XP; ++u) {
            const int r = u * 32 + rsub;
            if (r < XROWS) *(u32x4*)(xs + r * 128 + ((c16 ^ (((r >> 1) & 1) << 2)) << 4)) = xv[u];
        }
#pragma unroll
        for (int u = 0; u < ZP; ++u) {
            const int r = u * 32 + rsub;
            *(u32x4*)(zs + r * 128 + ((c16 ^ (((r >> 1) & 1) << 2)) << 4)) = zv[u];
        }
        __syncthreads();
        if (tile + spx < t_end) prefetch(tile + spx);

        // ---- 8 k-steps of 16 voxels: (d, h-pair), each 3 groups (b) of 3 MFMAs.  The transposing reads of group q + 1 are
        // issued before the MFMAs of group q (LDS latency is ~3 MFMA slots; hipcc on its own issues them right before use)
        // Halo-row ring (round 3): the operands of tap b = 2 at h-pair hp (rows 2hp+2 on lanes 0-31, 2hp+3 on lanes 32-63) ARE the
        // operands of tap b = 0 at h-pair hp+1, in the same lanes -- that group is read once and kept: 8 instead of 11 transposing
        // reads per k-step for three of the four h-pairs of a plane (8.75 on average; A/B on one box at (4,128^3): 1.755 -> 1.72 ms).  The loop is fully
        // unrolled, so the slot a group lives in is a compile-time function of q (two slots suffice: a kept group is consumed twice
        // in a row).  Also priced: forming the b = 1 group from the halves of b = 0 and b = 2 (v_permlane32_swap + select: 5 reads per
        // k-step) -- the swap destroys both sources, the copies push the kernel from 256 VGPRs into 34 spills; not kept.
        u32x2 gq[2][3], zq[2][2];
        auto reuse = [](int q) { return q % 3 == 0 && (q / 3) % (TH / 2) != 0; };      // b == 0 and not the first h-pair of a plane
        auto slot = [&](int q) { int s_ = 0; for (int i = 1; i <= q; ++i) if (!reuse(i)) s_ ^= 1; return s_; };
        auto issue = [&](int q, u32x2 (&g)[3], u32x2 (&z)[2]) {       // q = (kd * 4 + hp) * 3 + b
            const int ks = q / 3, b = q % 3, kd = ks / (TH / 2), hp = ks % (TH / 2);
            const int dX = (kd * XH * XW + hp * 2 * XW) * 128;
            if (!reuse(q)) {
                g[0] = tr_read(xs + xoff[b][0] + dX); g[1] = tr_read(xs + xoff[b][1] + dX); g[2] = tr_read(xs + xoff[b][2] + dX);
            }
            if (b == 0) {
                const int dZ = (kd * TH * TW + hp * 2 * TW) * 128;
                z[0] = tr_read(zs + zoff[0] + dZ); z[1] = tr_read(zs + zoff[1] + dZ);
            }
        };
        issue(0, gq[0], zq[0]);
        bf16x8 bv;
#pragma unroll
        for (int q = 0; q < TD * (TH / 2) * 3; ++q) {
            const int b = q % 3;
            if (q + 1 < TD * (TH / 2) * 3) issue(q + 1, gq[slot(q + 1)], zq[((q + 1) / 3) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const u32x2 g0 = gq[slot(q)][0], g1 = gq[slot(q)][1], g2 = gq[slot(q)][2];
            if (b == 0) {
                const u32x2 z0 = zq[(q / 3) & 1][0], z1 = zq[(q / 3) & 1][1];
                bv = __builtin_bit_cast(bf16x8, (u32x4){z0.x, z0.y, z1.x, z1.y});
            }
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, (u32x4){g0.x, g0.y, g1.x, g1.y});
            const bf16x8 a1 = __builtin_bit_cast(
                bf16x8, (u32x4){__builtin_amdgcn_alignbit(g0.y, g0.x, 16), __builtin_amdgcn_alignbit(g1.x, g0.y, 16),
                                __builtin_amdgcn_alignbit(g1.y, g1.x, 16), __builtin_amdgcn_alignbit(g2.x, g1.y, 16)});
            const bf16x8 a2 = __builtin_bit_cast(bf16x8, (u32x4){g0.y, g1.x, g1.y, g2.x});
            acc[b * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bv, acc[b * 3 + 0], 0, 0, 0);
            acc[b * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bv, acc[b * 3 + 1], 0, 0, 0);
            acc[b * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bv, acc[b * 3 + 2], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
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


// ---------------------------------------------------------------------------------------------------------------------
// LDS-DMA variant (round 4): the same contraction, staged by `buffer_load_dwordx4 ... lds` into a ring of THREE one-plane
// tiles.  The register-staged kernel above holds the next tile in 44 VGPRs, can therefore only look ONE tile ahead, and
// spends a barrier + 11 ds_write_b128 + a barrier per tile between its MFMA phases (MFMA busy 0.55 at (4,128^3)).  Here
//   * tile = 1 x 8 x 8 voxels: 100 x halo rows (+ 4 rows the last w 8-11 read runs over) + 64 dz rows = 21 pieces of 1 KB
//     (8 rows x 128 B, the lane-linear image of one DMA instruction); three buffers = 63 KB, two workgroups per CU;
//   * the LDS image is the register kernel's (64-B halves swapped on rows with bit 1 set): lane (row, slot) of a piece
//     fetches chunk slot ^ (row bit 1) << 2, i.e. the swizzle moves to the source side;
//   * tile i + 2 is requested at the top of iteration i, right behind the single barrier of the iteration (which also
//     says that buffer (i + 2) % 3 = (i - 1) % 3 is no longer read), and awaited two iterations later with
//     s_waitcnt vmcnt(<pieces of tile i + 1>): two tiles of latency cover, no data registers, no ds_write;
//   * the DMA is issued from inline asm (fdn_lds_dma16_untracked): hipcc puts s_waitcnt vmcnt(0) between a BUILTIN LDS-DMA
//     and the next ds_read_b64_tr_b16, which would wait for the newest request at the top of every K loop;
//   * addresses are straight-line per piece (box row / column + tile origin, clamped for x, past the range for dz -> the
//     DMA stores zeros): an interior fast path cost more scalar branches than it saved;
//   * walk units are (n, th, tw) columns cut into depth segments and walked along d (see the kernel): the three depth-tap
//     workgroups of a walk share x planes and dz tiles in their XCD's L2.
// Measured at (4,128^3), same box: 1.56 ms vs 1.63 ms (tools/abl_wgrad_bf16.py); cfg4 step 63.9 -> 61.3 ms.
namespace {
constexpr int DXROWS = XH * XW;                    // 100 halo rows of one plane
constexpr int DXSLOTS = (DXROWS + 2 + 7) / 8;      // 13 pieces
constexpr int DZSLOTS = TH * TW / 8;               // 8 pieces
constexpr int DSLOTS = DXSLOTS + DZSLOTS;          // 21
constexpr int DBUFB = DSLOTS * 1024;
constexpr int DNBUF = 3;
constexpr int DJ = (DSLOTS + 3) / 4;               // pieces per wave: 6 (wave 0) or 5
constexpr int DLDS_BYTES = DNBUF * DBUFB;
static_assert(DSLOTS == 21 && DJ == 6, "the s_waitcnt vmcnt(n) of the tile loop counts 6 pieces on wave 0 and 5 on waves 1-3");
}  // namespace

// (a __device__ body + thin __global__ wrappers: one layer per launch, or several layers of one grid in ONE launch -- wgrad64_bf16_dma_batch_kernel
// below.  block = the workgroup's index among the 3 S workgroups of its layer; 3 S is a multiple of 8, so block & 7 is still its XCD.)
__device__ __forceinline__ void wgrad64_bf16_dma_body(const Wgrad64BfArgs& p, const int block, char* const smem) {
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const int li = lane & 31;
    const int kh = lane >> 5;
    const int mq = wave & 1, nq = wave >> 1;
    const int xcd = block & 7, rr = block >> 3;
    const int a = rr % 3;            // kernel-depth tap
    const int kk = rr / 3;           // walk index inside the XCD, 0 .. S/8 - 1
    const int split = kk * 8 + xcd;
    const int spx = p.S >> 3;        // walks per XCD
    // Walk units: (n, th, tw) columns cut into depth segments, walked along d.  The three depth-tap workgroups of a walk sit on the
    // same column at the same time: tap a reads x plane d + a - 1, so a plane requested by tap 2 is requested again by tap 1 one
    // tile later and by tap 0 two tiles later (L2 hits), and all three share the dz tile.  The concurrent walks of an XCD are on
    // DIFFERENT columns (h / w neighbours), not on one column at plane-stride distances.
    const int nunits = p.N * p.nth * p.ntw * p.nseg;
    const int tq = nunits >> 3, trem = nunits & 7;
    const int u_begin = xcd * tq + (xcd < trem ? xcd : trem), u_end = u_begin + tq + (xcd < trem ? 1 : 0);
    const int u0 = u_begin + kk;
    int nt = 0;
    for (int u = u0; u < u_end; u += spx) { const int sg = u % p.nseg; nt += min(p.seg_len, p.D - sg * p.seg_len); }

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // ---- per-lane addresses of the transposing reads at h-pair 0 (see the register kernel) ----
    const int g = lane >> 4, pq = lane & 15;
    const int chan_b = (g & 1) * 32 + (pq & 3) * 8;
    int xoff[3][3], zoff[2];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int row = ((g >> 1) + b) * XW + 4 * q + (pq >> 2);
            xoff[b][q] = row * 128 + (((mq ^ (row >> 1)) & 1) << 6) + chan_b;
        }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = (g >> 1) * TW + 4 * q + (pq >> 2);
        zoff[q] = DXSLOTS * 1024 + row * 128 + (((nq ^ (row >> 1)) & 1) << 6) + chan_b;
    }

    // ---- this thread's part of the (up to) DJ pieces of its wave: piece 4 j + wave; x pieces 0..12, dz pieces 13..20.
    // Straight-line address arithmetic per piece (row / column of the box + tile origin, clamped for x, range-checked for dz:
    // 6 VALU) instead of an interior fast path: the branches of the fast path cost more scalar instructions than they saved. ----
    int zh1[DJ], zw1[DJ];        // row / column inside the box, halo shift (-1) included for x
    unsigned cb[DJ];             // chunk byte offset
#pragma unroll
    for (int j = 0; j < DJ; ++j) {
        const int slot = 4 * j + wave;
        const int prow = (lane >> 3) + 8 * (slot < DXSLOTS ? slot : slot - DXSLOTS);     // row of the LDS image (of its operand)
        const int c = (lane & 7) ^ (((prow >> 1) & 1) << 2);                           // chunk that lives at this position
        int zh, zw;
        if (slot < DXSLOTS) { const int r = prow < DXROWS ? prow : DXROWS - 1; zh = r / XW; zw = r - zh * XW - 1; zh -= 1; }
        else { zh = prow / TW; zw = prow - zh * TW; }
        zh1[j] = zh; zw1[j] = zw;
        cb[j] = (unsigned)(c * 16);
    }
    const fdn_i32x4 xrs = fdn_raw_rsrc(p.x, p.bytes), zrs = fdn_raw_rsrc(p.dz, p.bytes);
    const unsigned lds0 = fdn_lds_addr(smem);

    // ---- cursor: unit u = ((n * nth + th) * ntw + tw) * nseg + seg, plane td inside the segment; units advance by spx, decoded
    // incrementally (scalar work only) ----
    const int per_h = p.ntw * p.nseg, per_n = p.nth * per_h;
    int tn, th, tw, sg, td, dend, sn, sh, sw, ss;
    {
        int b = u0;
        tn = b / per_n; b -= tn * per_n;
        th = b / per_h; b -= th * per_h;
        tw = b / p.nseg; sg = b - tw * p.nseg;
        b = spx;
        sn = b / per_n; b -= sn * per_n;
        sh = b / per_h; b -= sh * per_h;
        sw = b / p.nseg; ss = b - sw * p.nseg;
        td = sg * p.seg_len; dend = min(td + p.seg_len, p.D);
    }
    // requests the cursor tile into buffer `buf` and moves the cursor on
    auto request = [&](int buf) {
        if (FDN_DBG_BITS(p) & 1) return;
        const bool fixed = (FDN_DBG_BITS(p) & 8) != 0;          // ablation: every request reads the same (cache-resident) tile
        const int p0h = (fixed ? 1 : th) * TH, p0w = (fixed ? 1 + (split & 7) : tw) * TW;
        const int tdd = fixed ? 1 : td, tnn = fixed ? 0 : tn;
        const int qd = min(max(tdd + a - 1, 0), p.D - 1);
        const unsigned xplane = (unsigned)((tnn * p.D + qd) * p.H * p.W) * 128u;
        const unsigned zplane = (unsigned)((tnn * p.D + tdd) * p.H * p.W) * 128u;
        const unsigned dst = lds0 + (unsigned)(buf * DBUFB);
#pragma unroll
        for (int j = 0; j < DJ; ++j) {
            // piece 4 j + wave: x for j < 3, x (wave 0) / dz for j = 3, dz for j = 4, dz (wave 0) / none for j = 5
            const bool isx = 4 * j + 3 < DXSLOTS || (4 * j < DXSLOTS && 4 * j + wave_s < DXSLOTS);
            const bool isz = !isx && (4 * j + 3 < DSLOTS || (4 * j < DSLOTS && 4 * j + wave_s < DSLOTS));
            const int qh = p0h + zh1[j], qw = p0w + zw1[j];
            if (isx) {
                const int ch = min(max(qh, 0), p.H - 1), cw = min(max(qw, 0), p.W - 1);
                fdn_lds_dma16_untracked(xrs, dst + (unsigned)((4 * j + wave_s) * 1024), (unsigned)((ch * p.W + cw) * 128) + cb[j], xplane);
            } else if (isz) {
                const unsigned vo = qh < p.H && qw < p.W ? (unsigned)((qh * p.W + qw) * 128) + cb[j] : 0xffffffffu;   // past the range: zeros
                fdn_lds_dma16_untracked(zrs, dst + (unsigned)((4 * j + wave_s) * 1024), vo, zplane);
            }
        }
        if (++td == dend) {          // next unit
            sg += ss; if (sg >= p.nseg) { sg -= p.nseg; ++tw; }
            tw += sw; if (tw >= p.ntw) { tw -= p.ntw; ++th; }
            th += sh; if (th >= p.nth) { th -= p.nth; ++tn; }
            tn += sn;
            td = sg * p.seg_len; dend = min(td + p.seg_len, p.D);
        }
    };

    if (nt > 0) request(0);
    if (nt > 1) request(1);
    int buf = 0;
#pragma unroll 1
    for (int i = 0; i < nt; ++i) {
        // tile i has landed (this wave's pieces; the barrier collects the other waves'), tile i + 1 may still be on its way
        if (i + 1 < nt) {
            if (wave_s == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!(FDN_DBG_BITS(p) & 4)) fdn_barrier_lds();
        const int b2 = buf == 0 ? 2 : buf - 1;          // (i + 2) % 3: the buffer tile i - 1 was read from
        if (i + 2 < nt) request(b2);
        const char* xs = smem + buf * DBUFB;

        // ---- 4 k-steps of 16 voxels (h-pairs), each 3 groups (b) of 3 MFMAs; halo-row ring as in the register kernel ----
        u32x2 gq[2][3], zq[2][2];
        auto reuse = [](int q) { return q % 3 == 0 && q != 0; };            // b == 0 and not the first h-pair
        auto slot = [&](int q) { int s_ = 0; for (int k = 1; k <= q; ++k) if (!reuse(k)) s_ ^= 1; return s_; };
        auto issue = [&](int q, u32x2 (&gg)[3], u32x2 (&z)[2]) {            // q = hp * 3 + b
            if (FDN_DBG_BITS(p) & 2) return;
            const int hp = q / 3, b = q % 3;
            const int dX = hp * 2 * XW * 128;
            if (!reuse(q)) {
                gg[0] = tr_read(xs + xoff[b][0] + dX); gg[1] = tr_read(xs + xoff[b][1] + dX); gg[2] = tr_read(xs + xoff[b][2] + dX);
            }
            if (b == 0) {
                const int dZ = hp * 2 * TW * 128;
                z[0] = tr_read(xs + zoff[0] + dZ); z[1] = tr_read(xs + zoff[1] + dZ);
            }
        };
        if (FDN_DBG_BITS(p) & 2) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int v = 0; v < 3; ++v) gq[u][v] = (u32x2){0x3f803f80u, 0x3f803f80u};
                zq[u][0] = zq[u][1] = (u32x2){0x3f803f80u, 0x3f803f80u};
            }
        }
        issue(0, gq[0], zq[0]);
        bf16x8 bv;
#pragma unroll
        for (int q = 0; q < (TH / 2) * 3; ++q) {
            const int b = q % 3;
            if (q + 1 < (TH / 2) * 3) issue(q + 1, gq[slot(q + 1)], zq[((q + 1) / 3) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            const u32x2 g0 = gq[slot(q)][0], g1 = gq[slot(q)][1], g2 = gq[slot(q)][2];
            if (b == 0) {
                const u32x2 z0 = zq[(q / 3) & 1][0], z1 = zq[(q / 3) & 1][1];
                bv = __builtin_bit_cast(bf16x8, (u32x4){z0.x, z0.y, z1.x, z1.y});
            }
            const bf16x8 a0 = __builtin_bit_cast(bf16x8, (u32x4){g0.x, g0.y, g1.x, g1.y});
            const bf16x8 a1 = __builtin_bit_cast(
                bf16x8, (u32x4){__builtin_amdgcn_alignbit(g0.y, g0.x, 16), __builtin_amdgcn_alignbit(g1.x, g0.y, 16),
                                __builtin_amdgcn_alignbit(g1.y, g1.x, 16), __builtin_amdgcn_alignbit(g2.x, g1.y, 16)});
            const bf16x8 a2 = __builtin_bit_cast(bf16x8, (u32x4){g0.y, g1.x, g1.y, g2.x});
            acc[b * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bv, acc[b * 3 + 0], 0, 0, 0);
            acc[b * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bv, acc[b * 3 + 1], 0, 0, 0);
            acc[b * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bv, acc[b * 3 + 2], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        buf = buf == 2 ? 0 : buf + 1;
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

__global__ __launch_bounds__(256, 2) void wgrad64_bf16_dma_kernel(Wgrad64BfArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    wgrad64_bf16_dma_body(p, (int)blockIdx.x, smem);
}

// Several layers of the SAME grid in one launch (fdn_conv3d_wgrad_bf16_batch; the fp32 path's wgrad64_wino_batch_kernel idea): a layer
// gets S = 168 / layers splits (a multiple of 8), so the launch still fills the chip about once, but a workgroup walks layers-times
// more tiles of ITS layer between prologue and output stage and the partial sums (and the one batched reduction) shrink by that factor.
// At the cfg4 low-res grid (4 x 32^3) a layer alone on the chip leaves a walk 12 one-plane tiles.
constexpr int kWgBfBatchMax = 8;
struct Wgrad64BfBatch {
    Wgrad64BfArgs a;                     // x / dz / partial of layer 0; everything else common to the layers
    const uint16_t* x[kWgBfBatchMax];
    const uint16_t* dz[kWgBfBatchMax];
};
struct WgBfDwTable { float* dw[kWgBfBatchMax]; };

__global__ __launch_bounds__(256, 2) void wgrad64_bf16_dma_batch_kernel(Wgrad64BfBatch b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int per = 3 * b.a.S;
    const int layer = (int)blockIdx.x / per;
    Wgrad64BfArgs p = b.a;
    p.x = b.x[layer]; p.dz = b.dz[layer];
    p.partial = b.a.partial + (size_t)layer * b.a.S * 27 * 4096;
    wgrad64_bf16_dma_body(p, (int)blockIdx.x - layer * per, smem);
}

// dw[layer] = sum over the S partials of that layer (wgrad64_reduce_kernel with a table of outputs; blockIdx.y = layer)
__global__ __launch_bounds__(256) void wgrad64_reduce_batch_kernel(const float* __restrict__ partial_base, WgBfDwTable t, int S) {
    __shared__ f32x4 red[3][64];
    const float* partial = partial_base + (size_t)blockIdx.y * S * 27 * 4096;
    const int col = threadIdx.x & 63, qtr = threadIdx.x >> 6;
    const int e4 = blockIdx.x * 64 + col;
    const f32x4* p = (const f32x4*)partial + e4;
    const int s0q = (S * qtr) >> 2, s1q = (S * (qtr + 1)) >> 2;
    f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    int s = s0q;
    for (; s + 2 <= s1q; s += 2) {
        s0 += p[(size_t)(s + 0) * (27 * 1024)];
        s1 += p[(size_t)(s + 1) * (27 * 1024)];
    }
    for (; s < s1q; ++s) s0 += p[(size_t)s * (27 * 1024)];
    const f32x4 v = s0 + s1;
    if (qtr) red[qtr - 1][col] = v;
    __syncthreads();
    if (qtr == 0) ((f32x4*)t.dw[blockIdx.y])[e4] = (v + red[0][col]) + (red[1][col] + red[2][col]);
}

namespace {
int wgrad64bf_splits(int N, int D, int H, int W) {
    // (counted in two-plane tiles for both kernels: the one-plane kernel then has >= 4 tiles per walk)
    const long long ntiles = (long long)N * ((D + TD - 1) / TD) * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);
    long long S = 168;                 // 3*168 = 504 workgroups ~ 2 per CU; a multiple of 8 (one walk set per XCD)
    while (S > 8 && ntiles / 2 < S) S -= 8;
    return (int)S;
}
}  // namespace

size_t fdn_wgrad64_bf16_workspace_bytes(int N, int D, int H, int W) {
    return (size_t)wgrad64bf_splits(N, D, H, W) * 27 * 4096 * sizeof(float);
}

int fdn_wgrad64_bf16_launch(const uint16_t* x, const uint16_t* dz, float* dw, void* ws, size_t ws_bytes, int N, int D, int H,
                            int W, hipStream_t s) {
    Wgrad64BfArgs a;
    a.x = x; a.dz = dz; a.partial = (float*)ws;
    a.N = N; a.D = D; a.H = H; a.W = W;
    a.S = wgrad64bf_splits(N, D, H, W);
    a.dbg = fdn_wgrad64bf_dbg;
    if (ws_bytes < (size_t)a.S * 27 * 4096 * sizeof(float)) {
        fdn_set_error("wgrad64_bf16: workspace too small");
        return FDN_ERR_WORKSPACE;
    }
    a.nth = (H + TH - 1) / TH; a.ntw = (W + TW - 1) / TW;
    // the LDS-DMA kernel addresses x / dz through 32-bit buffer offsets; tensors of 4 GB and more keep the register-staged kernel
    const long long bytes = (long long)N * D * H * W * 128;
    if (bytes < (1ll << 32) - 4096 && !fdn_wgrad64bf_variant) {
        a.bytes = (unsigned)bytes;
        a.ntd = D;
        a.ntiles = N * a.ntd * a.nth * a.ntw;
        // depth segments: long enough for the plane reuse between the taps (<= 32 planes), short enough that every walk of
        // every XCD gets >= ~4 units
        int want = (int)((long long)a.ntiles / ((long long)a.S * 4));
        want = want < 1 ? 1 : (want > 32 ? 32 : want);
        a.nseg = (D + want - 1) / want;
        a.seg_len = (D + a.nseg - 1) / a.nseg;
        a.nseg = (D + a.seg_len - 1) / a.seg_len;
        if (int rc = fdn_func_max_lds((const void*)wgrad64_bf16_dma_kernel, DLDS_BYTES, "wgrad64_bf16")) return rc;
        hipLaunchKernelGGL(wgrad64_bf16_dma_kernel, dim3(3 * a.S), dim3(256), DLDS_BYTES, s, a);
        FDN_CHECK_LAUNCH("wgrad64_bf16_dma_kernel");
    } else {
        a.bytes = 0;
        a.ntd = (D + TD - 1) / TD;
        a.ntiles = N * a.ntd * a.nth * a.ntw;
        hipLaunchKernelGGL(wgrad64_bf16_kernel, dim3(3 * a.S), dim3(256), LDS_BYTES, s, a);
        FDN_CHECK_LAUNCH("wgrad64_bf16_kernel");
    }
    hipLaunchKernelGGL(wgrad64_reduce_kernel, dim3(27 * 1024 / 64), dim3(256), 0, s, (const float*)ws, dw, a.S);
    FDN_CHECK_LAUNCH("wgrad64_reduce_kernel");
    return FDN_OK;
}

// ---- several layers of one grid: chunks of <= 7 layers, each ONE launch + one reduction (3 S layers ~ 504 workgroups) ----
static int wgrad64bf_batch_splits(int chunk) { int S = (168 / chunk) & ~7; return S < 8 ? 8 : S; }
bool fdn_wgrad64_bf16_batch_ok(int n_layers, int N, int D, int H, int W) {
    const long long tiles1 = (long long)N * D * ((H + TH - 1) / TH) * ((W + TW - 1) / TW);       // one-plane tiles of a layer
    return n_layers >= 2 && (long long)N * D * H * W * 128 < (1ll << 32) - 4096 && !fdn_wgrad64bf_variant && tiles1 >= 4 * 80;     // every walk of every chunk size gets >= 4 tiles
}
size_t fdn_wgrad64_bf16_batch_workspace_bytes(int n_layers, int N, int D, int H, int W) {
    const size_t one = fdn_wgrad64_bf16_workspace_bytes(N, D, H, W);
    size_t all = 0;
    for (int chunk = 2; chunk <= 7 && chunk <= n_layers; ++chunk) {
        const size_t b = (size_t)chunk * wgrad64bf_batch_splits(chunk) * 27 * 4096 * sizeof(float);
        if (b > all) all = b;
    }
    return all > one ? all : one;
}
int fdn_wgrad64_bf16_batch_launch(const uint16_t* const* x, const uint16_t* const* dz, float* const* dw, int n_layers, void* ws, size_t ws_bytes,
                                  int N, int D, int H, int W, hipStream_t s) {
    FDN_REQUIRE(fdn_wgrad64_bf16_batch_ok(n_layers, N, D, H, W), "wgrad64_bf16 (batched): %d layers of %dx%dx%dx%d are not batchable", n_layers, N, D, H, W);
    FDN_REQUIRE(ws_bytes >= fdn_wgrad64_bf16_batch_workspace_bytes(n_layers, N, D, H, W), "wgrad64_bf16 (batched): workspace too small");
    if (int rc = fdn_func_max_lds((const void*)wgrad64_bf16_dma_batch_kernel, DLDS_BYTES, "wgrad64_bf16_batch")) return rc;
    for (int first = 0; first < n_layers;) {
        int chunk = n_layers - first;
        if (chunk > 7) chunk = (chunk == 8) ? 4 : 7;           // (8 = 4 + 4 rather than 7 + 1)
        if (chunk == 1) {                                       // a lone layer: the single-layer launch
            if (int rc = fdn_wgrad64_bf16_launch(x[first], dz[first], dw[first], ws, ws_bytes, N, D, H, W, s)) return rc;
            first += 1;
            continue;
        }
        Wgrad64BfBatch b;
        WgBfDwTable t;
        for (int i = 0; i < chunk; ++i) {
            FDN_REQUIRE(x[first + i] && dz[first + i] && dw[first + i], "wgrad64_bf16 (batched): NULL pointer for layer %d", first + i);
            b.x[i] = x[first + i]; b.dz[i] = dz[first + i]; t.dw[i] = dw[first + i];
        }
        for (int i = chunk; i < kWgBfBatchMax; ++i) { b.x[i] = nullptr; b.dz[i] = nullptr; t.dw[i] = nullptr; }
        Wgrad64BfArgs& a = b.a;
        a.x = b.x[0]; a.dz = b.dz[0]; a.partial = (float*)ws;
        a.N = N; a.D = D; a.H = H; a.W = W;
        a.S = wgrad64bf_batch_splits(chunk);
        a.dbg = fdn_wgrad64bf_dbg;
        a.nth = (H + TH - 1) / TH; a.ntw = (W + TW - 1) / TW;
        a.bytes = (unsigned)((long long)N * D * H * W * 128);
        a.ntd = D;
        a.ntiles = N * a.ntd * a.nth * a.ntw;
        int want = (int)((long long)a.ntiles / ((long long)a.S * 4));
        want = want < 1 ? 1 : (want > 32 ? 32 : want);
        a.nseg = (D + want - 1) / want;
        a.seg_len = (D + a.nseg - 1) / a.nseg;
        a.nseg = (D + a.seg_len - 1) / a.seg_len;
        hipLaunchKernelGGL(wgrad64_bf16_dma_batch_kernel, dim3(3 * a.S * chunk), dim3(256), DLDS_BYTES, s, b);
        FDN_CHECK_LAUNCH("wgrad64_bf16_dma_batch_kernel");
        hipLaunchKernelGGL(wgrad64_reduce_batch_kernel, dim3(27 * 1024 / 64, chunk), dim3(256), 0, s, (const float*)ws, t, a.S);
        FDN_CHECK_LAUNCH("wgrad64_reduce_batch_kernel");
        first += chunk;
    }
    return FDN_OK;
}

#ifdef FDN_TEST_HOOKS
extern "C" int fdn_debug_set_wgrad64_bf16_dbg(int bits) { fdn_wgrad64bf_dbg = bits; return FDN_OK; }
extern "C" int fdn_debug_set_wgrad64_bf16_variant(int v) { fdn_wgrad64bf_variant = v; return FDN_OK; }
#endif

// The three 64 -> 1 head convolutions (SR4DFlowNet.py:40,43,46) -- forward, input gradient (with the halo fold and the
// producer's activation gradient) and weight gradient -- as small fp32 MFMA GEMMs plus a scalar stencil, instead of 1728
// VALU FMAs per voxel.  With one output channel the 3x3x3 conv factors into
//     forward : z[v][t] = sum_c x[v][c] w[t][c]            (V x 64) x (64 x 27) GEMM, per voxel, no neighbours
//               y[o]    = b + sum_t z[clamp(o + t - 1)][t]  27-point gather of scalars
//     backward: A[i][t] = sum_{o : clamp(o + t - 1) = i} dz[o]   (scalar stencil: 1 term inside the volume, up to 8 at a
//                                                                 corner -- this IS MirrorPadGrad, applied to scalars)
//               dx[i][c] = act'(y_prev[i][c]) sum_t A[i][t] w[t][c]      (V x 27) x (27 x 64) GEMM
//               dW[t][c] = sum_i A[i][t] x[i][c]                         (27 x V) x (V x 64) GEMM
// so every 64-channel row is touched once (forward: once per tile incl. halo) and the kernels run at the HBM roof.
// fp32 storage: all three use v_mfma_f32_32x32x2_f32 (K is 64, 27 or the voxel count): fp32 products and accumulation exactly like the
// VALU kernels they replace.  bf16 storage (T = bf16 bits): v_mfma_f32_32x32x16_bf16 on operands split into bf16 pieces -- forward and
// input gradient on hi + lo pairs (exact to 2^-17; their results are rounded to bf16 anyway), the weight gradient with EXACT products
// (x is bf16, the fp32 scalars split into three pieces).  The weight gradient took two attempts: with lane = voxel (A transposed through
// wave-private LDS planes, as in the fp32 path) the bf16 MFMAs made it SLOWER, 479 -> 572 us at (4,128^3) -- the write -> wait -> read
// chain of the LDS patches, not matrix time, bounds that layout; with lane = TAP (every lane forms its tap's scalars for 16 voxels
// straight from the dz halo), x requested a tile ahead and three workgroups per CU it runs 490 -> 355 us.
#include "fdn_common.h"
#include <type_traits>
#include <stdlib.h>

namespace {

__device__ __forceinline__ int clampi(int v, int hi) { return min(max(v, 0), hi); }

constexpr int H_TD = 4, H_TH = 8, H_TW = 8;                 // forward tile: 256 outputs, one per thread
constexpr int H_XD = H_TD + 2, H_XH = H_TH + 2, H_XW = H_TW + 2;
constexpr int H_HV = H_XD * H_XH * H_XW;                    // 600 halo voxels
constexpr int H_NMT = (H_HV + 31) / 32;                     // 19 M-tiles of 32 halo voxels (+1 dummy slot: 5 per wave)
constexpr int H_HVP = 20 * 32 + 1;                          // z plane stride: all 20 slots (no row predicate), odd (conflict-free)

// ---------------------------------------------------------------------------------------------------------------
// forward.  Phase 1: z[hv][t] for the 600 halo voxels of the tile (19 M-tiles of 32 voxels over 4 waves; A fragments are
// 16-B chunks straight from global, K order permuted like conv64_mfma.hip: lane (i,kh) holds cin 8g+4kh+{0..3}).
// Phase 2: one output per thread = 27 LDS reads.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256, 2) void head_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ y, int N, int D,
                                                          int H, int W, int ntd, int nth, int ntw, int iters, int ldy, int y_coff,
                                                          int act, float alpha, int xcd_walk) {
    extern __shared__ __attribute__((aligned(16))) float zbuf[];      // [27][H_HVP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int tiles_per_n = ntd * nth * ntw, ntiles = N * tiles_per_n;

    // K order: a lane holds E = 16 B / sizeof(T) consecutive channels per chunk (4 fp32 / 8 bf16): lane (i,kh), chunk g,
    // element s  <->  cin = 2E g + E kh + s.   B fragments: w[t = li][that cin], zero for the 5 padding columns.
    constexpr int E = 16 / (int)sizeof(T), NCH = 32 / E;
    // bf16 storage: the chunk IS the v_mfma_f32_32x32x16_bf16 operand (k = 8 kh + s of 16-channel block g), and the fp32
    // weights enter as an exact-to-2^-17 bf16 pair w = hi + lo: two bf16 MFMAs per block (8 per M-tile, 32 cycles each)
    // instead of 32 fp32 MFMAs of 64 cycles -- the z-GEMM drops from the matrix-bound to the HBM-bound side.
    // fp32 storage (round 6): the same bf16 pipe with EXACT splits -- x (as loaded) and w each in three bf16 pieces, six cross terms per
    // 16-channel block, fp32 accumulation (fdn_common.h): 24 MFMAs of 32 cycles per M-tile instead of 32 fp32 MFMAs of 64 -- the
    // z-GEMM over the tile's 600 halo voxels was 55 us of matrix time per launch at (8,48^3).  A lane's chunk c then holds cin
    // 16 (c >> 1) + 8 kh + 4 (c & 1) + s, so that chunks 2G, 2G + 1 are the 8 consecutive channels of block G's operand.
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    bf16x8 wh[E == 8 ? NCH : 1], wl[E == 8 ? NCH : 1];
    bf16x8 w3[E == 4 ? 4 : 1][3];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const float wv = li < 27 ? w[li * 64 + 16 * g + 8 * kh + s] : 0.f;     // (both storage types: block g, k = 8 kh + s)
            const __bf16 h = (__bf16)wv;
            if constexpr (E == 8) {
                wh[g][s] = h;
                wl[g][s] = (__bf16)(wv - (float)h);
            } else {
                const float r1 = wv - (float)h;
                const __bf16 m = (__bf16)r1;
                w3[g][0][s] = h; w3[g][1][s] = m; w3[g][2][s] = (__bf16)(r1 - (float)m);
            }
        }
    const float b0 = bias ? bias[0] : 0.f;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    // Each wave owns M-tiles wave, wave+4, ... of a tile: 5 steps per tile (the 20th slot is a dummy on wave 3).  A step's
    // NCH 16-B chunk loads are issued TWO steps ahead into a ring of three register buffers (kept packed; converted at
    // use); 5 steps per tile x 3 buffers -> the schedule repeats every 3 tiles, which are unrolled.  All control flow is
    // uniform and branch-free around the loads (tile indices past the end are clamped, their stores predicated), so every
    // wait is a counted vmcnt.  16-B lane loads matter: with 8-B loads the L1 sees every 64-B sector 8 times.
    // lane-constant part of the addressing: halo coordinates of this lane's row in each of its 5 M-tile slots
    int zpk[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        int hv = (wave + 4 * k) * 32 + li;
        hv = hv < H_HV ? hv : H_HV - 1;
        const int zd = hv / (H_XH * H_XW);
        const int r2 = hv - zd * (H_XH * H_XW);
        const int zh = r2 / H_XW;
        zpk[k] = zd | (zh << 8) | ((r2 - zh * H_XW) << 16);
    }
    struct TileOrg { int n, d, h, w; };
    auto decode = [&](int tile) {
        int b = min(tile, ntiles - 1);
        TileOrg o;
        o.n = b / tiles_per_n;
        b -= o.n * tiles_per_n;
        const int tdi = b / (nth * ntw);
        b -= tdi * (nth * ntw);
        const int thi = b / ntw;
        o.d = tdi * H_TD; o.h = thi * H_TH; o.w = (b - thi * ntw) * H_TW;
        return o;
    };
    auto load_tile = [&](const TileOrg& o, int k, u32x4 (&av)[NCH]) {
        const int zd = zpk[k] & 255, zh = (zpk[k] >> 8) & 255, zw = zpk[k] >> 16;
        const int qd = clampi(o.d + zd - 1, D - 1), qh = clampi(o.h + zh - 1, H - 1), qw = clampi(o.w + zw - 1, W - 1);
        const T* xp = x + ((size_t)o.n * D * H * W + ((size_t)qd * H + qh) * W + qw) * 64 + kh * 8;
#pragma unroll
        for (int g = 0; g < NCH; ++g) av[g] = *(const u32x4*)(xp + (E == 8 ? g * 16 : (g >> 1) * 16 + (g & 1) * 4));
    };
    const int G = gridDim.x;
    // XCD-aware walk: workgroup ids are dealt round-robin to the 8 XCDs; within a generation of G tiles every XCD takes one
    // contiguous run, so tiles that share halo rows (2.3x over-read per tile) meet in the same L2
    const int my = xcd_walk ? ((int)blockIdx.x & 7) * (G >> 3) + min((int)blockIdx.x & 7, G & 7) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    // 32 MFMAs of one M-tile slot into acc
    auto mfma_step = [&](const u32x4 (&src)[NCH], f32x16& acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int g = 0; g < NCH; ++g) {
            if constexpr (E == 4) {
                if (g & 1) continue;                                   // chunks g, g + 1 = block g / 2
                fdn_u32x2 h0, m0, l0, h1, m1, l1;
                fdn_split3(__builtin_bit_cast(f32x4, src[g]), h0, m0, l0);
                fdn_split3(__builtin_bit_cast(f32x4, src[g + 1]), h1, m1, l1);
                const bf16x8 ah = __builtin_bit_cast(bf16x8, (u32x4){h0.x, h0.y, h1.x, h1.y});
                const bf16x8 am = __builtin_bit_cast(bf16x8, (u32x4){m0.x, m0.y, m1.x, m1.y});
                const bf16x8 al = __builtin_bit_cast(bf16x8, (u32x4){l0.x, l0.y, l1.x, l1.y});
                const int G = g >> 1;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, w3[G][0], acc, 0, 0, 0);       // small terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, w3[G][2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, w3[G][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, w3[G][0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, w3[G][1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, w3[G][0], acc, 0, 0, 0);
            } else {
                const bf16x8 a = __builtin_bit_cast(bf16x8, src[g]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wh[g], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wl[g], acc, 0, 0, 0);
            }
        }
    };
    // C[voxel row][tap li] -> zbuf[tap][row]; rows of the dummy slot land in the padding
    float* zw0 = zbuf + li * H_HVP + wave * 32 + 4 * kh;
    auto z_store = [&](const f32x16& acc, int k) {
        if (li < 27) {
#pragma unroll
            for (int r = 0; r < 16; ++r) zw0[k * 128 + (r & 3) + 8 * (r >> 2)] = acc[r];
        }
    };
    // phase 2: one output per thread
    auto stencil = [&](const TileOrg& o, bool valid) {
        __syncthreads();
        const int od = tid >> 6, oh = (tid >> 3) & 7, ow = tid & 7;
        const int pd = o.d + od, ph = o.h + oh, pw = o.w + ow;
        float s = b0;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int bb = 0; bb < 3; ++bb)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    s += zbuf[((a * 3 + bb) * 3 + c) * H_HVP + ((od + a) * H_XH + oh + bb) * H_XW + ow + c];
        if (valid && pd < D && ph < H && pw < W)
            y[((size_t)o.n * D * H * W + ((size_t)pd * H + ph) * W + pw) * ldy + y_coff] = fdn_act(s, act, alpha);
        __syncthreads();               // zbuf is rewritten by the next tile
    };
    u32x4 buf[3][NCH];
    TileOrg cur = decode(my), nxt = decode(my + G);
    load_tile(cur, 0, buf[0]);
    load_tile(cur, 1, buf[1]);
    auto do_tile = [&](auto basec, int tile) {
        constexpr int BASE = decltype(basec)::value;        // ring position of this tile's step 0
        f32x16 acc[2];
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            load_tile(k + 2 < 5 ? cur : nxt, (k + 2) % 5, buf[(BASE + k + 2) % 3]);
            __builtin_amdgcn_sched_barrier(0);      // hipcc otherwise sinks the loads below the 32 MFMAs (distance 2 -> 1)
            mfma_step(buf[(BASE + k) % 3], acc[k & 1]);
            if (k > 0) z_store(acc[(k - 1) & 1], k - 1);     // the previous slot's LDS writes ride in this slot's MFMA shadow
        }
        z_store(acc[0], 4);
        stencil(cur, tile < ntiles);
        cur = nxt;
        nxt = decode(tile + 2 * G);
    };
#pragma unroll 1
    for (int it = 0; it < iters; it += 3) {
        const int t0 = my + it * G;
        do_tile(std::integral_constant<int, 0>{}, t0);           // steps 0-4   -> ring 0,1,2,0,1
        do_tile(std::integral_constant<int, 2>{}, t0 + G);       // steps 5-9   -> ring 2,0,1,2,0
        do_tile(std::integral_constant<int, 1>{}, t0 + 2 * G);   // steps 10-14 -> ring 1,2,0,1,2
    }
}

}  // namespace

template <typename T>
int fdn_head_fwd_launch(const T* x, const float* w, const float* bias, float* y, int N, int D, int H, int W, int ldy, int y_coff,
                        int act, float alpha, hipStream_t s, int xcd_walk) {
    const int ntd = (D + H_TD - 1) / H_TD, nth = (H + H_TH - 1) / H_TH, ntw = (W + H_TW - 1) / H_TW;
    const size_t lds = (size_t)27 * H_HVP * sizeof(float);
    if (int rc = fdn_func_max_lds((const void*)head_fwd_kernel<T>, (int)lds, "head_fwd")) return rc;
    // persistent grid: every workgroup runs `iters` (a multiple of 3, see the kernel) tiles, at most 2 workgroups per CU
    const int ntiles = N * ntd * nth * ntw;
    const int iters = 3 * ((ntiles + 3 * 512 - 1) / (3 * 512));
    const int grid = (ntiles + iters - 1) / iters;
    hipLaunchKernelGGL(head_fwd_kernel<T>, dim3((unsigned)grid), dim3(256), lds, s, x, w, bias, y, N, D, H, W, ntd, nth, ntw, iters,
                       ldy, y_coff, act, alpha, xcd_walk);
    FDN_CHECK_LAUNCH("head_fwd_kernel");
    return FDN_OK;
}
template int fdn_head_fwd_launch<float>(const float*, const float*, const float*, float*, int, int, int, int, int, int, int, float, hipStream_t, int);
template int fdn_head_fwd_launch<uint16_t>(const uint16_t*, const float*, const float*, float*, int, int, int, int, int, int, int, float, hipStream_t, int);

// ---------------------------------------------------------------------------------------------------------------
// input gradient of a head, with MirrorPadGrad and the producer's activation gradient:
//   dx[i][c] = act'(y_prev[i][c]) * sum_t A[i][t] w[t][c],   A[i][t] = sum_{o : clamp(o + t - 1) = i} dz[o].
// A is separable: per axis, with n[-1], n[0], n[+1] the dz neighbours of voxel i (zero outside the volume),
//   interior : tap 0 -> n[+1]         tap 1 -> n[0]   tap 2 -> n[-1]
//   i == 0   : tap 0 -> n[+1] + n[0]  tap 1 -> n[0]   tap 2 -> 0
//   i == D-1 : tap 0 -> 0             tap 1 -> n[0]   tap 2 -> n[-1] + n[0]          (D == 1: all three -> n[0])
// Tile = 4 x 8 x 8 voxels; the dz halo (600 scalars) is staged in LDS; a wave owns two M-tiles of 32 voxels (one d plane,
// 4 h rows); lane = voxel computes the 27 folded taps from its 3x3x3 neighbourhood and feeds them as the B operand of
// C[channel][voxel] = W^T[channel][tap] * A^T[tap][voxel]  (14 K-steps x 2 channel tiles of v_mfma_f32_32x32x2_f32).
// The weight rows are permuted (sigma, as in conv64_bf16.hip) so a lane ends up with 16 consecutive channels of its voxel:
// the mask loads and the stores are 16-B vectors.  Optional: per-channel sums of the result (BiasAddGrad of the producer).
// ---------------------------------------------------------------------------------------------------------------
namespace {

template <typename T>
__global__ __launch_bounds__(256, 2) void head_dgrad_kernel(const float* __restrict__ dz, const float* __restrict__ w,
                                                            const T* __restrict__ yprev, int act, float alpha,
                                                            T* __restrict__ out, float* __restrict__ bpart, int N, int D, int H,
                                                            int W, int ntd, int nth, int ntw, int lddz, int dz_coff,
                                                            const uint16_t* __restrict__ ymask) {
    __shared__ float zs[2][H_HV + 8];           // dz halo of the current / next tile
    __shared__ float bred[4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int tiles_per_n = ntd * nth * ntw, ntiles = N * tiles_per_n;
    const float slope = act == FDN_ACT_RELU ? 0.f : (act == FDN_ACT_LEAKY ? alpha : 1.f);

    // A operand: W^T rows permuted: row q of channel tile m <-> channel 32 m + 16 ((q>>2)&1) + (q&3) + 4 (q>>3);
    // lane (q = li, kh) holds w[t = 2s + kh][that channel] for the 14 K-steps
    // fp32 storage (VROW): the product is formed as D[voxel][channel] instead -- lane (li, kh) then holds ONE channel (32 m + li) of 16
    // voxels, and every mask load / store instruction covers whole 128-B lines (32 lanes x 4 B per voxel) instead of 64 scattered
    // 16-B pieces.
    constexpr bool VROW = sizeof(T) == 4;
    // bf16 storage: the result is rounded to bf16 on the way out, so the 27-tap contraction runs on v_mfma_f32_32x32x16_bf16 with both
    // operands split into bf16 pairs v = hi + lo (exact to 2^-17): hi*hi + hi*lo + lo*hi, 2 K-blocks of 16 taps x 2 channel tiles x 3 = 12
    // MFMAs of 32 cycles instead of 28 fp32 MFMAs of 64 (0.19 ms of matrix time per launch at (4,128^3)).  Lane (row li, kb = kh) holds
    // taps 16 blk + 8 kb + j, j = 0..7, of its row's channel.
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    float wa[VROW ? 2 : 1][VROW ? 14 : 1];
    bf16x8 wbh[VROW ? 1 : 2][VROW ? 1 : 2], wbl[VROW ? 1 : 2][VROW ? 1 : 2];       // [channel tile][K-block]
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int ch = VROW ? 32 * m + li : 32 * m + 16 * ((li >> 2) & 1) + (li & 3) + 4 * (li >> 3);
        if constexpr (VROW) {
#pragma unroll
            for (int s = 0; s < 14; ++s) wa[m][s] = (2 * s + kh) < 27 ? w[(2 * s + kh) * 64 + ch] : 0.f;
        } else {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int t = 16 * blk + 8 * kh + j;
                    const float wv = t < 27 ? w[t * 64 + ch] : 0.f;
                    const __bf16 h = (__bf16)wv;
                    wbh[m][blk][j] = h;
                    wbl[m][blk][j] = (__bf16)(wv - (float)h);
                }
        }
    }
    struct TileOrg { int n, d, h, w; };
    auto decode = [&](int tile) {
        int b = min(tile, ntiles - 1);
        TileOrg o;
        o.n = b / tiles_per_n;
        b -= o.n * tiles_per_n;
        const int tdi = b / (nth * ntw);
        b -= tdi * (nth * ntw);
        const int thi = b / ntw;
        o.d = tdi * H_TD; o.h = thi * H_TH; o.w = (b - thi * ntw) * H_TW;
        return o;
    };
    auto stage = [&](const TileOrg& o, float* dst) {            // zero outside the volume
        for (int i = tid; i < H_HV; i += 256) {
            const int zd = i / (H_XH * H_XW);
            const int r2 = i - zd * (H_XH * H_XW);
            const int zh = r2 / H_XW;
            const int qd = o.d + zd - 1, qh = o.h + zh - 1, qw = o.w + (r2 - zh * H_XW) - 1;
            float v = 0.f;
            if ((unsigned)qd < (unsigned)D && (unsigned)qh < (unsigned)H && (unsigned)qw < (unsigned)W)
                v = dz[((size_t)o.n * D * H * W + ((size_t)qd * H + qh) * W + qw) * lddz + dz_coff];
            dst[i] = v;
        }
    };
    float bsum[2][16];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) bsum[m][r] = 0.f;

    // the next tile's halo is requested before this tile's arithmetic and written to LDS after it (the loads' latency would otherwise
    // sit between the barrier and the first M-tile)
    constexpr int NST = (H_HV + 255) / 256;
    float stv[NST];
    auto stage_load = [&](const TileOrg& o) {
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int i = tid + 256 * u;
            const int zd = i / (H_XH * H_XW);
            const int r2 = i - zd * (H_XH * H_XW);
            const int zh = r2 / H_XW;
            const int qd = o.d + zd - 1, qh = o.h + zh - 1, qw = o.w + (r2 - zh * H_XW) - 1;
            float v = 0.f;
            if (i < H_HV && (unsigned)qd < (unsigned)D && (unsigned)qh < (unsigned)H && (unsigned)qw < (unsigned)W)
                v = dz[((size_t)o.n * D * H * W + ((size_t)qd * H + qh) * W + qw) * lddz + dz_coff];
            stv[u] = v;
        }
    };
    auto stage_store = [&](float* dst) {
#pragma unroll
        for (int u = 0; u < NST; ++u)
            if (tid + 256 * u < H_HV) dst[tid + 256 * u] = stv[u];
    };
    stage(decode(blockIdx.x), zs[0]);
    int par = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, par ^= 1) {
        __syncthreads();                                         // zs[par] complete; zs[par^1] no longer read
        const bool has_next = tile + (int)gridDim.x < ntiles;
        if (has_next) stage_load(decode(tile + gridDim.x));
        const TileOrg o = decode(tile);
        const float* z = zs[par];
#pragma unroll 1
        for (int mt = wave * 2; mt < wave * 2 + 2; ++mt) {       // M-tile: d plane mt >> 1, h rows 4 (mt & 1) .. +3
            const int vd = mt >> 1, vh = 4 * (mt & 1) + (li >> 3), vw = li & 7;
            const int gd = o.d + vd, gh = o.h + vh, gw = o.w + vw;
            // fp32 storage: the 32 mask values of this lane (one channel of 16 voxels, two channel tiles) are requested NOW, so that
            // their latency hides behind the fold arithmetic and the MFMAs (loaded one by one next to their stores they serialised
            // into 32 round trips per M-tile: 192 us per launch at (8,48^3) against 59 us without the mask)
            float ym[2][16];
            // fp32 storage with a sign mask (conv64_wino2d_kernel.h: planar [cout / 16][voxel] words, grids with W % 4 == 0): the lane's
            // channel 32 m + li is bit li & 15 of plane 2 m + (li >> 4); the four voxels of a register row are four consecutive words --
            // eight 8-B loads instead of the 32 rows above (7 MB instead of 226 MB per (8,48^3) launch)
            fdn_u32x2 ymw[VROW ? 2 : 1][VROW ? 4 : 1];
            if constexpr (VROW) {
                if (ymask) {
                    const int pw = min(o.w + 4 * kh, W - 4), ph0 = o.h + 4 * (mt & 1);
                    const size_t nv = (size_t)N * D * H * W;
                    const size_t plane = (size_t)o.n * D * H * W + (size_t)min(gd, D - 1) * H * W;
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr)
                            ymw[m][rr] = *(const fdn_u32x2*)(ymask + (size_t)(2 * m + (li >> 4)) * nv + plane + (size_t)min(ph0 + rr, H - 1) * W + pw);
                } else
                if (yprev) {
                    const int pw = o.w + 4 * kh, ph0 = o.h + 4 * (mt & 1);
                    const size_t plane = (size_t)o.n * D * H * W + (size_t)min(gd, D - 1) * H * W;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const size_t e = (plane + (size_t)min(ph0 + (r >> 2), H - 1) * W + min(pw + (r & 3), W - 1)) * 64 + li;
                        ym[0][r] = fdn_ld1(yprev + e);
                        ym[1][r] = fdn_ld1(yprev + e + 32);
                    }
                }
            }
            // bf16 storage: this lane's 32 mask values (16 consecutive channels of its voxel in each channel tile) = four 16-B vectors,
            // requested here for the same reason
            constexpr int EVP = FdnVec<T>::E;
            float yq[VROW ? 1 : 2][VROW ? 1 : 16];
            unsigned ymk[2] = {0u, 0u};              // bf16 storage with a sign mask (conv64_args.h: [voxel][cout / 16] words): this lane's 16 channels per channel tile = one word
            if constexpr (!VROW) {
                if (ymask) {
                    const size_t voxp = (size_t)o.n * D * H * W + ((size_t)min(gd, D - 1) * H + min(gh, H - 1)) * W + min(gw, W - 1);
                    ymk[0] = ymask[voxp * 4 + kh];
                    ymk[1] = ymask[voxp * 4 + 2 + kh];
                } else if (yprev) {
                    const size_t voxp = (size_t)o.n * D * H * W + ((size_t)min(gd, D - 1) * H + min(gh, H - 1)) * W + min(gw, W - 1);
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int r0 = 0; r0 < 16; r0 += EVP) {
                            float t[EVP];
                            FdnVec<T>::ld(yprev + voxp * 64 + 32 * m + 16 * kh + r0, t);
#pragma unroll
                            for (int q = 0; q < EVP; ++q) yq[m][r0 + q] = t[q];
                        }
                }
            }
            // 3x3x3 neighbourhood (halo coordinates vd..vd+2 etc.), then the per-axis fold transforms
            float nb[3][3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b)
#pragma unroll
                    for (int c = 0; c < 3; ++c) nb[a][b][c] = z[((vd + a) * H_XH + vh + b) * H_XW + vw + c];
            // axis rule: f[0] = lo ? n[+1] + n[0] : (hi ? 0 : n[+1]); f[1] = n[0]; f[2] = hi ? n[-1] + n[0] : (lo ? 0 : n[-1]);
            // with D == 1 (lo && hi): f[0] = f[2] = n[0].   n[-1] = index 0, n[0] = index 1, n[+1] = index 2.
#define FDN_FOLD_AXIS(LO, HI, NM, N0, NP, F0, F2)                                  \
            {                                                                      \
                const float nm_ = NM, n0_ = N0, np_ = NP;                          \
                F0 = (LO) ? ((HI) ? n0_ : np_ + n0_) : ((HI) ? 0.f : np_);         \
                F2 = (HI) ? ((LO) ? n0_ : nm_ + n0_) : ((LO) ? 0.f : nm_);         \
            }
            const bool wl = gw == 0, wh = gw == W - 1, hl = gh == 0, hh = gh == H - 1, dl = gd == 0, dh = gd == D - 1;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    float f0, f2;
                    FDN_FOLD_AXIS(wl, wh, nb[a][b][0], nb[a][b][1], nb[a][b][2], f0, f2);
                    nb[a][b][0] = f0; nb[a][b][2] = f2;              // now indexed by tap c
                }
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float f0, f2;
                    FDN_FOLD_AXIS(hl, hh, nb[a][0][c], nb[a][1][c], nb[a][2][c], f0, f2);
                    nb[a][0][c] = f0; nb[a][2][c] = f2;
                }
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    float f0, f2;
                    FDN_FOLD_AXIS(dl, dh, nb[0][b][c], nb[1][b][c], nb[2][b][c], f0, f2);
                    nb[0][b][c] = f0; nb[2][b][c] = f2;
                }
#undef FDN_FOLD_AXIS
            // B operand: lane (voxel, kh) supplies A[voxel][t = 2s + kh]
            f32x16 acc[2];
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
            if constexpr (VROW) {
#pragma unroll
                for (int s = 0; s < 14; ++s) {
                    const int t0 = 2 * s, t1 = 2 * s + 1;
                    const float e0 = nb[t0 / 9][(t0 / 3) % 3][t0 % 3];
                    const float e1 = t1 < 27 ? nb[t1 / 9][(t1 / 3) % 3][t1 % 3] : 0.f;
                    const float bv = kh ? e1 : e0;
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, wa[0][s], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(bv, wa[1][s], acc[1], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    bf16x8 ah, al;                       // this lane's voxel, taps 16 blk + 8 kh + j
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int t0 = 16 * blk + j, t1 = 16 * blk + 8 + j;
                        const float e0 = nb[t0 / 9][(t0 / 3) % 3][t0 % 3];
                        const float e1 = t1 < 27 ? nb[(t1 % 27) / 9][((t1 % 27) / 3) % 3][t1 % 3] : 0.f;
                        const float av = kh ? e1 : e0;
                        const __bf16 h = (__bf16)av;
                        ah[j] = h;
                        al[j] = (__bf16)(av - (float)h);
                    }
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wbh[m][blk], ah, acc[m], 0, 0, 0);
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wbh[m][blk], al, acc[m], 0, 0, 0);
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wbl[m][blk], ah, acc[m], 0, 0, 0);
                    }
                }
            }
            if constexpr (VROW) {
                // register r of lane (li, kh) = voxel row (r & 3) + 8 (r >> 2) + 4 kh of the M-tile: h row r >> 2, w (r & 3) + 4 kh
                const int pw = o.w + 4 * kh, ph0 = o.h + 4 * (mt & 1);
                const size_t plane = (size_t)o.n * D * H * W + (size_t)min(gd, D - 1) * H * W;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int qh = ph0 + (r >> 2), qw = pw + (r & 3);
                    const bool in2 = gd < D && qh < H && qw < W;
                    const size_t e = (plane + (size_t)min(qh, H - 1) * W + min(qw, W - 1)) * 64 + li;
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        float v = acc[m][r];
                        if (ymask) v *= ((ymw[m][r >> 2][(r & 3) >> 1] >> (16 * (r & 1) + (li & 15))) & 1u) ? 1.f : slope;
                        else if (yprev) v *= ym[m][r] > 0.f ? 1.f : slope;
                        if (in2) {
                            fdn_st1(out + e + 32 * m, v);
                            bsum[m][0] += v;
                        }
                    }
                }
                continue;
            }
            // lane (voxel li, half kh) holds channels 32 m + 16 kh + r
            const bool inside = gd < D && gh < H && gw < W;
            const size_t vox = (size_t)o.n * D * H * W + ((size_t)min(gd, D - 1) * H + min(gh, H - 1)) * W + min(gw, W - 1);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const size_t e = vox * 64 + 32 * m + 16 * kh;
                constexpr int EV = FdnVec<T>::E;                 // 16-B vectors: 4 fp32 / 8 bf16 channels per access
#pragma unroll
                for (int r0 = 0; r0 < 16; r0 += EV) {
                    float v[EV];
#pragma unroll
                    for (int q = 0; q < EV; ++q) v[q] = acc[m][r0 + q];
                    if (!VROW && ymask) {
#pragma unroll
                        for (int q = 0; q < EV; ++q) v[q] *= ((ymk[m] >> (r0 + q)) & 1u) ? 1.f : slope;
                    } else if (yprev) {
#pragma unroll
                        for (int q = 0; q < EV; ++q) v[q] *= yq[VROW ? 0 : m][VROW ? 0 : r0 + q] > 0.f ? 1.f : slope;
                    }
                    if (inside) {
                        FdnVec<T>::st(out + e + r0, v);
#pragma unroll
                        for (int q = 0; q < EV; ++q) bsum[m][r0 + q] += v[q];
                    }
                }
            }
        }
        if (has_next) stage_store(zs[par ^ 1]);
    }
    if (bpart && VROW) {
        // per-channel sums: a lane holds channel 32 m + li; fold the two lane halves, then the 4 waves through LDS
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            float v = bsum[m][0];
            v += __shfl_xor(v, 32, 64);
            if (kh == 0) bred[wave][32 * m + li] = v;
        }
        __syncthreads();
        if (tid < 64) bpart[(size_t)blockIdx.x * 64 + tid] = (bred[0][tid] + bred[1][tid]) + (bred[2][tid] + bred[3][tid]);
    } else if (bpart) {
        // per-channel sums: fold the 32 voxel lanes of each half-wave, then the 4 waves through LDS
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = bsum[m][r];
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
                v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64);
                if (li == 0) bred[wave][32 * m + 16 * kh + r] = v;
            }
        __syncthreads();
        if (tid < 64) bpart[(size_t)blockIdx.x * 64 + tid] = (bred[0][tid] + bred[1][tid]) + (bred[2][tid] + bred[3][tid]);
    }
}

}  // namespace

// grid size of the persistent dgrad kernel = number of bias partial rows the caller must provide
int fdn_head_dgrad_blocks(int N, int D, int H, int W) {
    const long long ntiles = (long long)N * ((D + H_TD - 1) / H_TD) * ((H + H_TH - 1) / H_TH) * ((W + H_TW - 1) / H_TW);
    return (int)(ntiles < 1024 ? ntiles : 1024);
}

template <typename T>
int fdn_head_dgrad_launch(const float* dz, const float* w, const T* y_prev, int act, float alpha, T* dz_prev, float* bpart, int N,
                          int D, int H, int W, int lddz, int dz_coff, hipStream_t s, const uint16_t* ymask) {
    const int ntd = (D + H_TD - 1) / H_TD, nth = (H + H_TH - 1) / H_TH, ntw = (W + H_TW - 1) / H_TW;
    hipLaunchKernelGGL(head_dgrad_kernel<T>, dim3((unsigned)fdn_head_dgrad_blocks(N, D, H, W)), dim3(256), 0, s, dz, w, y_prev, act,
                       alpha, dz_prev, bpart, N, D, H, W, ntd, nth, ntw, lddz, dz_coff, ymask);
    FDN_CHECK_LAUNCH("head_dgrad_kernel");
    return FDN_OK;
}
template int fdn_head_dgrad_launch<float>(const float*, const float*, const float*, int, float, float*, float*, int, int, int, int, int, int, hipStream_t, const uint16_t*);
template int fdn_head_dgrad_launch<uint16_t>(const float*, const float*, const uint16_t*, int, float, uint16_t*, float*, int, int, int, int, int, int, hipStream_t, const uint16_t*);

// ---------------------------------------------------------------------------------------------------------------
// weight gradient of a head:  dW[t][c] = sum_i A[i][t] x[i][c]  with the same folded scalar stencil A as the input gradient
// (MirrorPadGrad moved from the 64-channel rows onto the scalars).  C[tap][channel] accumulates in registers over all the
// voxels of a persistent workgroup: per chunk of 32 voxels a wave computes A (lane = voxel) and transposes it through a
// private LDS patch, stages the 32 x rows (coalesced 16-B loads, fp32 in LDS) and issues 16 K-steps x 2 channel tiles of
// v_mfma_f32_32x32x2_f32.  Partials [grid][27*64] are summed by reduce_partials_kernel.
// ---------------------------------------------------------------------------------------------------------------
namespace {

template <typename T>
__global__ __launch_bounds__(256, sizeof(T) == 2 ? 3 : 2) void head_wgrad_kernel(const T* __restrict__ x, const float* __restrict__ dz,
                                                            float* __restrict__ partial, int N, int D, int H, int W, int ntd,
                                                            int nth, int ntw, int lddz, int dz_coff) {
    __shared__ float zs[2][H_HV + 8];
    __shared__ __attribute__((aligned(16))) float xs[4][32 * 64];      // per wave: 32 voxels x 64 channels
    __shared__ float as[4][32 * 33];                                   // per wave: A^T [tap][voxel], stride 33
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, kh = lane >> 5;
    const int tiles_per_n = ntd * nth * ntw, ntiles = N * tiles_per_n;
    constexpr int E = 16 / (int)sizeof(T);          // elements per 16-B chunk
    constexpr int CPR = 64 / E;                     // chunks per voxel row (16 fp32 / 8 bf16)
    constexpr int NLD = 32 * CPR / 64;              // 16-B loads per lane and chunk of 32 voxels (8 / 4)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    struct TileOrg { int n, d, h, w; };
    auto decode = [&](int tile) {
        int b = min(tile, ntiles - 1);
        TileOrg o;
        o.n = b / tiles_per_n;
        b -= o.n * tiles_per_n;
        const int tdi = b / (nth * ntw);
        b -= tdi * (nth * ntw);
        const int thi = b / ntw;
        o.d = tdi * H_TD; o.h = thi * H_TH; o.w = (b - thi * ntw) * H_TW;
        return o;
    };
    auto stage = [&](const TileOrg& o, float* dst) {
        for (int i = tid; i < H_HV; i += 256) {
            const int zd = i / (H_XH * H_XW);
            const int r2 = i - zd * (H_XH * H_XW);
            const int zh = r2 / H_XW;
            const int qd = o.d + zd - 1, qh = o.h + zh - 1, qw = o.w + (r2 - zh * H_XW) - 1;
            float v = 0.f;
            if ((unsigned)qd < (unsigned)D && (unsigned)qh < (unsigned)H && (unsigned)qw < (unsigned)W)
                v = dz[((size_t)o.n * D * H * W + ((size_t)qd * H + qh) * W + qw) * lddz + dz_coff];
            dst[i] = v;
        }
    };
    // x rows of chunk mt of tile o: lane loads 16-B piece (lane % CPR) of rows lane / CPR + (64 / CPR) u
    auto load_x = [&](const TileOrg& o, int mt, u32x4 (&v)[NLD]) {
#pragma unroll
        for (int u = 0; u < NLD; ++u) {
            const int row = lane / CPR + (64 / CPR) * u;
            const int gd = min(o.d + (mt >> 1), D - 1), gh = min(o.h + 4 * (mt & 1) + (row >> 3), H - 1), gw = min(o.w + (row & 7), W - 1);
            v[u] = *(const u32x4*)(x + ((size_t)o.n * D * H * W + ((size_t)gd * H + gh) * W + gw) * 64 + (lane % CPR) * E);
        }
    };
    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    float* xw = xs[wave];
    float* aw = as[wave];

    constexpr int NST = (H_HV + 255) / 256;
    float stv[NST];
    auto stage_load = [&](const TileOrg& o) {
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int i = tid + 256 * u;
            const int zd = i / (H_XH * H_XW);
            const int r2 = i - zd * (H_XH * H_XW);
            const int zh = r2 / H_XW;
            const int qd = o.d + zd - 1, qh = o.h + zh - 1, qw = o.w + (r2 - zh * H_XW) - 1;
            float v = 0.f;
            if (i < H_HV && (unsigned)qd < (unsigned)D && (unsigned)qh < (unsigned)H && (unsigned)qw < (unsigned)W)
                v = dz[((size_t)o.n * D * H * W + ((size_t)qd * H + qh) * W + qw) * lddz + dz_coff];
            stv[u] = v;
        }
    };
    auto stage_store = [&](float* dst) {
#pragma unroll
        for (int u = 0; u < NST; ++u)
            if (tid + 256 * u < H_HV) dst[tid + 256 * u] = stv[u];
    };
    TileOrg o = decode(blockIdx.x);
    stage(o, zs[0]);
    u32x4 xv[NLD], xv2[E == 8 ? NLD : 1];          // (bf16 storage: both chunks of a tile are requested a tile ahead)
    if ((int)blockIdx.x < ntiles) {
        load_x(o, wave * 2, xv);
        if constexpr (E == 8) load_x(o, wave * 2 + 1, xv2);
    }
    int par = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x, par ^= 1) {
        __syncthreads();
        const bool has_next = tile + (int)gridDim.x < ntiles;
        const TileOrg on = decode(tile + gridDim.x);
        if (has_next) stage_load(on);                            // requested now, written to LDS after this tile's arithmetic
        const float* z = zs[par];
#pragma unroll 1
        for (int q = 0; q < 2; ++q) {
            const int mt = wave * 2 + q;
            if constexpr (E == 8) {
                // ---- bf16 storage (round 5): lane = TAP.  C[tap][channel] += A^T[tap][voxel] x[voxel][channel] on v_mfma_f32_32x32x16_bf16: the
                // row operand of lane (tap li, kb = kh) is that tap's folded scalar at voxels 16 kb2 + 8 kb + j, j = 0..7 (h row 2 kb2 + kb of the
                // chunk, w = j) -- which the lane forms ITSELF from the staged dz halo: inside the volume A[v][t] is dz at v + 1 - t, one LDS
                // read per value (16 per lane and chunk; the fp32 path reads 27 per voxel-lane and transposes A through LDS), on the volume's
                // faces the same per-axis rule as there (tap 0 takes n[+1] plus n[0] on the low face, tap 2 n[-1] plus n[0] on the high face).
                // The fp32 scalars split exactly into three bf16 pieces (8 + 8 + 8 mantissa bits) and x is bf16, so the products are exact and
                // the sum is what the fp32 MFMAs give.  x stays bf16 in LDS (128 B per voxel, halves swapped on rows with bit 1 set) and
                // comes back voxel-strided through ds_read_b64_tr_b16.  12 MFMAs of 32 cycles per chunk instead of 32 of 64, and no
                // write -> wait -> read round trip for A (the first attempt of this round kept lane = voxel and lost 19 % to that chain). ----
                typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
                typedef short v4i16 __attribute__((ext_vector_type(4)));
                typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                // x rows of this chunk -> LDS; the same chunk of the NEXT tile is requested now, a whole tile (two chunks) ahead: with one
                // chunk of lead (the fp32 path) the 8 waves of a CU keep 32 KB in flight, 3.3 TB/s at 2.5 us of latency whatever the arithmetic costs
                if (q == 0) {
#pragma unroll
                    for (int u = 0; u < NLD; ++u) {
                        const int row = lane / CPR + (64 / CPR) * u;
                        *(u32x4*)((char*)xw + row * 128 + (((lane % CPR) * 16) ^ (((row >> 1) & 1) << 6))) = xv[u];
                    }
                    if (has_next) load_x(on, wave * 2, xv);
                } else {
#pragma unroll
                    for (int u = 0; u < NLD; ++u) {
                        const int row = lane / CPR + (64 / CPR) * u;
                        *(u32x4*)((char*)xw + row * 128 + (((lane % CPR) * 16) ^ (((row >> 1) & 1) << 6))) = xv2[u];
                    }
                    if (has_next) load_x(on, wave * 2 + 1, xv2);
                }
                const bool tapv = li < 27;                                   // (li >= 27: padding rows of the MFMA, read like the centre tap, zeroed below)
                const int ta = tapv ? li / 9 : 1, tb = tapv ? (li / 3) % 3 : 1, tc = tapv ? li % 3 : 1;       // this lane's tap
                // chunk-level (scalar) facts: which axes have a volume face inside this chunk -- only those get the face term n[0] -- and
                // whether some of its voxels lie outside the volume (ragged last tiles)
                const int mtu = __builtin_amdgcn_readfirstlane(mt);
                const int vd = mtu >> 1, hq = 4 * (mtu & 1);
                const int gd = o.d + vd;
                const bool needD = gd == 0 || gd >= D - 1, needH = o.h + hq == 0 || o.h + hq + 4 > H - 1, needW = o.w == 0 || o.w + H_TW > W - 1;
                const bool ragged = gd >= D || o.h + hq + 4 > H || o.w + H_TW > W;
                const bool d2 = (ta == 0 && gd == 0) || (ta == 2 && gd == D - 1);
                float av[2][8];
#pragma unroll
                for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
                    for (int jj = 0; jj < 8; ++jj) av[kb2][jj] = 0.f;
                // 2 x 2 x 2 source terms per value: per axis the tap's own neighbour (tap 0: n[+1], 1: n[0], 2: n[-1]) and, on a face, n[0] again;
                // a combination that uses a face term is skipped by a scalar branch unless the chunk touches that face (81 % of the chunks
                // of a 128^3 grid run combination 0 alone: one LDS read per value)
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const bool ud = c & 4, uh = c & 2, uw = c & 1;
                    if ((ud && !needD) || (uh && !needH) || (uw && !needW)) continue;
                    const int zb = ((vd + 1 + (ud ? 0 : 1 - ta)) * H_XH + hq + kh + 1 + (uh ? 0 : 1 - tb)) * H_XW + 1 + (uw ? 0 : 1 - tc);
#pragma unroll
                    for (int kb2 = 0; kb2 < 2; ++kb2) {
                        const int gh = o.h + hq + 2 * kb2 + kh;
                        const bool h2 = (tb == 0 && gh == 0) || (tb == 2 && gh == H - 1);
                        const bool dh_on = (!ud || d2) && (!uh || h2);
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) {
                            const bool w2 = (tc == 0 && o.w + jj == 0) || (tc == 2 && o.w + jj == W - 1);
                            const float zv = z[zb + 2 * kb2 * H_XW + jj];
                            av[kb2][jj] += (c == 0 || (dh_on && (!uw || w2))) ? zv : 0.f;
                        }
                    }
                }
                if (ragged) {
#pragma unroll
                    for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj)
                            if (!(gd < D && o.h + hq + 2 * kb2 + kh < H && o.w + jj < W)) av[kb2][jj] = 0.f;
                }
                if (!tapv) {
#pragma unroll
                    for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
                        for (int jj = 0; jj < 8; ++jj) av[kb2][jj] = 0.f;
                }
                bf16x8 ap[2][3];
#pragma unroll
                for (int kb2 = 0; kb2 < 2; ++kb2)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float a0 = av[kb2][j];
                        const __bf16 p0 = (__bf16)a0;
                        const float r1 = a0 - (float)p0;
                        const __bf16 p1 = (__bf16)r1;
                        ap[kb2][0][j] = p0; ap[kb2][1][j] = p1; ap[kb2][2][j] = (__bf16)(r1 - (float)p1);
                    }
                __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the x rows are in LDS (wave-private patch)
                __builtin_amdgcn_wave_barrier();
                const int g = lane >> 4, pq = lane & 15;
#pragma unroll
                for (int kb2 = 0; kb2 < 2; ++kb2) {
                    const int r0 = 16 * kb2 + 8 * (g >> 1) + (pq >> 2);
                    const int xo = 32 * (g & 1) + 8 * (pq & 3);
                    bf16x8 xb[2];
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const char* a0 = (const char*)xw + r0 * 128 + ((64 * m) ^ (((r0 >> 1) & 1) << 6)) + xo;
                        const char* a1 = (const char*)xw + (r0 + 4) * 128 + ((64 * m) ^ ((((r0 + 4) >> 1) & 1) << 6)) + xo;
                        const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4i16 __attribute__((address_space(3)))*)a0));
                        const u32x2 hi = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4i16 __attribute__((address_space(3)))*)a1));
                        xb[m] = __builtin_bit_cast(bf16x8, (u32x4){lo.x, lo.y, hi.x, hi.y});
                    }
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[kb2][c], xb[0], acc[0], 0, 0, 0);
                        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[kb2][c], xb[1], acc[1], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_wave_barrier();         // all reads done before the next chunk overwrites the patch
                continue;
            }
            // ---- x rows -> LDS (fp32), next chunk's loads in flight ----
#pragma unroll
            for (int u = 0; u < NLD; ++u) {
                const int row = lane / CPR + (64 / CPR) * u;
                float* dst = xw + row * 64 + (lane % CPR) * E;
                if constexpr (E == 4) {
                    *(f32x4*)dst = __builtin_bit_cast(f32x4, xv[u]);
                } else {
                    f32x4 lo, hi;
                    lo.x = __builtin_bit_cast(float, xv[u].x << 16); lo.y = __builtin_bit_cast(float, xv[u].x & 0xffff0000u);
                    lo.z = __builtin_bit_cast(float, xv[u].y << 16); lo.w = __builtin_bit_cast(float, xv[u].y & 0xffff0000u);
                    hi.x = __builtin_bit_cast(float, xv[u].z << 16); hi.y = __builtin_bit_cast(float, xv[u].z & 0xffff0000u);
                    hi.z = __builtin_bit_cast(float, xv[u].w << 16); hi.w = __builtin_bit_cast(float, xv[u].w & 0xffff0000u);
                    *(f32x4*)dst = lo; *(f32x4*)(dst + 4) = hi;
                }
            }
            if (q == 0) load_x(o, mt + 1, xv);
            else if (has_next) load_x(on, wave * 2, xv);
            // ---- folded taps of this lane's voxel (see head_dgrad_kernel) ----
            const int vd = mt >> 1, vh = 4 * (mt & 1) + (li >> 3), vw = li & 7;
            const int gd = o.d + vd, gh = o.h + vh, gw = o.w + vw;
            float nb[3][3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b)
#pragma unroll
                    for (int c = 0; c < 3; ++c) nb[a][b][c] = z[((vd + a) * H_XH + vh + b) * H_XW + vw + c];
            const bool wl = gw == 0, wh = gw == W - 1, hl = gh == 0, hh = gh == H - 1, dl = gd == 0, dh = gd == D - 1;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const float nm = nb[a][b][0], n0 = nb[a][b][1], np = nb[a][b][2];
                    nb[a][b][0] = np + (wl ? n0 : 0.f);      // tap 0 <- n[+1] (+ n[0] on the low face)
                    nb[a][b][2] = nm + (wh ? n0 : 0.f);      // tap 2 <- n[-1] (+ n[0] on the high face)
                }
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float nm = nb[a][0][c], n0 = nb[a][1][c], np = nb[a][2][c];
                    nb[a][0][c] = np + (hl ? n0 : 0.f);
                    nb[a][2][c] = nm + (hh ? n0 : 0.f);
                }
#pragma unroll
            for (int b = 0; b < 3; ++b)
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float nm = nb[0][b][c], n0 = nb[1][b][c], np = nb[2][b][c];
                    nb[0][b][c] = np + (dl ? n0 : 0.f);
                    nb[2][b][c] = nm + (dh ? n0 : 0.f);
                }
            const bool inside = gd < D && gh < H && gw < W;
            // A^T[tap][voxel]: half 0 writes taps 0..13, half 1 taps 14..26 (and zeros rows 27..31 once per chunk)
#pragma unroll
            for (int t = 0; t < 14; ++t) {
                const int tt = t + 14 * kh;
                const int t0 = t, t1 = t + 14;
                const float e0 = nb[t0 / 9][(t0 / 3) % 3][t0 % 3];
                const float e1 = t1 < 27 ? nb[t1 / 9][(t1 / 3) % 3][t1 % 3] : 0.f;
                aw[tt * 33 + li] = inside ? (kh ? e1 : e0) : 0.f;
            }
            if (kh == 0) {
#pragma unroll
                for (int t = 28; t < 32; ++t) aw[t * 33 + li] = 0.f;
            }
            // wave-private LDS: make the writes visible to the other lanes of the wave
            __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0)
            __builtin_amdgcn_wave_barrier();
            // ---- 16 K-steps of 2 voxels ----
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const float av = aw[li * 33 + 2 * s + kh];
                const float b0 = xw[(2 * s + kh) * 64 + li], b1 = xw[(2 * s + kh) * 64 + 32 + li];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc[1], 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();         // all reads done before the next chunk overwrites the patches
        }
        if (has_next) stage_store(zs[par ^ 1]);
        o = on;
    }
    // C[tap][channel]: lane (channel li of tile m, half kh) holds taps (r&3) + 8 (r>>2) + 4 kh; fold the 4 waves through LDS
    __syncthreads();
    float* red = &xs[0][0];                 // 4 x 27*64 floats fit in the x patches (4 x 2048)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (t < 27) red[wave * 2048 + t * 64 + 32 * m + li] = acc[m][r];
        }
    __syncthreads();
    for (int i = tid; i < 27 * 64; i += 256)
        partial[(size_t)blockIdx.x * (27 * 64) + i] = (red[i] + red[2048 + i]) + (red[4096 + i] + red[6144 + i]);
}

}  // namespace

// persistent grid = partial rows the caller provides: two workgroups per CU for fp32 storage (54.5 KB of LDS), three for bf16 (37.6 KB, 153 VGPRs)
int fdn_head_wgrad_blocks(int N, int D, int H, int W, int elem_bytes) {
    const long long ntiles = (long long)N * ((D + H_TD - 1) / H_TD) * ((H + H_TH - 1) / H_TH) * ((W + H_TW - 1) / H_TW);
    const long long cap = elem_bytes == 2 ? 768 : 512;
    return (int)(ntiles < cap ? ntiles : cap);
}

template <typename T>
int fdn_head_wgrad_launch(const T* x, const float* dz, float* partial, int N, int D, int H, int W, int lddz, int dz_coff,
                          hipStream_t s) {
    const int ntd = (D + H_TD - 1) / H_TD, nth = (H + H_TH - 1) / H_TH, ntw = (W + H_TW - 1) / H_TW;
    hipLaunchKernelGGL(head_wgrad_kernel<T>, dim3((unsigned)fdn_head_wgrad_blocks(N, D, H, W, (int)sizeof(T))), dim3(256), 0, s, x, dz, partial, N, D,
                       H, W, ntd, nth, ntw, lddz, dz_coff);
    FDN_CHECK_LAUNCH("head_wgrad_kernel");
    return FDN_OK;
}
template int fdn_head_wgrad_launch<float>(const float*, const float*, float*, int, int, int, int, int, int, hipStream_t);
template int fdn_head_wgrad_launch<uint16_t>(const uint16_t*, const float*, float*, int, int, int, int, int, int, hipStream_t);

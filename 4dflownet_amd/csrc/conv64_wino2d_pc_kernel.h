// FDN_ALGO_WINO_BF16X3: the F(4,3) x F(4,3) kernel of conv64_wino2d_kernel.h with the Winograd-domain products on the bf16 matrix pipe
// (operands split exactly into three bf16 pieces, six cross terms, fp32 accumulation: see SPLIT there) as a PRODUCER / CONSUMER kernel.
//
// Why another structure.  On the bf16 pipe the K loop of a stage shrinks to 6/16 of its fp32-MFMA time, and what the barrier-synchronous
// kernel does between K loops -- load the stage's input rows, transform, split, write LDS -- does not: measured at (8,48^3) the K loops
// alone take 0.24 ms and the staging between them another 0.20 ms with two workgroups per CU (profiles/r6_wino_bf16x3.txt), because a
// workgroup that stages has its matrix pipe idle and the co-resident one is in the same position half of the time.  Here the two phases
// belong to different waves of ONE persistent workgroup per CU (512 threads, 8 waves, 2 per SIMD):
//   waves 0-3  consumers: wave w owns 32 cells x cout [16w, 16w+16) exactly as in the synchronous kernel: K loop on
//              v_mfma_f32_16x16x32_bf16, fold into the 128 output registers, epilogue.  They never touch global input rows.
//   waves 4-7  producers: thread (staged cell-plane r, 16-B chunk of 4 cin) loads the input rows of its item for the NEXT stage pass,
//              forms V = B_h^T x, applies B_w^T, splits every value into its three bf16 pieces and writes the six xw planes.  They hold no
//              accumulators, so an item's 24 chunks (two items for wave 4: 40 rows x 8 chunks = 320 items on 256 threads) are
//              requested in one go, right behind the transform of the previous pass -- the round trip runs while the wave waits at
//              the barrier for the consumers.
// LDS: two buffers of six planes (rows of 3 pieces x 32 cin + 32-B pad = 224 B: 54 KB each); in iteration k the producers fill buffer
// k & 1 with stage pass k while the consumers contract stage pass k - 1 out of the other one; ONE LDS-only barrier per iteration
// (vector-memory loads stay in flight across it).  A workgroup walks its tiles (XCD-aware order, tile i of workgroup g = g + i * grid)
// without leaving the loop: the producers run ahead into the next tile while the consumers write the finished one out.
// Every wave executes the same number of barriers (the iteration count is a kernel argument times a wave-uniform tile count).
#pragma once
#include <type_traits>
#include "conv64_wino2d_kernel.h"

namespace {

constexpr int kPcRow = kW2RowS;
constexpr int kPcPlane = kW2Rows * kPcRow + 64;
constexpr int kPcBuf = 6 * kPcPlane;
constexpr int kPcLds = 2 * kPcBuf;
constexpr int kPcThreads = 512;

template <bool FUSED>
__device__ __forceinline__ void conv64_wino2d_pc_body(const Wino2Args& p, char* const smem) {
    constexpr int NSS = 12;                                   // stage passes per tile: 6 H coordinates x 2 cin halves
    const int tid = threadIdx.x;
    const int wave_s = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const bool producer = wave_s >= 4;

    const int tiles_per_n = p.ntd * p.nth * p.ntw;
    const int T = p.N * tiles_per_n;
    const int G = (int)gridDim.x, wg = (int)blockIdx.x;
    const int nmine = (T - wg + G - 1) / G;                   // tiles wg, wg + G, ... (launcher: G <= T)
    const int K = nmine * NSS;

    // tile i of this workgroup -> sample and first output voxel (scalar; the XCD-aware order of conv64_wino2d_body applied to wg + i * G)
    auto tile_of = [&](int i, int& n, int& p0d, int& p0h, int& p0w) {
        int b = wg + i * G;
        if (!(FDN_DBG_BITS(p) & 128)) {
            const int qq = T >> 3, r = T & 7, xcd = b & 7;
            b = xcd * qq + min(xcd, r) + (b >> 3);
        }
        n = fdn_udiv40(b, p.mg_tpn_hi, p.mg_tpn_lo);
        b -= n * tiles_per_n;
        const int tdi = fdn_udiv40(b, p.mg_thw_hi, p.mg_thw_lo);
        b -= tdi * (p.nth * p.ntw);
        const int thi = fdn_udiv40(b, p.mg_ntw_hi, p.mg_ntw_lo);
        p0d = p.obd + tdi * p.td; p0h = p.obh + thi * p.ch * 4; p0w = p.obw + (b - thi * p.ntw) * p.cw * 4;
    };
    const unsigned sample_bytes = (unsigned)(p.ID * p.IH * p.IW) * 256u;

    if (producer) {
        // =========================== producers ===========================
        const int ptid = (wave_s - 4) * 64 + lane;
        const bool has2 = wave_s == 4;                        // scalar: the wave that carries the 64 extra items (rows 32 .. 39)
        const unsigned chunk16 = (unsigned)(ptid & 7) * 16u;
        unsigned ro[2][6], co[2][6];                          // per item: byte offsets of its six input rows / six input columns (kW2Big: reads zero)
        int vrow[2];                                          // LDS byte offset of the item's row chunk in a plane, -1: no such row in this tile
        __amdgpu_buffer_rsrc_t xrsrc;
        auto plan = [&](int i) {
            int n, p0d, p0h, p0w;
            tile_of(i, n, p0d, p0h, p0w);
            xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(p.x + (size_t)n * p.ID * p.IH * p.IW * 64), 0, sample_bytes, 0x00020000);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int r = (ptid >> 3) + 32 * it;
                const bool have = r < p.rows && (it == 0 || has2);
                vrow[it] = have ? r * kPcRow + (ptid & 7) * 8 : -1;
                const int rr = have ? r : 0;
                const int zd = fdn_div20(rr, p.mg_cpp);
                const int j = rr - zd * p.cpp;
                const int mh = fdn_div20(j, p.mg_cw);
                const int q0h = p0h - 1 + p.off + 4 * mh, q0w = p0w - 1 + p.off + 4 * (j - mh * p.cw);
                int qd = p0d - 1 + p.off + zd;
                bool okd = true;
                if (p.zero_mode) okd = (unsigned)qd < (unsigned)p.ID;
                else qd = min(max(qd, 0), p.ID - 1);
#pragma unroll
                for (int jj = 0; jj < 6; ++jj) {
                    int qh = q0h + jj;
                    bool ok = okd;
                    if (p.zero_mode) ok = ok && (unsigned)qh < (unsigned)p.IH;
                    else qh = min(max(qh, 0), p.IH - 1);
                    ro[it][jj] = ok ? (unsigned)((qd * p.IH + qh) * p.IW) * 256u : kW2Big;
                }
#pragma unroll
                for (int ii = 0; ii < 6; ++ii) {
                    int qw = q0w + ii;
                    bool ok = true;
                    if (p.zero_mode) ok = (unsigned)qw < (unsigned)p.IW;
                    else qw = min(max(qw, 0), p.IW - 1);
                    co[it][ii] = ok ? (unsigned)qw * 256u : kW2Big;
                }
            }
        };
        // the raw rows of the item in flight: [row a, b, c, d][column].  Item 0's rows cross the barrier (requested behind the previous pass's
        // transform); the 64 extra items of wave 4 are requested behind item 0's transform, into the same registers
        f32x4 X[1][4][6];
        auto issue_item = [&](int it, int ss) {
            const int xh = ss >> 1;
            const unsigned cb = chunk16 + (unsigned)(ss & 1) * 128u;
            // rows of B_h^T[xh]: stage 0: 0, 2, 4; stage 5: 1, 3, 5; stages 1 .. 4: 1, 2, 3, 4 (the coefficients differ)
            if (xh == 0 || xh == 5) {
                const int o = xh == 0 ? 0 : 1;
#pragma unroll
                for (int rw = 0; rw < 3; ++rw)
#pragma unroll
                    for (int ii = 0; ii < 6; ++ii)
                        X[0][rw][ii] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, (o ? ro[it][2 * rw + 1] : ro[it][2 * rw]) + co[it][ii] + cb, 0, 0));
            } else {
#pragma unroll
                for (int rw = 0; rw < 4; ++rw)
#pragma unroll
                    for (int ii = 0; ii < 6; ++ii)
                        X[0][rw][ii] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ro[it][rw + 1] + co[it][ii] + cb, 0, 0));
            }
        };
        auto issue = [&](int ss) {
            if (FDN_DBG_BITS(p) & 4) return;
            issue_item(0, ss);
        };
        auto transform_item = [&](int it, int ss, char* buf) {
            const int xh = ss >> 1;
            const bool ends = xh == 0 || xh == 5;
            const float ca = ends ? kPp : (xh == 1 ? -kPab2 : (xh == 2 ? kPab2 : (xh == 3 ? -kPa2b : kPa2b)));
            const float cbb = ends ? -kPs : (xh <= 2 ? -kPb2 : -kPa2);
            const float cc = ends ? 1.f : (xh == 1 ? kPa : (xh == 2 ? -kPa : (xh == 3 ? kPb : -kPb)));
            f32x4 v[6];
            if (ends) {
#pragma unroll
                for (int ii = 0; ii < 6; ++ii) v[ii] = ca * X[0][0][ii] + cbb * X[0][1][ii] + X[0][2][ii];
            } else {
#pragma unroll
                for (int ii = 0; ii < 6; ++ii) v[ii] = ca * X[0][0][ii] + cbb * X[0][1][ii] + (cc * X[0][2][ii] + X[0][3][ii]);
            }
            if (vrow[it] < 0) return;
            char* vp = buf + vrow[it];
            const f32x4 t1 = v[4] - kPb2 * v[2], t2 = kPa * v[3] - kPab2 * v[1];
            const f32x4 t3 = v[4] - kPa2 * v[2], t4 = kPb * v[3] - kPa2b * v[1];
            const f32x4 o[6] = {kPp * v[0] - kPs * v[2] + v[4], t1 + t2, t1 - t2, t3 + t4, t3 - t4, kPp * v[1] - kPs * v[3] + v[5]};
#pragma unroll
            for (int xw = 0; xw < 6; ++xw) {
                fdn_u32x2 hi, mid, lo;
                fdn_split3(o[xw], hi, mid, lo);
                *(fdn_u32x2*)(vp + xw * kPcPlane) = hi;
                *(fdn_u32x2*)(vp + xw * kPcPlane + 64) = mid;
                *(fdn_u32x2*)(vp + xw * kPcPlane + 128) = lo;
            }
        };
        plan(0);
        issue(0);
        int ss = 0, ti = 0;                                   // stage pass within the tile, tile
#pragma unroll 1
        for (int k = 0; k <= K; ++k) {
            if (k < K) {
                char* buf = smem + (k & 1) * kPcBuf;
                if (!(FDN_DBG_BITS(p) & 4)) {
                    transform_item(0, ss, buf);
                    if (has2) { issue_item(1, ss); transform_item(1, ss, buf); }
                }
                if (++ss == NSS) { ss = 0; ++ti; if (k + 1 < K) plan(ti); }
                if (k + 1 < K) issue(ss);                     // the next pass's rows fly while this wave waits at the barrier
            }
            fdn_barrier_lds();
        }
        return;
    }

    // =========================== consumers ===========================
    const int c = lane & 15, q = lane >> 4;
    const int ng = p.td * p.cpp;
    int abase[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        int m = mb * 16 + c;
        m = m < ng ? m : ng - 1;
        abase[mb] = m * kPcRow + q * 16;
    }
    const int tapstep = p.cpp * kPcRow;
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void*)p.up, 0, 4 * NSS * 18 * kW2UnitS, 0x00020000);
    const int wvoff_s = wave_s * (NSS * 18 * kW2UnitS) + lane * 16;
    const int ustride = (FDN_DBG_BITS(p) & 1) ? 0 : kW2UnitS;
    fdn_bf16x8 US[2][3], VS[3][2];
    auto ldu = [&](int slot, int step) {                      // step = (ss * 3 + kd) * 6 + xw of the wave's stream
#pragma unroll
        for (int pc = 0; pc < 3; ++pc)
            US[slot][pc] = __builtin_bit_cast(fdn_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(srs, wvoff_s + pc * 1024, step * ustride, 0));
    };
    auto ldv = [&](int pc, int o) {                           // o: byte offset of (buffer, depth tap, xw plane)
        VS[pc][0] = __builtin_bit_cast(fdn_bf16x8, *(const fdn_u32x4*)(smem + abase[0] + o + pc * 64));
        VS[pc][1] = __builtin_bit_cast(fdn_bf16x8, *(const fdn_u32x4*)(smem + abase[1] + o + pc * 64));
    };
    f32x4 Y[4][4][2];
    ldu(0, 0);
    int ss = 0, ti = 0;
    int n = 0, p0d = 0, p0h = 0, p0w = 0;
#pragma unroll 1
    for (int k = 0; k <= K; ++k) {
        if (k >= 1) {
            if (ss == 0) {
                tile_of(ti, n, p0d, p0h, p0w);
#pragma unroll
                for (int hr = 0; hr < 4; ++hr)
#pragma unroll
                    for (int wi = 0; wi < 4; ++wi)
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb) Y[hr][wi][mb] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            const int xh = ss >> 1;
            const int bo = ((k - 1) & 1) * kPcBuf;
            f32x4 acc[6][2];
#pragma unroll
            for (int xi = 0; xi < 6; ++xi) {
                acc[xi][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                acc[xi][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            // ---- K loop of the stage pass: 3 depth taps x 6 xw; one step = 32 cin x 6 terms x 2 M-blocks = 12 MFMAs of 16 cycles, fed by
            // 3 weight pieces (16 B per lane each, L1/L2; requested a step ahead -- across stage passes and tiles: the stream is walked
            // cyclically) and 3 x 2 cell pieces (ds_read_b128; each re-requested right behind its last use) ----
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) ldv(pc, bo);
#pragma unroll 1
            for (int kd = 0; kd < 3; ++kd) {
                const bool last = kd == 2;
                const int tapb = bo + kd * tapstep, tapb_n = last ? tapb : tapb + tapstep;
                const int st0 = (ss * 3 + kd) * 6;
                const int st_n = st0 + 6 == NSS * 18 ? 0 : st0 + 6;       // first step of the next tap / stage pass / tile
#pragma unroll
                for (int xw = 0; xw < 6; ++xw) {
                    const int su = xw & 1;
                    const int vo_n = xw < 5 ? tapb + (xw + 1) * kPcPlane : tapb_n;
                    auto mm = [&](int pu, int pv) {
                        acc[xw][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(US[su][pu], VS[pv][0], acc[xw][0], 0, 0, 0);
                        acc[xw][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(US[su][pu], VS[pv][1], acc[xw][1], 0, 0, 0);
                    };
                    mm(0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    ldu(su ^ 1, xw < 5 ? st0 + xw + 1 : st_n);
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
            // ---- fold the pass: t = A_w^T M, then Y_hr += A_h^T[hr][xh] t (conv64_wino2d_body) ----
            {
                const float sg = (xh & 1) ? 1.f : -1.f;
                const float m = xh <= 2 ? kPa : kPb;
                const bool mid = xh >= 1 && xh <= 4;
                const float ch[4] = {xh < 5 ? 1.f : 0.f, mid ? sg * m : 0.f, mid ? m * m : 0.f, mid ? sg * m * m * m : (xh == 5 ? 1.f : 0.f)};
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    const f32x4 s12 = acc[1][mb] + acc[2][mb], d12 = acc[1][mb] - acc[2][mb];
                    const f32x4 s34 = acc[3][mb] + acc[4][mb], d34 = acc[3][mb] - acc[4][mb];
                    f32x4 t[4];
                    t[0] = acc[0][mb] + s12 + s34;
                    t[1] = kPa * d12 + kPb * d34;
                    t[2] = kPa2 * s12 + kPb2 * s34;
                    t[3] = kPa3 * d12 + kPb3 * d34 + acc[5][mb];
#pragma unroll
                    for (int wi = 0; wi < 4; ++wi)
#pragma unroll
                        for (int hr = 0; hr < 4; ++hr) Y[hr][wi][mb] += ch[hr] * t[wi];
                }
            }
            if (++ss == NSS) {
                ss = 0; ++ti;
                // ---- epilogue of the tile (conv64_wino2d_body's, the cell indices formed in registers): lane = cell c of each M-block x
                // cout 16w + 4q .. + 3; 4 x 4 voxels per cell ----
                if (!(FDN_DBG_BITS(p) & 8)) {
                    const int cofs = wave_s * 16 + q * 4;
                    const float slope = p.act == FDN_ACT_RELU ? 0.f : (p.act == FDN_ACT_LEAKY ? p.alpha : 1.f);
                    int g0[2], gf0[2], hw0[2];
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        const int mi = mb * 16 + c;
                        int g = -1, gf = -1, hw = 0;
                        if (mi < ng) {
                            const int md = fdn_div20(mi, p.mg_cpp);
                            const int j = mi - md * p.cpp;
                            const int mh = fdn_div20(j, p.mg_cw);
                            const int pd = p0d + md, ph = p0h + 4 * mh, pw = p0w + 4 * (j - mh * p.cw);
                            if (pd < p.obd + p.ebd && ph < p.obh + p.ebh && pw < p.obw + p.ebw) {
                                g = ((n * p.OD + pd) * p.OH + ph) * p.OW + pw;
                                hw = ph | (pw << 16);
                                if (FUSED) {
                                    const int id = pd - 1;
                                    if (id >= 1 && id <= p.ID - 2) gf = ((n * p.ID + id) * p.IH + (ph - 1)) * p.IW + (pw - 1);
                                }
                            }
                        }
                        g0[mb] = g; gf0[mb] = gf; hw0[mb] = hw;
                    }
                    constexpr int HG = 1;                     // rows of one M-block per group (the weight fragments of the next pass stay live beside Y: one row's operands fit)
                    constexpr int GPB = 4 / HG, NG = 2 * GPB;
                    if (FUSED) {
                        int fi[2][HG][4];
                        f32x4 sk[2][HG][4], ym[2][HG][4];
                        auto fload = [&](int g, int buf) {
                            const int mb = g / GPB, h0 = (g % GPB) * HG;
                            const int ph = hw0[mb] & 0xffff, pw = hw0[mb] >> 16;
#pragma unroll
                            for (int hr = 0; hr < HG; ++hr)
#pragma unroll
                                for (int wi = 0; wi < 4; ++wi) {
                                    const int ih = ph + h0 + hr - 1, iw = pw + wi - 1;
                                    const bool in = g0[mb] >= 0 && gf0[mb] >= 0 && ih >= 1 && ih <= p.IH - 2 && iw >= 1 && iw <= p.IW - 2;
                                    fi[buf][hr][wi] = in ? gf0[mb] + (h0 + hr) * p.IW + wi : -1;
                                    const size_t o = (size_t)(in ? fi[buf][hr][wi] : 0) * 64 + cofs;
                                    sk[buf][hr][wi] = p.fskip ? *(const f32x4*)(p.fskip + o) : (f32x4){0.f, 0.f, 0.f, 0.f};
                                    ym[buf][hr][wi] = p.fy ? *(const f32x4*)(p.fy + o) : (f32x4){1.f, 1.f, 1.f, 1.f};
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
                                    for (int e = 0; e < 4; ++e) v[e] = in ? (z[e] + sk[buf][hr][wi][e]) * (ym[buf][hr][wi][e] > 0.f ? 1.f : slope) : z[e];
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
                        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                        if (p.bias) bv = *(const f32x4*)(p.bias + cofs);
                        // (two copies of the group loop, chosen once by a scalar branch: with the residual tested per group hipcc spills its rows)
                        auto run = [&](auto has_res) {
                            constexpr bool RES = decltype(has_res)::value;
                            f32x4 rv[2][HG][4];
                            auto rload = [&](int g, int buf) {
                                const int mb = g / GPB, h0 = (g % GPB) * HG;
#pragma unroll
                                for (int hr = 0; hr < HG; ++hr)
#pragma unroll
                                    for (int wi = 0; wi < 4; ++wi)
                                        rv[buf][hr][wi] = *(const f32x4*)(p.res + (size_t)((g0[mb] >= 0 ? g0[mb] : 0) + (h0 + hr) * p.OW + wi) * 64 + cofs);
                            };
                            if constexpr (RES) rload(0, 0);
#pragma unroll
                            for (int g = 0; g < NG; ++g) {
                                if constexpr (RES) { if (g + 1 < NG) rload(g + 1, (g + 1) & 1); }
                                const int mb = g / GPB, h0 = (g % GPB) * HG, buf = g & 1;
                                if (g0[mb] < 0) continue;
#pragma unroll
                                for (int hr = 0; hr < HG; ++hr)
#pragma unroll
                                    for (int wi = 0; wi < 4; ++wi) {
                                        f32x4 v = Y[h0 + hr][wi][mb] + bv;
                                        if constexpr (RES) v += rv[buf][hr][wi];
#pragma unroll
                                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], slope * v[e]);
                                        *(f32x4*)(p.y + (size_t)(g0[mb] + (h0 + hr) * p.OW + wi) * 64 + cofs) = v;
                                    }
                            }
                        };
                        if (p.res) run(std::true_type{}); else run(std::false_type{});
                    }
                }
            }
        }
        fdn_barrier_lds();
    }
}

}  // namespace

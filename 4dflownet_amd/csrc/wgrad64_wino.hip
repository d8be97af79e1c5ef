// Weight gradient of the 3x3x3 64 -> 64 conv (Conv3DBackpropFilterV2 behind tape.gradient,
// src/Network/TrainerController.py:223, for the layers of src/Network/SR4DFlowNet.py:18-46) with Winograd F(3,4) along W:
//   dW[a,b,t][ci][co] = sum_{n,d,h,w} x[n, clamp(d+a-1), clamp(h+b-1), clamp(w+t-1)][ci] * dz[n,d,h,w][co]
// For a group of 4 consecutive w the 3 taps t need 12 products; with the 6 input voxels x[4p-1 .. 4p+4] transformed by the
// same B^T as the forward kernel (conv64_wino.hip) and the 4 gradients by G' (6x4), 6 products suffice:
//   M_xi += (B^T x)_xi (G' dz)_xi   summed over all groups,     dW[.,.,t] = sum_xi A'^T[t][xi] M_xi      (A'^T is 3x6)
// and because the output transform is linear it runs ONCE, in the reduction kernel, on the accumulated 64x64 matrices.
// On MI355X the fp32 matrix rate equals the fp32 vector rate, so halving the multiplies is the only way past the fp32 MFMA
// roofline of the direct kernel (wgrad64_mfma.hip, 0.88 of peak).
//
// Decomposition: grid = (S splits of the tile list) x (3 kernel-depth taps a); ONE workgroup of 8 waves per CU.
//   A workgroup walks tiles of 1 x 6 x 8 voxels (6 lines of 2 groups) and accumulates the 18 matrices M[b][xi] (b = height
//   tap, xi = Winograd coordinate) of its depth tap: wave (quadrant, parity) owns a 32x32 (ci,co) quadrant of the 9 matrices
//   with xi = 2 xp + parity -- 144 accumulator registers that stay resident across all tiles, two waves per SIMD.  (Four
//   waves with all 18 quadrant accumulators = 288 registers exceed the 256 AGPRs: hipcc then shuttles ~60 accumulator tiles
//   per tile between the register files, 980 v_accvgpr moves per loop body -- measured no faster than the direct kernel.)
//   K = groups: one v_mfma_f32_32x32x2_f32 contracts the two groups of a line; its operands are single floats per lane, stored
//   in LDS as [line][group][xi pair][channel][parity] (conflict-free ds_read_b32: 12 reads per 9 MFMAs; the direct kernel 10).
//   Transform on the way in: waves 0-3: thread (halo line, group, 16-B channel chunk) loads the 6 x chunks (edge clamp applied),
//   waves 4-6: thread (line, group, chunk) the 4 dz chunks (zero outside the volume); each forms its 6 transformed chunks and
//   writes them to LDS.
//   Tile pipeline through THREE LDS buffers exactly as in the direct kernel: while tile k is contracted, the raw registers of
//   tile k+1 are transformed and written to buffer (k+1)%3, the raw rows of tile k+2 are loaded, one barrier per tile.
//   Partials M[S][a][b][xi] go to the workspace; wgrad64_wino_reduce_kernel sums over S and applies A'^T.
#include "fdn_common.h"

namespace {

struct WgWinoArgs {
    const float* x;
    const float* dz;
    float* partial;
    int N, D, H, W;
    int nth, ntw, ntiles, S;
    unsigned bytes;              // size of x (= of dz) in bytes; < 4 GB (checked by the launcher)
};

constexpr int WTH = 6, WTG = 2, WTW = 4 * WTG;          // tile: 1 x 6 x 8 voxels
constexpr int GROWB = 1536;                             // bytes per (line, group): 3 xi pairs x 64 channels x 2 floats
constexpr int VBYTES = (WTH + 2) * WTG * GROWB;         // transformed x: 8 halo lines
constexpr int ZBYTES = WTH * WTG * GROWB;               // transformed dz
constexpr int WBUFB = VBYTES + ZBYTES;                  // 43 008 B; three buffers = 126 KB

__global__ __launch_bounds__(512, 1) void wgrad64_wino_kernel(WgWinoArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int kh = lane >> 5;
    const int mq = wave & 1, nq = (wave >> 1) & 1;
    const int eh = wave >> 2;        // parity of the Winograd coordinates this wave accumulates: xi = 2 xp + eh
    const int a = blockIdx.y;        // kernel-depth tap
    const int split = blockIdx.x;
    const int c16 = tid & 15;        // 16-B chunk (4 channels) of a 256-B row
    const int ig = (tid >> 4) & 1;   // group of this thread's transform item
    const int il = (tid >> 5) & 7;   // line of the item: waves 0-3: x halo lines 0..7; waves 4-6: dz lines 0..5
    const bool xitem = tid < 256;    // wave-uniform
    const bool zitem = tid >= 256 && il < WTH;

    f32x16 acc[3][3];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int xp = 0; xp < 3; ++xp)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][xp][r] = 0.f;

    // tile walk: tile = split + k*S, decoded incrementally (n, d, th, tw) with S pre-split the same way -- scalar work only
    const int per_d = p.nth * p.ntw, per_n = p.D * per_d;
    int tn, td, th, tw;
    {
        int b = split;
        tn = b / per_n; b -= tn * per_n;
        td = b / per_d; b -= td * per_d;
        th = b / p.ntw; tw = b - th * p.ntw;
    }
    int sn, sd, sh, sw;
    {
        int b = p.S;
        sn = b / per_n; b -= sn * per_n;
        sd = b / per_d; b -= sd * per_d;
        sh = b / p.ntw; sw = b - sh * p.ntw;
    }
    const int nk = (p.ntiles - split + p.S - 1) / p.S;      // tiles of this workgroup (>= 1 by construction of S)
    int kload = 0;
    auto advance = [&]() {                                  // cursor -> next tile of this workgroup (stays on the last one)
        if (kload + 1 < nk) {
            ++kload;
            tw += sw; if (tw >= p.ntw) { tw -= p.ntw; ++th; }
            th += sh; if (th >= p.nth) { th -= p.nth; ++td; }
            td += sd; if (td >= p.D) { td -= p.D; ++tn; }
            tn += sn;
        }
    };

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.dz, 0, p.bytes, 0x00020000);
    f32x4 xr[6], zr[4];
    auto bload = [&](__amdgpu_buffer_rsrc_t r, unsigned vo) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)vo, 0, 0));
    };
    // raw rows of the cursor tile: x voxels 4g-1 .. 4g+4 of halo line il (edge clamp == SYMMETRIC p=1), dz voxels 4g .. 4g+3
    auto load_x = [&](int nn) {
        if (!xitem) return;
        const int qd = min(max(td + a - 1, 0), p.D - 1);
        const int qh = min(max(th * WTH - 1 + il, 0), p.H - 1);
        const int qw = min(max(tw * WTW + 4 * ig - 1 + nn, 0), p.W - 1);
        xr[nn] = bload(xrs, (unsigned)((((tn * p.D + qd) * p.H + qh) * p.W + qw) * 256 + c16 * 16));
    };
    auto load_z = [&](int j) {
        if (!zitem) return;
        const int qh = th * WTH + il, qw = tw * WTW + 4 * ig + j;
        const bool ok = qh < p.H && qw < p.W;
        zr[j] = bload(zrs, ok ? (unsigned)((((tn * p.D + td) * p.H + qh) * p.W + qw) * 256 + c16 * 16) : 0xffffffffu);
    };
    // transformed pairs (xi = 2 xp, 2 xp + 1) of this thread's 4 channels -> [line][group][xp][channel][2]
    auto put = [&](char* dst, f32x4 va, f32x4 vb) {
        *(f32x4*)dst = (f32x4){va.x, vb.x, va.y, vb.y};
        *(f32x4*)(dst + 16) = (f32x4){va.z, vb.z, va.w, vb.w};
    };
    auto write_v = [&](int xp, char* buf) {
        // B^T of F(4,3)/F(3,4): (4,0,-5,0,1,0) (0,-4,-4,1,1,0) (0,4,-4,-1,1,0) (0,-2,-1,2,1,0) (0,2,-1,-2,1,0) (0,4,0,-5,0,1)
        if (!xitem) return;
        char* dst = buf + (il * WTG + ig) * GROWB + xp * 512 + c16 * 32;
        const f32x4 x0 = xr[0], x1 = xr[1], x2 = xr[2], x3 = xr[3], x4 = xr[4], x5 = xr[5];
        if (xp == 0) put(dst, 4.f * x0 - 5.f * x2 + x4, (x4 - 4.f * x2) + (x3 - 4.f * x1));
        else if (xp == 1) put(dst, (x4 - 4.f * x2) - (x3 - 4.f * x1), (x4 - x2) + 2.f * (x3 - x1));
        else put(dst, (x4 - x2) - 2.f * (x3 - x1), 4.f * x1 - 5.f * x3 + x5);
    };
    auto write_z = [&](int xp, char* buf) {
        // G' of F(3,4): (1/4,0,0,0) -1/6(1,1,1,1) -1/6(1,-1,1,-1) 1/24(1,2,4,8) 1/24(1,-2,4,-8) (0,0,0,1)
        if (!zitem) return;
        char* dst = buf + VBYTES + (il * WTG + ig) * GROWB + xp * 512 + c16 * 32;
        const f32x4 z0 = zr[0], z1 = zr[1], z2 = zr[2], z3 = zr[3];
        const float s6 = -1.f / 6, s24 = 1.f / 24;
        if (xp == 0) put(dst, 0.25f * z0, s6 * ((z0 + z2) + (z1 + z3)));
        else if (xp == 1) put(dst, s6 * ((z0 + z2) - (z1 + z3)), s24 * ((z0 + 4.f * z2) + (2.f * z1 + 8.f * z3)));
        else put(dst, s24 * ((z0 + 4.f * z2) - (2.f * z1 + 8.f * z3)), z3);
    };

    // ---- prologue: tile 0 -> buffer 0, tile 1 -> registers ----
#pragma unroll
    for (int nn = 0; nn < 6; ++nn) load_x(nn);
#pragma unroll
    for (int j = 0; j < 4; ++j) load_z(j);
#pragma unroll
    for (int xp = 0; xp < 3; ++xp) { write_v(xp, smem); write_z(xp, smem); }
    advance();
#pragma unroll
    for (int nn = 0; nn < 6; ++nn) load_x(nn);
#pragma unroll
    for (int j = 0; j < 4; ++j) load_z(j);
    __syncthreads();

    // operands: lane (li, kh) = channel (mq*32 + li) resp. (nq*32 + li) of group kh, coordinate parity eh
    const int lane_v = kh * GROWB + (mq * 32 + li) * 8 + eh * 4;
    const int lane_z = VBYTES + kh * GROWB + (nq * 32 + li) * 8 + eh * 4;
    float V[2][3], Z[2][3];
    auto issue_v = [&](const char* buf, int line, float (&v)[3]) {
#pragma unroll
        for (int xp = 0; xp < 3; ++xp) v[xp] = *(const float*)(buf + lane_v + line * (WTG * GROWB) + xp * 512);
    };
    auto issue_z = [&](const char* buf, int line, float (&z)[3]) {
#pragma unroll
        for (int xp = 0; xp < 3; ++xp) z[xp] = *(const float*)(buf + lane_z + line * (WTG * GROWB) + xp * 512);
    };
    int bcur = 0;
    issue_z(smem, 0, Z[0]);
    issue_v(smem, 0, V[0]);
#pragma unroll 1
    for (int k = 0; k < nk; ++k) {
        const int bnxt = bcur == 2 ? 0 : bcur + 1;
        const char* cur = smem + bcur * WBUFB;
        char* nxt = smem + bnxt * WBUFB;
        // 18 slots: slot s = (dz line q = s/3, height tap b = s%3) -> x halo line q + b; 3 MFMAs per wave and slot
#pragma unroll
        for (int s = 0; s < 18; ++s) {
            const int q = s / 3, b = s % 3;
            if (s == 12) __syncthreads();       // tile k+1 is complete in `nxt`; every wave is done with tile k-1's buffer
            // read-ahead: operands of the next slot (the first slot of tile k+1 at the end)
            if (s + 1 < 18) {
                issue_v(cur, (s + 1) / 3 + (s + 1) % 3, V[(s + 1) & 1]);
                if ((s + 1) % 3 == 0) issue_z(cur, (s + 1) / 3, Z[((s + 1) / 3) & 1]);
            } else {
                issue_v(nxt, 0, V[0]);
                issue_z(nxt, 0, Z[0]);
            }
            __builtin_amdgcn_sched_barrier(0);
            // pipeline stages, pinned to slots: transform + write tile k+1, then load the raw rows of tile k+2
            if (s < 3) write_v(s, nxt);
            else if (s < 6) write_z(s - 3, nxt);
            else if (s == 6) advance();
            else if (s < 13) load_x(s - 7);
            else if (s < 17) load_z(s - 13);
#pragma unroll
            for (int xp = 0; xp < 3; ++xp)
                acc[b][xp] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[s & 1][xp], Z[q & 1][xp], acc[b][xp], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        bcur = bnxt;
    }

    // ---- this workgroup's partial M[a][b][xi] ----
    float* out = p.partial + ((size_t)split * 3 + a) * 18 * 4096;
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int xp = 0; xp < 3; ++xp)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = mq * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                out[(size_t)(b * 6 + 2 * xp + eh) * 4096 + ci * 64 + nq * 32 + li] = acc[b][xp][r];
            }
}

// dw[a][b][t][e] = sum_xi A'^T[t][xi] sum_s partial[s][a][b][xi][e],  A'^T = (1,1,1,1,1,0) (0,1,-1,2,-2,0) (0,1,1,4,4,1).
// Block = 64 float4 columns of one (a,b) x 4 quarters of S, combined through LDS in a fixed order (deterministic).
__global__ __launch_bounds__(256) void wgrad64_wino_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dw, int S) {
    __shared__ f32x4 red[3][6][64];
    const int col = threadIdx.x & 63, qtr = threadIdx.x >> 6;
    const int ab = blockIdx.x >> 4;                              // 9 (a,b) x 16 blocks of 64 columns
    const int e4 = (blockIdx.x & 15) * 64 + col;                 // float4 column within the 64x64 matrix
    const f32x4* p = (const f32x4*)partial + (size_t)ab * 6 * 1024 + e4;
    const int s0q = (S * qtr) >> 2, s1q = (S * (qtr + 1)) >> 2;
    f32x4 m[6];
#pragma unroll
    for (int xi = 0; xi < 6; ++xi) m[xi] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int s = s0q; s < s1q; ++s)
#pragma unroll
        for (int xi = 0; xi < 6; ++xi) m[xi] += p[(size_t)s * (54 * 1024) + xi * 1024];
    if (qtr) {
#pragma unroll
        for (int xi = 0; xi < 6; ++xi) red[qtr - 1][xi][col] = m[xi];
    }
    __syncthreads();
    if (qtr == 0) {
#pragma unroll
        for (int xi = 0; xi < 6; ++xi) m[xi] = (m[xi] + red[0][xi][col]) + (red[1][xi][col] + red[2][xi][col]);
        const f32x4 s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
        f32x4* o = (f32x4*)dw + (size_t)ab * 3 * 1024 + e4;
        o[0] = m[0] + s12 + s34;
        o[1024] = d12 + 2.f * d34;
        o[2048] = s12 + 4.f * s34 + m[5];
    }
}

int wgrad64_wino_splits(int N, int D, int H, int W) {
    const long long ntiles = (long long)N * D * ((H + WTH - 1) / WTH) * ((W + WTW - 1) / WTW);
    long long S = 85;                  // 3 * 85 = 255 workgroups: one (8 waves, 126 KB of LDS) per CU
    if (ntiles < S) S = ntiles > 0 ? ntiles : 1;
    return (int)S;
}

}  // namespace

size_t fdn_wgrad64_wino_workspace_bytes(int N, int D, int H, int W) {
    return (size_t)wgrad64_wino_splits(N, D, H, W) * 54 * 4096 * sizeof(float);
}

int fdn_wgrad64_wino_launch(const float* x, const float* dz, float* dw, void* ws, size_t ws_bytes, int N, int D, int H,
                            int W, hipStream_t s) {
    WgWinoArgs a;
    a.x = x; a.dz = dz; a.partial = (float*)ws;
    a.N = N; a.D = D; a.H = H; a.W = W;
    a.nth = (H + WTH - 1) / WTH; a.ntw = (W + WTW - 1) / WTW;
    a.ntiles = N * D * a.nth * a.ntw;
    a.S = wgrad64_wino_splits(N, D, H, W);
    FDN_REQUIRE((long long)N * D * H * W * 256 < (1ll << 32), "wgrad64: x of %dx%dx%dx%dx64 floats exceeds the 32-bit buffer addressing", N, D, H, W);
    FDN_REQUIRE(ws_bytes >= (size_t)a.S * 54 * 4096 * sizeof(float), "wgrad64 (winograd): workspace too small");
    a.bytes = (unsigned)((long long)N * D * H * W * 256);
    const size_t lds = (size_t)3 * WBUFB;
    if (int rc = fdn_func_max_lds((const void*)wgrad64_wino_kernel, (int)lds, "wgrad64_wino")) return rc;
    hipLaunchKernelGGL(wgrad64_wino_kernel, dim3(a.S, 3), dim3(512), lds, s, a);
    FDN_CHECK_LAUNCH("wgrad64_wino_kernel");
    hipLaunchKernelGGL(wgrad64_wino_reduce_kernel, dim3(9 * 16), dim3(256), 0, s, (const float*)ws, dw, a.S);
    FDN_CHECK_LAUNCH("wgrad64_wino_reduce_kernel");
    return FDN_OK;
}

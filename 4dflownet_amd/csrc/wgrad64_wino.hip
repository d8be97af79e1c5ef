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
// Decomposition: grid = (S splits of the tile list) x (3 kernel-depth taps a), XCD-aware order; ONE workgroup of 8 waves per CU.
//   A workgroup walks tiles of 1 x 6 x 8 voxels (6 lines of 2 groups) and accumulates the 18 matrices M[b][xi] (b = height
//   tap, xi = Winograd coordinate) of its depth tap: wave (quadrant, parity) owns a 32x32 (ci,co) quadrant of the 9 matrices
//   with xi = 2 xp + parity -- 144 accumulator registers that stay resident across all tiles, two waves per SIMD.  (Four
//   waves with all 18 quadrant accumulators = 288 registers exceed the 256 AGPRs: hipcc then shuttles ~60 accumulator tiles
//   per tile between the register files, 980 v_accvgpr moves per loop body -- measured no faster than the direct kernel.)
//   K = groups: one v_mfma_f32_32x32x2_f32 contracts the two groups of a line; its operands are single floats per lane, stored
//   in LDS per (line, group) and parity as a pair plane + a single plane (one ds_read_b64 + one ds_read_b32 fetch a wave's
//   three coordinates: 8 reads per 9 MFMAs; the direct kernel needs 10).
//   Transform on the way in: waves 0-3: thread (halo line, group, 16-B channel chunk) loads the 6 x chunks (edge clamp applied),
//   waves 4-6: thread (line, group, chunk) the 4 dz chunks (zero outside the volume); each forms its 6 transformed chunks and
//   writes them to LDS.
//   Tile pipeline through THREE LDS buffers exactly as in the direct kernel: while tile k is contracted, the raw registers of
//   tile k+1 are transformed and written to buffer (k+1)%3, the raw rows of tile k+2 are loaded, one barrier per tile.
//   The output transform runs inside the workgroup; partials dW[S][27] go to the workspace and are summed by the direct
//   kernel's reduction (wgrad64_reduce_kernel: same partial layout).
//
// Round 4 -- F(3,2) along D on top of it (DEP = true, D even): two dz planes d0, d0+1 and the three depth taps need 6 plane
// products; with the four x planes x[d0-1 .. d0+2] combined by the B^T of F(2,3) (x0-x2, x1+x2, x2-x1, x1-x3), the dz planes by
// A (z0, z0+z1, z0-z1, -z1) four suffice:  M[xd] += V[xd] (x) Z[xd],  dW[a] = sum_xd G^T[a][xd] M[xd]  (G^T = (1,1/2,1/2,0)
// (0,1/2,-1/2,0) (0,1/2,1/2,1)) -- a third fewer multiplies again.  The grid becomes S splits x 4 depth COORDINATES xd; a workgroup
// walks tiles of plane PAIRS and is otherwise the kernel above: every transform item loads its chunk from TWO planes and combines
// them before the W transform.  x's second plane travels by direct-to-LDS loads into a private 24-KB scratch (no registers; the
// kernel has none to spare), dz's second plane in two extra registers.  The depth output transform runs in the reduction kernel
// (wgrad64_reduce_dep_kernel), which sums the S x 4 partials with the G^T weights.
#include "fdn_common.h"
#include <type_traits>

namespace {

struct WgWinoArgs {
    const float* x;
    const float* dz;
    float* partial;
    int N, D, H, W;
    int DT;                      // depth units of the tile walk: D planes, or D / 2 plane pairs (DEP)
    int nth, ntw, ntiles, S;
    unsigned bytes;              // size of x (= of dz) in bytes; < 4 GB (checked by the launcher)
    int dbg;                     // ablation bits (test build): 1 = no raw loads, 2 = no transform / LDS writes, 4 = no LDS operand reads, 8 = no XCD placement, 16 = depth-fastest tile order, 32 = no tile walk (every iteration on the first tiles)
};

FDN_HOOK_VAR(int, fdn_wgrad64_wino_dbg, 0);

constexpr int WTH = 6, WTG = 2, WTW = 4 * WTG;          // tile: 1 x 6 x 8 voxels
constexpr int GROWB = 1536;                             // bytes per (line, group): 2 parities x (64 ch x 2 floats + 64 ch x 1 float)
constexpr int VBYTES = (WTH + 2) * WTG * GROWB;         // transformed x: 8 halo lines
constexpr int ZBYTES = WTH * WTG * GROWB;               // transformed dz
constexpr int WBUFB = VBYTES + ZBYTES;                  // 43 008 B; three buffers = 126 KB
constexpr int kLocSlot = 5, kXSlot = 6, kZSlot = 13;      // pipeline slots: tile walk / raw x rows / raw dz rows of tile k+2
constexpr int kLocSlotZ = 10;                             // ... the walk of the dz waves: the two waves of a SIMD (w, w + 4: one x wave, one dz wave) must
                                                          // not sit in their ~50 scalar instructions at the same time -- the SIMD then issues no MFMA
constexpr int XSCRATCH = 6 * 4096;                      // DEP: x chunks of the second plane, [chunk 6][x-thread 256] x 16 B, filled by LDS-DMA
constexpr int ZSCRATCH = 2 * 3072;                      // DEP: dz chunks 2, 3 of the second plane, [2][dz-thread 192] x 16 B

// (a __device__ body + thin __global__ wrappers: one layer per launch, or several layers of the same grid in ONE launch -- the
// batched form, wgrad64_wino_batch_kernel below.  q = the workgroup's place in the XCD-aware order of its layer, see wgrad64_place.)
template <bool DEP>
__device__ __forceinline__ void wgrad64_wino_body(const WgWinoArgs& p, const int q, char* const smem) {
    constexpr int NP = DEP ? 4 : 3;                     // workgroups per split: depth coordinates xd (DEP) or depth taps a
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int li = lane & 31;
    const int kh = lane >> 5;
    const int mq = wave & 1, nq = (wave >> 1) & 1;
    const int eh = wave >> 2;        // parity of the Winograd coordinates this wave accumulates: xi = 2 xp + eh
    // XCD-aware placement: 1-D grid of 3 S workgroups whose ids are dealt round-robin to the 8 XCDs (each with its own L2).  The
    // (split, tap) pairs q = 3 split + a are cut into 8 contiguous ranges, one per XCD: the three depth taps of a split (same dz
    // tile) and ~10 consecutive splits (w / h neighbours: shared halo lines) run on ONE XCD at the same time.  Measured at (8,48^3):
    // HBM reads 763 -> 553 MB per launch (x + dz = 453 MB), 0.889 -> 0.865 ms.  (A depth-fastest tile order -- the 3 x 10 workgroups of an
    // XCD then share x planes as well -- cuts the reads to 353 MB but runs 0.90 ms: the concurrent tiles then differ by multiples of
    // the plane stride, 9 x 64 KB at 48^3, and pile onto the same memory channels.  Test-build bit 16.)
    const int split = q / NP;
    const int a = q - NP * split;        // kernel-depth tap, or depth coordinate xd (DEP)
    const int c16 = tid & 15;        // 16-B chunk (4 channels) of a 256-B row
    const int ig = (tid >> 4) & 1;   // group of this thread's transform item
    const int il = (tid >> 5) & 7;   // line of the item: waves 0-3: x halo lines 0..7; waves 4-6: dz lines 0..5
    // transform items are whole waves (x: waves 0-3, lines 0..7; dz: waves 4-6, lines 0..5; wave 7 has none): the flags are
    // formed from the scalar wave id so that the item code sits behind scalar branches, not exec masks
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const bool xitem = wave_s < 4;
    const bool zitem = wave_s >= 4 && wave_s < 7;
    static_assert(WTH == 6 && WTG == 2, "item-to-wave map above");

    f32x16 acc[3][3];
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int xp = 0; xp < 3; ++xp)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][xp][r] = 0.f;

    // tile walk: tile = split + k*S in the order (n, d, th, tw), decoded incrementally with S pre-split the same way: scalar work only
    const bool dfast = (FDN_DBG_BITS(p) & 16) != 0;
    const int per_c = p.DT, per_r = p.ntw * per_c, per_n = p.nth * per_r;
    const int per_d = p.nth * p.ntw;
    int tn, td, th, tw;
    int sn, sd, sh, sw;
    if (dfast) {
        int b = split;
        tn = b / per_n; b -= tn * per_n;
        th = b / per_r; b -= th * per_r;
        tw = b / per_c; td = b - tw * per_c;
        b = p.S;
        sn = b / per_n; b -= sn * per_n;
        sh = b / per_r; b -= sh * per_r;
        sw = b / per_c; sd = b - sw * per_c;
    } else {
        int b = split;
        tn = b / per_n; b -= tn * per_n;
        td = b / per_d; b -= td * per_d;
        th = b / p.ntw; tw = b - th * p.ntw;
        b = p.S;
        sn = b / per_n; b -= sn * per_n;
        sd = b / per_d; b -= sd * per_d;
        sh = b / p.ntw; sw = b - sh * p.ntw;
    }
    const int nk = (p.ntiles - split + p.S - 1) / p.S;      // tiles of this workgroup (>= 1 by construction of S)
    int kload = 0;
    auto advance = [&]() {                                  // cursor -> next tile of this workgroup (stays on the last one)
        if (kload + 1 < nk) {
            ++kload;
            if (dfast) {
                td += sd; if (td >= p.DT) { td -= p.DT; ++tw; }
                tw += sw; if (tw >= p.ntw) { tw -= p.ntw; ++th; }
                th += sh; if (th >= p.nth) { th -= p.nth; ++tn; }
            } else {
                tw += sw; if (tw >= p.ntw) { tw -= p.ntw; ++th; }
                th += sh; if (th >= p.nth) { th -= p.nth; ++td; }
                td += sd; if (td >= p.DT) { td -= p.DT; ++tn; }
            }
            tn += sn;
        }
    };

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.dz, 0, p.bytes, 0x00020000);
    // the same two buffers for the direct-to-LDS loads of the second plane, issued OUTSIDE the compiler's bookkeeping (fdn_common.h):
    // hipcc counts a builtin LDS-DMA as a pending LDS write and puts s_waitcnt vmcnt(0) in front of the next LDS read it cannot prove
    // disjoint -- here the operand reads of the very next slot, i.e. a full memory round trip inside the MFMA stream after EVERY raw-row
    // load.  (Round 4's build happened to escape it; the constants of round 5's interpolation points did not: 0.60 -> 0.75 ms until
    // the ISA was read.)  The kernel owns the ordering: the scratch is private to the issuing wave, read back behind the explicit
    // s_waitcnt vmcnt(0) of combine_x / combine_z, and refilled only after those reads have been consumed by the transform.
    const fdn_i32x4 xrs_raw = fdn_raw_rsrc(p.x, p.bytes), zrs_raw = fdn_raw_rsrc(p.dz, p.bytes);
    f32x4 raw[6];                         // this thread's raw rows: 6 x chunks (waves 0-3) or 4 dz chunks (waves 4-6; DEP: + chunks 0, 1 of the second plane)
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    char* const xscr = smem + 3 * WBUFB + wave_u * 1024;     // DEP: this wave's 1-KB pieces of the x scratch (chunk nn at + nn * 4096)
    char* const zscr = smem + 3 * WBUFB + XSCRATCH + (wave_u & 3) * 1024;   // ... and of the dz scratch (chunks 2, 3 of the second plane at + 0 / + 3072)
    // DEP: the two planes of this workgroup's depth coordinate and their weights: V = xA + sx xB, Z = cA zA + cB zB
    // (the second plane must not lie BELOW the first: a buffer load's scalar offset cannot be negative -- coordinate 2 is therefore
    // formed as -(x2 - x1) = x1 - x2 with the sign moved to the dz side, -(z0 - z1))
    const float sx = a == 1 ? 1.f : -1.f;
    const float cA = a >= 2 ? -1.f : 1.f, cB = 1.f;
    const bool has_zb = DEP && (a == 1 || a == 2);
    unsigned dxb = 0, dzb = 0;            // byte distance of the second plane from the first (>= 0)
    auto bload = [&](__amdgpu_buffer_rsrc_t r, unsigned vo, unsigned so) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)vo, (int)so, 0));
    };
    // Raw rows of the cursor tile: x voxels 4g-1 .. 4g+4 of halo line il (edge clamp == SYMMETRIC p=1), dz voxels 4g .. 4g+3
    // (zero outside the volume).  The (sample, plane, tile origin) part of an address is a scalar offset and the (line, voxel,
    // chunk) part a per-thread constant, so a tile whose halo box lies inside its plane costs NO vector ALU work (every VALU
    // instruction takes its cycles from the fp32 MFMA stream of its SIMD); border tiles clamp per thread.
    // per-thread constant part, relative to the tile's halo origin (x) / origin (dz); the voxel part nn * 256 is an instruction immediate
    const unsigned tc0 = (unsigned)((il * p.W + 4 * ig) * 256 + c16 * 16);
    unsigned rowoff = 0;                  // border tiles: byte offset of this thread's (clamped) line, chunk included
    bool x_in = false, z_in = false;
    unsigned x_so = 0, z_so = 0;
    auto locate = [&]() {                 // once per tile, after advance(): scalar unless the tile touches the plane border
        int qd, zd;
        if (DEP) {
            // plane pair (d0, d0 + 1); x planes of coordinate xd: (d0-1, d0+1) (d0, d0+1) (d0, d0+1) (d0, d0+2), edge-clamped;
            // dz planes: d0 | d0, d0+1 | d0, d0+1 | d0+1
            const int d0 = 2 * td;
            const int pa = a == 0 ? d0 - 1 : d0, pb = a == 3 ? d0 + 2 : d0 + 1;
            qd = min(max(pa, 0), p.D - 1);
            dxb = (unsigned)((min(max(pb, 0), p.D - 1) - qd) * p.H * p.W) * 256u;
            zd = a == 3 ? d0 + 1 : d0;
            dzb = (unsigned)(p.H * p.W) * 256u;
        } else {
            qd = min(max(td + a - 1, 0), p.D - 1);
            zd = td;
        }
        const int h0 = th * WTH - 1, w0 = tw * WTW - 1;
        const unsigned xplane = (unsigned)((tn * p.D + qd) * p.H * p.W) * 256u;
        const unsigned zplane = (unsigned)((tn * p.D + zd) * p.H * p.W) * 256u;
        // (spelled as wave-uniform values: the compiler otherwise folds these into the per-thread item masks and wraps every
        // load in a waterfall loop over its scalar offset)
        x_in = __builtin_amdgcn_readfirstlane((int)(h0 >= 0 && h0 + WTH + 2 <= p.H && w0 >= 0 && w0 + WTW + 2 <= p.W)) != 0;
        z_in = __builtin_amdgcn_readfirstlane((int)(h0 + 1 + WTH <= p.H && w0 + 1 + WTW <= p.W)) != 0;
        x_so = xplane + (unsigned)((h0 * p.W + w0) * 256);
        z_so = zplane + (unsigned)(((h0 + 1) * p.W + w0 + 1) * 256);
        if (!x_in && xitem) rowoff = xplane + (unsigned)(min(max(h0 + il, 0), p.H - 1) * p.W * 256 + c16 * 16);
        if (!z_in && zitem) rowoff = h0 + 1 + il < p.H ? zplane + (unsigned)((h0 + 1 + il) * p.W * 256 + c16 * 16) : 0xffffffffu;
    };
    auto load_x = [&](int nn) {
        if (!xitem || (FDN_DBG_BITS(p) & 1)) return;
        const unsigned vo = x_in ? tc0 + (unsigned)(nn * 256) : rowoff + (unsigned)(min(max(tw * WTW + 4 * ig - 1 + nn, 0), p.W - 1) * 256);
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)(x_in ? x_so : 0u));   // scalar operand: no waterfall loop
        raw[nn] = bload(xrs, vo, so);
        if (DEP) fdn_lds_dma16_untracked(xrs_raw, fdn_lds_addr(xscr) + (unsigned)(nn * 4096), vo, so + dxb);     // the same chunk of the second plane -> scratch
    };
    auto load_z = [&](int j) {
        if (!zitem || (FDN_DBG_BITS(p) & 1)) return;
        const int qw = tw * WTW + 4 * ig + j;
        const unsigned vo = z_in ? tc0 + (unsigned)(j * 256) : ((rowoff != 0xffffffffu && qw < p.W) ? rowoff + (unsigned)(qw * 256) : 0xffffffffu);
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((int)(z_in ? z_so : 0u));
        raw[j] = bload(zrs, vo, so);
        if (DEP && has_zb) {               // second plane: chunks 0, 1 in the two spare registers, chunks 2, 3 through the scratch
            const unsigned vb = vo == 0xffffffffu ? vo : vo + dzb;               // (so + dzb could wrap past a "reads zero" offset)
            if (j < 2) raw[4 + j] = bload(zrs, vb, so);
            else fdn_lds_dma16_untracked(zrs_raw, fdn_lds_addr(zscr) + (unsigned)((j - 2) * 3072), vb, so);
        }
    };
    // DEP: combine the two planes of an item before its W transform (x: second plane from the scratch this thread's own LDS-DMA filled)
    auto combine_x = [&]() {
        if (!DEP || !xitem || (FDN_DBG_BITS(p) & 2)) return;
        // hipcc orders a ds_read behind a pending LDS-DMA only across a barrier; this read-back of the wave's own pieces needs the
        // wait spelled out.  The pieces were requested a tile ago (slots 6-11 of the previous iteration): nothing is lost here.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int nn = 0; nn < 6; ++nn) raw[nn] += sx * *(const f32x4*)(xscr + nn * 4096 + lane * 16);
    };
    auto combine_z = [&]() {
        if (!DEP || !zitem || (FDN_DBG_BITS(p) & 2)) return;
        if (has_zb) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // as in combine_x (chunks 2, 3 come back from the scratch)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (has_zb) raw[j] = cA * raw[j] + cB * (j < 2 ? raw[4 + j] : *(const f32x4*)(zscr + (j - 2) * 3072 + lane * 16));
            else raw[j] = cA * raw[j];
        }
    };
    // LDS image of a (line, group): [parity e][ (xp0,xp1) pairs: 64 ch x 2 | xp2: 64 ch ]  (xi = 2 xp + e): a wave reads its three
    // coordinates with one ds_read_b64 + one ds_read_b32
    auto put = [&](char* dst, f32x4 v0, f32x4 v1, f32x4 v2) {
        *(f32x4*)(dst + c16 * 32) = (f32x4){v0.x, v1.x, v0.y, v1.y};
        *(f32x4*)(dst + c16 * 32 + 16) = (f32x4){v0.z, v1.z, v0.w, v1.w};
        *(f32x4*)(dst + 512 + c16 * 16) = v2;
    };
    // Interpolation points 0, +-a, +-b, inf with a = 3/4, b = 3/2 (round 5; rounds 2-4 used Lavin & Gray's a = 1, b = 2): the same even / odd
    // structure, every constant below exact in fp32, transform entries up to 3.4 instead of 8 -- a third of the fp32 error on trained
    // layers (conv64_wino2d_kernel.h has the measurement).  The 1 / N_xi factors of G' (64/81, -128/243, 32/243: not fp32 numbers) are
    // applied ONCE, to the accumulated matrices in the output transform below, not to every dz chunk.
    constexpr float pa = 0.75f, pb = 1.5f, pa2 = 0.5625f, pb2 = 2.25f, pa3 = 0.421875f, pb3 = 3.375f;
    constexpr float pab2 = 1.6875f, pa2b = 0.84375f, ps = 2.8125f, pp = 1.265625f;      // a b^2, a^2 b, a^2 + b^2, a^2 b^2
    auto write_v = [&](int e, char* buf) {
        // B^T of F(4,3)/F(3,4): (a2b2,0,-(a2+b2),0,1,0) (0,-+ab2,-b2,+-a,1,0) (0,-+a2b,-a2,+-b,1,0) (0,a2b2,0,-(a2+b2),0,1)
        if (!xitem || (FDN_DBG_BITS(p) & 2)) return;
        char* dst = buf + (il * WTG + ig) * GROWB + e * 768;
        const f32x4 x0 = raw[0], x1 = raw[1], x2 = raw[2], x3 = raw[3], x4 = raw[4], x5 = raw[5];
        const f32x4 t1 = x4 - pb2 * x2, t2 = pa * x3 - pab2 * x1, t3 = x4 - pa2 * x2, t4 = pb * x3 - pa2b * x1;
        if (e == 0) put(dst, pp * x0 - ps * x2 + x4, t1 - t2, t3 - t4);            // xi = 0, 2, 4  (points 0, -a, -b)
        else put(dst, t1 + t2, t3 + t4, pp * x1 - ps * x3 + x5);                    // xi = 1, 3, 5  (points +a, +b, inf)
    };
    auto write_z = [&](int e, char* buf) {
        // G' of F(3,4) without its 1 / N_xi row factors: (1,0,0,0) (1,+-a,a2,+-a3) (1,+-b,b2,+-b3) (0,0,0,1)
        if (!zitem || (FDN_DBG_BITS(p) & 2)) return;
        char* dst = buf + VBYTES + (il * WTG + ig) * GROWB + e * 768;
        const f32x4 z0 = raw[0], z1 = raw[1], z2 = raw[2], z3 = raw[3];
        const f32x4 e1 = z0 + pa2 * z2, o1 = pa * z1 + pa3 * z3, e2 = z0 + pb2 * z2, o2 = pb * z1 + pb3 * z3;
        if (e == 0) put(dst, z0, e1 - o1, e2 - o2);                                 // xi = 0, 2, 4
        else put(dst, e1 + o1, e2 + o2, z3);                                         // xi = 1, 3, 5
    };

    // ---- prologue: tile 0 -> buffer 0, tile 1 -> registers ----
    locate();
#pragma unroll
    for (int nn = 0; nn < 6; ++nn) load_x(nn);
#pragma unroll
    for (int j = 0; j < 4; ++j) load_z(j);
    combine_x(); combine_z();
#pragma unroll
    for (int e = 0; e < 2; ++e) { write_v(e, smem); write_z(e, smem); }
    advance();
    locate();
#pragma unroll
    for (int nn = 0; nn < 6; ++nn) load_x(nn);
#pragma unroll
    for (int j = 0; j < 4; ++j) load_z(j);
    __syncthreads();

    // operands: lane (li, kh) = channel (mq*32 + li) resp. (nq*32 + li) of group kh, coordinate parity eh
    const int lane_v = kh * GROWB + eh * 768 + (mq * 32 + li) * 8;
    const int lane_z = VBYTES + kh * GROWB + eh * 768 + (nq * 32 + li) * 8;
    const int lane_v1 = kh * GROWB + eh * 768 + 512 + (mq * 32 + li) * 4;
    const int lane_z1 = VBYTES + kh * GROWB + eh * 768 + 512 + (nq * 32 + li) * 4;
    // x operands live in a ring of four halo lines (line L in slot L & 3): a line is read once and serves the up to three (q, b)
    // slots with q + b = L -- 8 + 6 operand fetches per tile instead of 18 + 6
    f32x2 Vp[4] = {{1.f, 1.f}, {1.f, 1.f}, {1.f, 1.f}, {1.f, 1.f}}, Zp[2] = {{1.f, 1.f}, {1.f, 1.f}};
    float Vs[4] = {1.f, 1.f, 1.f, 1.f}, Zs[2] = {1.f, 1.f};
    auto issue_v = [&](const char* buf, int line, int slot) {
        if (FDN_DBG_BITS(p) & 4) return;
        Vp[slot] = *(const f32x2*)(buf + lane_v + line * (WTG * GROWB));
        Vs[slot] = *(const float*)(buf + lane_v1 + line * (WTG * GROWB));
    };
    auto issue_z = [&](const char* buf, int line, int slot) {
        if (FDN_DBG_BITS(p) & 4) return;
        Zp[slot] = *(const f32x2*)(buf + lane_z + line * (WTG * GROWB));
        Zs[slot] = *(const float*)(buf + lane_z1 + line * (WTG * GROWB));
    };
    int bcur = 0;
    issue_z(smem, 0, 0);
    issue_v(smem, 0, 0);
    // The tile loop exists three times, one copy per ROLE of the wave (0: transforms x, waves 0-3; 1: transforms dz, waves 4-6; 2: wave 7,
    // MFMAs only), chosen once by a scalar branch.  With ONE copy and the role tested at every stage, hipcc's wait-count bookkeeping merges
    // paths no wave can take ("skipped the transform of slot 0 but issues the loads of slot 6") and, depending on nothing more than the
    // register allocation of the transform code, may decide a raw-row register is still in flight when its slot is reloaded: round 5's
    // change of constants in write_v / write_z produced s_waitcnt vmcnt(0) after EVERY raw-row load, 0.60 -> 0.75 ms at (8,48^3).
    auto tile_loop = [&](auto role) {
    constexpr int ROLE = decltype(role)::value;
#pragma unroll 1
    for (int k = 0; k < nk; ++k) {
        const int bnxt = bcur == 2 ? 0 : bcur + 1;
        const char* cur = smem + bcur * WBUFB;
        char* nxt = smem + bnxt * WBUFB;
        // 18 slots: slot s = (dz line q = s/3, height tap b = s%3) -> x halo line q + b; 3 MFMAs per wave and slot
#pragma unroll
        for (int s = 0; s < 18; ++s) {
            const int q = s / 3, b = s % 3;
            if (s == 12 && !(FDN_DBG_BITS(p) & 64)) __syncthreads();       // tile k+1 is complete in `nxt`; every wave is done with tile k-1's buffer  (test build, bit 64: timing without it)
            // read-ahead: operands of the next slot (the first slot of tile k+1 at the end)
            if (s + 1 < 18) {
                const int ln = (s + 1) / 3 + (s + 1) % 3;                      // x halo line of the next slot: new iff b = 2, or q = 0
                if ((s + 1) % 3 == 2 || s + 1 < 3) issue_v(cur, ln, ln & 3);
                if ((s + 1) % 3 == 0) issue_z(cur, (s + 1) / 3, ((s + 1) / 3) & 1);
            } else {
                issue_v(nxt, 0, 0);
                issue_z(nxt, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            // pipeline stages, pinned to slots: transform + write tile k+1, then load the raw rows of tile k+2
            // (the raw rows of tile k+2 are requested as soon as tile k+1's have been consumed: they are read again a whole tile
            // later, in slots 0-3 of the next iteration -- requested after the barrier, the dz rows came back too late)
            if (ROLE == 0 && s == 0) combine_x();
            if (ROLE == 1 && s == 2) combine_z();
            if (ROLE == 0 && s < 2) write_v(s, nxt);
            if (ROLE == 1 && s >= 2 && s < 4) write_z(s - 2, nxt);
            if (ROLE == 0 && s == kLocSlot && !(FDN_DBG_BITS(p) & 32)) { advance(); locate(); }
            if (ROLE == 1 && s == kLocSlotZ && !(FDN_DBG_BITS(p) & 32)) { advance(); locate(); }      // (wave 7 loads nothing: no walk)
            if (ROLE == 0 && s >= kXSlot && s < kXSlot + 6) load_x(s - kXSlot);
            if (ROLE == 1 && s >= kZSlot && s < kZSlot + 4) load_z(s - kZSlot);
            acc[b][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vp[(q + b) & 3].x, Zp[q & 1].x, acc[b][0], 0, 0, 0);
            acc[b][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vp[(q + b) & 3].y, Zp[q & 1].y, acc[b][1], 0, 0, 0);
            acc[b][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vs[(q + b) & 3], Zs[q & 1], acc[b][2], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        bcur = bnxt;
    }
    };
    if (xitem) tile_loop(std::integral_constant<int, 0>{});
    else if (zitem) tile_loop(std::integral_constant<int, 1>{});
    else tile_loop(std::integral_constant<int, 2>{});

    // ---- output transform inside the workgroup, then ONE partial dW[a][b][t] per workgroup (half the partial traffic of writing
    // the 18 Winograd-domain matrices).  dW[t] = sum_xi A'^T[t][xi] c_xi M_xi with A'^T = (1,1,1,1,1,0) (0,a,-a,b,-b,0) (0,a2,a2,b2,b2,1)
    // and c = 1 / N_xi = (64/81, -128/243, -128/243, 32/243, 32/243, 1), the row factors of G' (see write_z);
    // a wave holds one parity of xi, so it forms its share of the three taps in place and the odd-parity wave of each quadrant
    // hands its share to the even one through LDS (the tile buffers are free now), in two rounds of <= 5 of the 9 (b,t) tiles. ----
    // (lane-level indices re-derived from mbcnt and the scalar wave id: held in registers through the three tile loops they cost a spill)
    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int li_e = lane_e & 31, kh_e = lane_e >> 5;
    const int mq_e = wave_s & 1, nq_e = (wave_s >> 1) & 1, eh_e = wave_s >> 2;
    {
        constexpr float c0 = 64.f / 81, ca = -128.f / 243, cb = 32.f / 243;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const f32x16 m0 = acc[b][0], m1 = acc[b][1], m2 = acc[b][2];       // xi = eh, 2 + eh, 4 + eh
            if (eh_e == 0) {           // xi 0, 2, 4: points 0, -a, -b
                acc[b][0] = c0 * m0 + ca * m1 + cb * m2;
                acc[b][1] = (-pa * ca) * m1 + (-pb * cb) * m2;
                acc[b][2] = (pa2 * ca) * m1 + (pb2 * cb) * m2;
            } else {                 // xi 1, 3, 5: points +a, +b, inf
                acc[b][0] = ca * m0 + cb * m1;
                acc[b][1] = (pa * ca) * m0 + (pb * cb) * m1;
                acc[b][2] = (pa2 * ca) * m0 + (pb2 * cb) * m1 + m2;
            }
        }
    }
    float* xch = (float*)smem + (size_t)(wave_s & 3) * (5 * 16 * 64) + lane_e;          // per quadrant: [tile 5][r 16][lane 64]
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        __syncthreads();                                    // tile buffers / the previous round's exchange are no longer read
        if (eh_e == 1) {
#pragma unroll
            for (int k = 0; k < 9; ++k)
                if ((k < 5) == (round == 0))
#pragma unroll
                    for (int r = 0; r < 16; ++r) xch[((k - 5 * round) * 16 + r) * 64] = acc[k / 3][k % 3][r];
        }
        __syncthreads();
        if (eh_e == 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k)
                if ((k < 5) == (round == 0))
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[k / 3][k % 3][r] += xch[((k - 5 * round) * 16 + r) * 64];
        }
    }
    if (eh_e == 0) {
        float* out = p.partial + ((size_t)split * (NP * 9) + a * 9) * 4096;     // DEP: [split][xd][b*3+t], mixed into the depth taps by the reduction
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ci = mq_e * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh_e;
                out[(size_t)k * 4096 + ci * 64 + nq_e * 32 + li_e] = acc[k / 3][k % 3][r];
            }
    }
}

// workgroup id -> place q in the XCD-aware order of G = NP * S * layers (split, tap) pairs (see the comment at the top of the body)
__device__ __forceinline__ int wgrad64_place(int block, int G, bool plain) {
    const int qx = G >> 3, rx = G & 7, xcd = block & 7, j = block >> 3;
    return plain ? block : xcd * qx + min(xcd, rx) + j;
}

template <bool DEP>
__global__ __launch_bounds__(512, 1) void wgrad64_wino_kernel(WgWinoArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    wgrad64_wino_body<DEP>(p, wgrad64_place((int)blockIdx.x, (DEP ? 4 : 3) * p.S, (FDN_DBG_BITS(p) & 8) != 0), smem);
}

// Several layers of the SAME grid in one launch (fdn_conv3d_wgrad_batch): the (split, coordinate) pairs of all layers form one list,
// placed on the XCDs like a single layer's (the pairs of a layer stay contiguous: neighbours share that layer's rows in L2); a
// layer gets S = 64 / layers splits, so the launch still fills the chip once, but a workgroup walks layers-times more tiles of ITS layer
// between the prologue and the output transform, and the partial sums it writes (and the reduction reads) shrink by the same factor.
// At the cfg2 low-res grid (8 x 24^3: 4.6 tiles per workgroup when a layer has the chip to itself) that is where the time goes.
constexpr int kWgBatchMax = 32;
struct WgWinoBatch {
    WgWinoArgs a;                        // x / dz / partial unused; everything else common to the layers
    const float* x[kWgBatchMax];
    const float* dz[kWgBatchMax];
    int nl;
};
struct WgDwTable { float* dw[kWgBatchMax]; };

template <bool DEP>
__global__ __launch_bounds__(512, 1) void wgrad64_wino_batch_kernel(WgWinoBatch b) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NP = DEP ? 4 : 3;
    const int per = NP * b.a.S;
    const int qg = wgrad64_place((int)blockIdx.x, per * b.nl, (FDN_DBG_BITS(b.a) & 8) != 0);
    const int layer = qg / per;
    WgWinoArgs p = b.a;
    p.x = b.x[layer]; p.dz = b.dz[layer];
    p.partial = b.a.partial + (size_t)layer * b.a.S * (NP * 9) * 4096;
    wgrad64_wino_body<DEP>(p, qg - layer * per, smem);
}

// DEP: (D / 2) depth units and 4 workgroups per split
bool wgrad64_wino_dep_ok(int D) { return D >= 2 && (D & 1) == 0; }
int wgrad64_wino_splits(int N, int D, int H, int W, bool dep) {
    const long long ntiles = (long long)N * (dep ? D / 2 : D) * ((H + WTH - 1) / WTH) * ((W + WTW - 1) / WTW);
    long long S = dep ? 64 : 85;       // 4 * 64 = 256 / 3 * 85 = 255 workgroups: one (8 waves, 126-150 KB of LDS) per CU
    if (ntiles < S) S = ntiles > 0 ? ntiles : 1;
    return (int)S;
}

// dw[a][k] = sum_s sum_xd G^T[a][xd] partial[s][xd][k]  (k over the 9 (b,t) taps x 64 x 64), G^T = (1,1/2,1/2,0) (0,1/2,-1/2,0) (0,1/2,1/2,1).
// 64 float4 columns x 8 eighths of S per block of 512 threads, combined through LDS in a fixed order (round 6: four quarters in 256 threads
// left every thread a chain of S / 4 dependent-issue iterations on 144 blocks: 11.7 us for 37 MB).
constexpr int kRedParts = 8;
__global__ __launch_bounds__(64 * kRedParts) void wgrad64_reduce_dep_kernel(const float* __restrict__ partial_base, WgDwTable dws, int S) {
    __shared__ f32x4 red[kRedParts - 1][3][64];
    const float* partial = partial_base + (size_t)blockIdx.y * S * 36 * 4096;       // blockIdx.y = layer of a batched launch
    float* dw = dws.dw[blockIdx.y];
    const int col = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int e4 = blockIdx.x * 64 + col;                       // 9*1024 float4 columns of one (b,t) block set
    const f32x4* p = (const f32x4*)partial + e4;
    const int s0q = (S * part) / kRedParts, s1q = (S * (part + 1)) / kRedParts;
    f32x4 m0 = {0.f, 0.f, 0.f, 0.f}, m1 = m0, m2 = m0, m3 = m0;
    for (int s = s0q; s < s1q; ++s) {
        const f32x4* ps = p + (size_t)s * (36 * 1024);
        m0 += ps[0]; m1 += ps[9 * 1024]; m2 += ps[18 * 1024]; m3 += ps[27 * 1024];
    }
    const f32x4 t0 = m0 + 0.5f * (m1 + m2), t1 = 0.5f * (m1 - m2), t2 = 0.5f * (m1 + m2) + m3;
    if (part) { red[part - 1][0][col] = t0; red[part - 1][1][col] = t1; red[part - 1][2][col] = t2; }
    __syncthreads();
    if (part == 0) {
        f32x4 r0 = t0, r1 = t1, r2 = t2;
#pragma unroll
        for (int q = 0; q < kRedParts - 1; ++q) { r0 += red[q][0][col]; r1 += red[q][1][col]; r2 += red[q][2][col]; }
        ((f32x4*)dw)[e4] = r0;
        ((f32x4*)dw)[9 * 1024 + e4] = r1;
        ((f32x4*)dw)[18 * 1024 + e4] = r2;
    }
}

}  // namespace

#ifdef FDN_TEST_HOOKS
extern "C" int fdn_debug_set_wgrad64_wino_dbg(int bits) { fdn_wgrad64_wino_dbg = bits; return FDN_OK; }
#endif

FDN_HOOK_VAR(int, fdn_wgrad64_wino_nodep, 0);          // test build: 1 = never the depth transform (the round-2 kernel)

size_t fdn_wgrad64_wino_workspace_bytes(int N, int D, int H, int W) {
    const size_t a = (size_t)wgrad64_wino_splits(N, D, H, W, false) * 27, b = wgrad64_wino_dep_ok(D) ? (size_t)wgrad64_wino_splits(N, D, H, W, true) * 36 : 0;
    return (a > b ? a : b) * 4096 * sizeof(float);
}

int fdn_wgrad64_wino_launch(const float* x, const float* dz, float* dw, void* ws, size_t ws_bytes, int N, int D, int H,
                            int W, hipStream_t s, int algo) {
    // F(3,2) along D on top of F(3,4) along W whenever D is even (FDN_ALGO_WINO_W keeps the W-only kernel selectable)
    const bool dep = wgrad64_wino_dep_ok(D) && algo != FDN_ALGO_WINO_W && !fdn_wgrad64_wino_nodep;
    WgWinoArgs a;
    a.x = x; a.dz = dz; a.partial = (float*)ws;
    a.N = N; a.D = D; a.H = H; a.W = W;
    a.DT = dep ? D / 2 : D;
    a.nth = (H + WTH - 1) / WTH; a.ntw = (W + WTW - 1) / WTW;
    a.ntiles = N * a.DT * a.nth * a.ntw;
    a.S = wgrad64_wino_splits(N, D, H, W, dep);
    FDN_REQUIRE((long long)N * D * H * W * 256 < (1ll << 32), "wgrad64: x of %dx%dx%dx%dx64 floats exceeds the 32-bit buffer addressing", N, D, H, W);
    FDN_REQUIRE(ws_bytes >= (size_t)a.S * (dep ? 36 : 27) * 4096 * sizeof(float), "wgrad64 (winograd): workspace too small");
    a.bytes = (unsigned)((long long)N * D * H * W * 256);
    a.dbg = fdn_wgrad64_wino_dbg;
    if (dep) {
        const size_t lds = (size_t)3 * WBUFB + XSCRATCH + ZSCRATCH;
        if (int rc = fdn_func_max_lds((const void*)wgrad64_wino_kernel<true>, (int)lds, "wgrad64_wino")) return rc;
        hipLaunchKernelGGL(wgrad64_wino_kernel<true>, dim3(4 * a.S), dim3(512), lds, s, a);
        FDN_CHECK_LAUNCH("wgrad64_wino_kernel");
        WgDwTable t;
        t.dw[0] = dw;
        hipLaunchKernelGGL(wgrad64_reduce_dep_kernel, dim3(9 * 1024 / 64), dim3(64 * kRedParts), 0, s, (const float*)ws, t, a.S);
        FDN_CHECK_LAUNCH("wgrad64_reduce_dep_kernel");
        return FDN_OK;
    }
    const size_t lds = (size_t)3 * WBUFB;
    if (int rc = fdn_func_max_lds((const void*)wgrad64_wino_kernel<false>, (int)lds, "wgrad64_wino")) return rc;
    hipLaunchKernelGGL(wgrad64_wino_kernel<false>, dim3(3 * a.S), dim3(512), lds, s, a);
    FDN_CHECK_LAUNCH("wgrad64_wino_kernel");
    return fdn_wgrad64_reduce_launch((const float*)ws, dw, a.S, s);
}

// ---- several layers of one grid in ONE launch (+ one reduction launch) ----
// Applicable when the depth-transformed kernel is (D even, FDN_ALGO_AUTO) and the layers' partial sums fit 32-bit tile counts;
// fdn_conv3d_wgrad_batch (api.hip) falls back to per-layer launches otherwise.
bool fdn_wgrad64_wino_batch_ok(int n_layers, int D, int algo) {
    return n_layers >= 2 && n_layers <= kWgBatchMax && wgrad64_wino_dep_ok(D) && algo != FDN_ALGO_WINO_W && algo != FDN_ALGO_DIRECT && !fdn_wgrad64_wino_nodep;
}
static int wgrad64_wino_batch_splits(int n_layers, int N, int D, int H, int W) {
    const long long ntiles = (long long)N * (D / 2) * ((H + WTH - 1) / WTH) * ((W + WTW - 1) / WTW);
    long long S = 64 / n_layers;         // 4 S n_layers <= 256 workgroups: one per CU, the chip filled once
    if (S < 1) S = 1;
    if (ntiles < S) S = ntiles > 0 ? ntiles : 1;
    return (int)S;
}
size_t fdn_wgrad64_wino_batch_workspace_bytes(int n_layers, int N, int D, int H, int W) {
    return (size_t)n_layers * wgrad64_wino_batch_splits(n_layers, N, D, H, W) * 36 * 4096 * sizeof(float);
}
int fdn_wgrad64_wino_batch_launch(const float* const* x, const float* const* dz, float* const* dw, int n_layers, void* ws, size_t ws_bytes,
                                  int N, int D, int H, int W, hipStream_t s) {
    FDN_REQUIRE(fdn_wgrad64_wino_batch_ok(n_layers, D, FDN_ALGO_AUTO), "wgrad64 (batched): %d layers at depth %d are not batchable", n_layers, D);
    FDN_REQUIRE((long long)N * D * H * W * 256 < (1ll << 32), "wgrad64: x of %dx%dx%dx%dx64 floats exceeds the 32-bit buffer addressing", N, D, H, W);
    WgWinoBatch b;
    WgDwTable t;
    b.nl = n_layers;
    for (int i = 0; i < n_layers; ++i) {
        FDN_REQUIRE(x[i] && dz[i] && dw[i], "wgrad64 (batched): NULL pointer for layer %d", i);
        b.x[i] = x[i]; b.dz[i] = dz[i]; t.dw[i] = dw[i];
    }
    WgWinoArgs& a = b.a;
    a.x = nullptr; a.dz = nullptr; a.partial = (float*)ws;
    a.N = N; a.D = D; a.H = H; a.W = W;
    a.DT = D / 2;
    a.nth = (H + WTH - 1) / WTH; a.ntw = (W + WTW - 1) / WTW;
    a.ntiles = N * a.DT * a.nth * a.ntw;
    a.S = wgrad64_wino_batch_splits(n_layers, N, D, H, W);
    FDN_REQUIRE(ws_bytes >= fdn_wgrad64_wino_batch_workspace_bytes(n_layers, N, D, H, W), "wgrad64 (batched): workspace too small");
    a.bytes = (unsigned)((long long)N * D * H * W * 256);
    a.dbg = fdn_wgrad64_wino_dbg;
    const size_t lds = (size_t)3 * WBUFB + XSCRATCH + ZSCRATCH;
    if (int rc = fdn_func_max_lds((const void*)wgrad64_wino_batch_kernel<true>, (int)lds, "wgrad64_wino_batch")) return rc;
    hipLaunchKernelGGL(wgrad64_wino_batch_kernel<true>, dim3(4 * a.S * n_layers), dim3(512), lds, s, b);
    FDN_CHECK_LAUNCH("wgrad64_wino_batch_kernel");
    hipLaunchKernelGGL(wgrad64_reduce_dep_kernel, dim3(9 * 1024 / 64, n_layers), dim3(64 * kRedParts), 0, s, (const float*)ws, t, a.S);
    FDN_CHECK_LAUNCH("wgrad64_reduce_dep_kernel");
    return FDN_OK;
}

#ifdef FDN_TEST_HOOKS
extern "C" int fdn_debug_set_wgrad64_wino_nodep(int on) { fdn_wgrad64_wino_nodep = on; return FDN_OK; }
#endif

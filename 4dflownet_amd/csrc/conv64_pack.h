// Per-element weight re-layout shared by the single-layer and the batched pack kernels (conv64_mfma.hip, conv64_wino.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Direct stream: Keras (27,64,64)[tap][cin][cout] -> [half][tap][g][kh][row j][s]
//   row j of a 32-row tile is MFMA row i = j & 31, which lands in accumulator register r = (i&3) + 4(i>>3) of lane half
//   kh' = (i>>2)&1; the stream stores output channel c(j) = (j & 32) + 16 kh' + r there, so that a lane's 16 registers are
//   16 consecutive channels
//   fwd  : cin = 32*half + 8g + 4kh + s, same tap, cout = c(j)
//   dgrad: contraction runs over cout of the layer, taps flipped: stream[..][j][s] = w[26-tap][ci = c(j)][co = 32*half+8g+4kh+s]
__device__ __forceinline__ void fdn_pack_direct_one(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wd, int idx) {
    const int s = idx & 3;
    const int j = (idx >> 2) & 63;
    const int kh = (idx >> 8) & 1;
    const int g = (idx >> 9) & 3;
    const int rest = idx >> 11;          // half*27 + tap
    const int half = rest / 27;
    const int tap = rest - half * 27;
    const int k = half * 32 + g * 8 + kh * 4 + s;
    const int cj = (j & 32) + 16 * ((j >> 2) & 1) + (j & 3) + 4 * ((j & 31) >> 3);
    if (wf) wf[idx] = w[(tap * 64 + k) * 64 + cj];
    if (wd) wd[idx] = w[((26 - tap) * 64 + cj) * 64 + k];
}

// Winograd-domain stream U = G g along the W taps, G of F(4,3):
//   (1/4,0,0) (-1/6,-1/6,-1/6) (-1/6,1/6,-1/6) (1/24,1/12,1/6) (1/24,-1/12,1/6) (0,0,1)
// layout [half][vt = (a*3+b)*6 + xi][k-group][kh][row j][s], row permutation c(j) as above.
//   fwd  : cin = 32*half + 8g + 4kh + s, cout = c(j):  U = sum_t G[xi][t] w[a][b][t][cin][cout]
//   dgrad: contraction over the layer's cout, taps flipped:  U = sum_t G[xi][t] w[2-a][2-b][2-t][ci = c(j)][co = 32*half+8g+4kh+s]
__device__ __forceinline__ void fdn_pack_wino_one(const float* __restrict__ w, float* __restrict__ uf, float* __restrict__ ud, int idx) {
    const int s = idx & 3;
    const int j = (idx >> 2) & 63;
    const int kh = (idx >> 8) & 1;
    const int g = (idx >> 9) & 3;
    const int rest = idx >> 11;          // half*54 + vt
    const int half = rest / 54;
    const int vt = rest - half * 54;
    const int tap9 = vt / 6;
    const int xi = vt - tap9 * 6;
    const int k = half * 32 + g * 8 + kh * 4 + s;
    const int cj = (j & 32) + 16 * ((j >> 2) & 1) + (j & 3) + 4 * ((j & 31) >> 3);
    const float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                           {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
    if (uf) {
        float v = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t) v = __builtin_fmaf(G[xi][t], w[((tap9 * 3 + t) * 64 + k) * 64 + cj], v);   // explicit fma: every caller rounds alike
        uf[idx] = v;
    }
    if (ud) {
        float v = 0.f;
#pragma unroll
        for (int t = 0; t < 3; ++t) v = __builtin_fmaf(G[xi][t], w[((26 - (tap9 * 3 + t)) * 64 + cj) * 64 + k], v);
        ud[idx] = v;
    }
}

// 2-D Winograd stream (conv64_wino2d.hip): U = Gh g Gw^T over the (kh, kw) taps, F(2,3) along H (Gh = (1,0,0) (1/2,1/2,1/2)
// (1/2,-1/2,1/2) (0,0,1)) x F(4,3) along W (G above), per depth tap kd.  The kernel forms the F(2,3) input coordinate 2 as
// x1 - x2 = -(B^T x)_2, so U[xh = 2] is stored NEGATED.
// layout [nb = cout/16][xh][kd][xw][g = cin/16][q][i][s]: a 1-KB unit = the row operand of v_mfma_f32_16x16x4_f32 for 4 k-steps:
// lane (i = lane & 15, q = lane >> 4), k-step s  <->  cout 16 nb + i, cin 16 g + 4 q + s.
//   fwd  : U = sum Gh[xh][kh] Gw[xw][kw] w[kd][kh][kw][cin][cout]
//   dgrad: contraction over the layer's cout, taps flipped: U = sum Gh Gw w[2-kd][2-kh][2-kw][ci = 16 nb + i][co = 16 g + 4 q + s]
__device__ __forceinline__ void fdn_pack_wino2d_one(const float* __restrict__ w, float* __restrict__ uf, float* __restrict__ ud, int idx) {
    const int s = idx & 3;
    const int i = (idx >> 2) & 15;
    const int q = (idx >> 6) & 3;
    const int g = (idx >> 8) & 3;
    int rest = idx >> 10;                // ((nb*4 + xh)*3 + kd)*6 + xw
    const int xw = rest % 6; rest /= 6;
    const int kd = rest % 3; rest /= 3;
    const int xh = rest & 3;
    const int nb = rest >> 2;
    const int k = 16 * g + 4 * q + s;
    const int cj = 16 * nb + i;
    const float G[6][3] = {{0.25f, 0.f, 0.f}, {-1.f / 6, -1.f / 6, -1.f / 6}, {-1.f / 6, 1.f / 6, -1.f / 6},
                           {1.f / 24, 1.f / 12, 1.f / 6}, {1.f / 24, -1.f / 12, 1.f / 6}, {0.f, 0.f, 1.f}};
    const float H[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {-0.5f, 0.5f, -0.5f}, {0.f, 0.f, 1.f}};     // row 2 = -(1/2,-1/2,1/2)
    if (uf) {
        float v = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            float r = 0.f;
#pragma unroll
            for (int t = 0; t < 3; ++t) r = __builtin_fmaf(G[xw][t], w[(((kd * 3 + kh) * 3 + t) * 64 + k) * 64 + cj], r);
            v = __builtin_fmaf(H[xh][kh], r, v);
        }
        uf[idx] = v;
    }
    if (ud) {
        float v = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            float r = 0.f;
#pragma unroll
            for (int t = 0; t < 3; ++t) r = __builtin_fmaf(G[xw][t], w[((26 - ((kd * 3 + kh) * 3 + t)) * 64 + cj) * 64 + k], r);
            v = __builtin_fmaf(H[xh][kh], r, v);
        }
        ud[idx] = v;
    }
}

// 2-D Winograd stream with F(4,3) along H as well (conv64_wino2d.hip, HM = 4): U = G g G^T over the (kh, kw) taps, per depth tap kd, G of
// F(4,3) on the points 0, +-3/4, +-3/2, inf (conv64_wino2d_kernel.h; G[k][j] = p_k^j / prod_{l != k} (p_k - p_l)):
//   (64/81, 0, 0) (-128/243, -+32/81, -8/27) (32/243, +-16/81, 8/27) (0, 0, 1)
// The entries are not fp32 numbers, so the stream is formed in double precision and rounded ONCE: U's rounding is the one error of the
// algorithm that is the same for every voxel (a fixed perturbation of the layer's kernel), and it should be half an ulp, not the three
// or four of an fp32 fma chain over rounded constants.
// layout [nb = cout/16][xh 0..5][kd][xw 0..5][g = cin/16][q][i][s], the 1-KB unit as in fdn_pack_wino2d_one; 108 * 64 * 64 floats.
// (fdn_pack_wino44_column below writes it.)

// u = hi + mid + lo exactly, each piece the round-to-nearest-even bf16 of what the pieces before it leave.  Spelled on the bit pattern
// (finite weights): every pack path rounds alike whatever conversion the compiler would pick for a (double ->) float -> bf16 chain.
__device__ __forceinline__ uint16_t fdn_bf16_rne_bits(float x) {
    const unsigned b = __builtin_bit_cast(unsigned, x);
    return (uint16_t)((b + 0x7fffu + ((b >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ void fdn_split_bf16x3(float u, uint16_t& hi, uint16_t& mid, uint16_t& lo) {
    hi = fdn_bf16_rne_bits(u);
    const float r1 = u - __builtin_bit_cast(float, (unsigned)hi << 16);
    mid = fdn_bf16_rne_bits(r1);
    lo = fdn_bf16_rne_bits(r1 - __builtin_bit_cast(float, (unsigned)mid << 16));
}

// Both F(4,3) x F(4,3) streams for ONE (kd, cin k, cout cj) of one direction: the 9 (kh, kw) weights are loaded once and all 36 (xh, xw)
// coordinates formed from them -- the per-element form above asks for 9 scattered 4-B loads per output, and the per-step re-pack
// (30 layers x 2 directions after every optimizer step) was bound by exactly those load instructions (0.175 ms per cfg2 step).
// Same expression, same order of the 9 terms as fdn_pack_wino44_one / fdn_pack_wino44s_one: bit-identical streams.
// u44 / u44s: the layer-and-direction's fp32 stream / bf16 x 3 stream, or nullptr.  dgrad: taps flipped, channels transposed.
// (Instantiated in ONE kernel, pack_conv64_wino44_kernel of conv64_mfma.hip, which both the single-layer and the batched pack launch:
// inlined into two kernels hipcc contracted the 9-term sums differently -- one fp32 value in 1.3 million came out an ulp apart.)
__device__ __forceinline__ void fdn_pack_wino44_column(const float* __restrict__ w, float* __restrict__ u44, uint16_t* __restrict__ u44s,
                                                        int kd, int k, int cj, bool dgrad) {
    const double G[6][3] = {{64.0 / 81, 0.0, 0.0}, {-128.0 / 243, -32.0 / 81, -8.0 / 27}, {-128.0 / 243, 32.0 / 81, -8.0 / 27},
                            {32.0 / 243, 16.0 / 81, 8.0 / 27}, {32.0 / 243, -16.0 / 81, 8.0 / 27}, {0.0, 0.0, 1.0}};
    double wv[3][3];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int tap = (kd * 3 + kh) * 3 + t;
            wv[kh][t] = (double)(dgrad ? w[((26 - tap) * 64 + cj) * 64 + k] : w[(tap * 64 + k) * 64 + cj]);
        }
    const int nb = cj >> 4, i = cj & 15;
    // fp32 stream: [nb][xh][kd][xw][g = k/16][q][i][s], k = 16 g + 4 q + s;  bf16 x 3: [nb][xh][pass = k/32][kd][xw][piece][lane = 16 q' + i][j], k = 32 pass + 8 q' + j
    const int o44 = (k >> 2) * 64 + i * 4 + (k & 3);
    const int pass = k >> 5, o44s = ((k >> 3) & 3) * 128 + i * 8 + (k & 7);
#pragma unroll
    for (int xh = 0; xh < 6; ++xh)
#pragma unroll
        for (int xw = 0; xw < 6; ++xw) {
            double v = 0.0;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int t = 0; t < 3; ++t) v += G[xh][kh] * G[xw][t] * wv[kh][t];
            const float u = (float)v;
            if (u44) u44[(size_t)(((nb * 6 + xh) * 3 + kd) * 6 + xw) * 1024 + o44] = u;
            if (u44s) {
                uint16_t* d = u44s + (size_t)((((nb * 6 + xh) * 2 + pass) * 3 + kd) * 6 + xw) * 1536 + o44s;
                fdn_split_bf16x3(u, d[0], d[512], d[1024]);
            }
        }
}

// The F(4,3) x F(4,3) stream once more, for FDN_ALGO_WINO_BF16X3 (conv64_wino2d_kernel.h, SPLIT): the same U (double precision, rounded once
// to fp32) split EXACTLY into three bf16 pieces u = hi + mid + lo, laid out as the row operand of v_mfma_f32_16x16x32_bf16:
// [nb = cout/16][xh 0..5][pass = cin/32][kd][xw 0..5][piece][lane][8]: lane (i = lane & 15, q = lane >> 4), element j <-> cout 16 nb + i,
// cin 32 pass + 8 q + j.  A wave's (stage, pass, kd, xw) step reads 3 KB: one 16-B load per lane and piece.
// The stream is 3 * 108 * 64 * 64 uint16_t = 162 * 64 * 64 float-sized slots (written by fdn_pack_wino44_column above).

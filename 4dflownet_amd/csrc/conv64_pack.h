// Per-element weight re-layout shared by the single-layer and the batched pack kernels (conv64_mfma.hip, conv64_wino.hip).
#pragma once
#include <hip/hip_runtime.h>

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

// pure-MFMA ceiling on this box: bf16 32x32x16, NACC independent accumulators per wave, 1-2 waves per SIMD.
// Operands are random-ish bit patterns (data-dependent power: zeros would flatter the clock).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned seed, int zeros) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 av, bv;
    unsigned h = seed + threadIdx.x * 2654435761u;
    for (int r = 0; r < 8; ++r) {
        h = h * 1664525u + 1013904223u; av[r] = (__bf16)(((int)(h >> 8) % 2001 - 1000) * 1e-3f);
        h = h * 1664525u + 1013904223u; bv[r] = (__bf16)(((int)(h >> 8) % 2001 - 1000) * 1e-3f);
        if (zeros) { av[r] = (__bf16)0.f; bv[r] = (__bf16)0.f; }      // all-zero operands: the flattering case (no datapath toggling)
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(float* d, int wg_per_cu, int iters, int zeros) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu;
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, d, iters, 12345u, zeros);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop = (double)grid * 4 * iters * 8 * NACC * 2.0 * 32 * 32 * 16;
    printf("%-7s operands, %d independent accumulators, %d waves/SIMD, %6d iters: %8.3f ms  %7.1f TFLOP/s\n",
           zeros ? "zero" : "random", NACC, wg_per_cu, iters, best, flop / best * 1e-9);
}
int main() {
    float* d; hipMalloc(&d, 1 << 24);
    // >= 4 independent accumulators x >= 2 waves per SIMD, short and long runs (power management reacts after
    // ~1 ms), random vs all-zero operands
    for (int zeros = 0; zeros <= 1; ++zeros)
        for (int wg_per_cu = 1; wg_per_cu <= 2; ++wg_per_cu)
            for (int iters : {2000, 50000}) {
                run<4>(d, wg_per_cu, iters, zeros);
                run<8>(d, wg_per_cu, iters / 2, zeros);
            }
    return 0;
}

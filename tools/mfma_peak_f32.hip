// pure-MFMA ceiling on this box for the fp32 path: v_mfma_f32_32x32x2_f32, NACC independent accumulators per wave, 1/2/4 waves per
// SIMD, random vs all-zero operands (data-dependent power: zeros flatter the clock), short and long launches.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, unsigned seed, int zeros) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float av[4], bv[4];
    unsigned h = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    for (int r = 0; r < 4; ++r) {
        h = h * 1664525u + 1013904223u; av[r] = ((int)(h >> 8) % 2001 - 1000) * 1e-3f;
        h = h * 1664525u + 1013904223u; bv[r] = ((int)(h >> 8) % 2001 - 1000) * 1e-3f;
        if (zeros) { av[r] = 0.f; bv[r] = 0.f; }
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[(u + i) & 3], acc[i], 0, 0, 0);
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
    const double flop = (double)grid * 4 * iters * 4 * NACC * 2.0 * 32 * 32 * 2;
    printf("%-7s operands, %d independent accumulators, %d waves/SIMD, %6d iters: %8.3f ms  %7.1f TFLOP/s (%.3f of 157.3)\n",
           zeros ? "zero" : "random", NACC, wg_per_cu, iters, best, flop / best * 1e-9, flop / best * 1e-9 / 157.3);
}
int main() {
    float* d; hipMalloc(&d, 1 << 24);
    for (int zeros = 0; zeros <= 1; ++zeros)
        for (int wg_per_cu : {1, 2, 4})
            for (int iters : {1500, 40000}) {
                run<6>(d, wg_per_cu, iters / wg_per_cu, zeros);
                run<9>(d, wg_per_cu, iters * 2 / 3 / wg_per_cu, zeros);
            }
    return 0;
}

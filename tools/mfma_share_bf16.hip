// bf16 counterpart of mfma_share.hip: does VALU / LDS work overlap with a v_mfma_f32_32x32x16_bf16 stream?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// MODE 0: MFMA only.  MODE 1: k v_fma after every MFMA, same wave.  MODE 2: k ds_read_b128 per MFMA (same wave, waited once per 16).
// MODE 3: waves 4-7 (SIMD partners) run a VALU loop of siters*64 fma while waves 0-3 run the MFMAs.
template <int MODE, int K>
__global__ __launch_bounds__(512) void k(float* out, long long* tm, int miters, int siters, float a, float b) {
    __shared__ f32x4 lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 512) lds[i] = (f32x4){a, b, a, b};
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    const long long t0 = wall_clock64();
    float s = 0.f;
    if (wave < 4) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        bf16x8 av, bv;
        for (int i = 0; i < 8; ++i) { av[i] = (__bf16)(a + i); bv[i] = (__bf16)b; }
        float sv[8]; for (int i = 0; i < 8; ++i) sv[i] = a * i;
        f32x4 r4[8];
        const unsigned laddr = (threadIdx.x & 63) * 16;
        const f32x4* gp = (const f32x4*)out + (threadIdx.x & 63);
        f32x4* gq = (f32x4*)out + 4096 + blockIdx.x * 512 + threadIdx.x;
        for (int q = 0; q < 8; ++q) r4[q] = (f32x4){a, b, a, b};
        for (int it = 0; it < miters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[u & 3], 0, 0, 0);
                if (MODE == 1) {
#pragma unroll
                    for (int q = 0; q < K; ++q) { sv[q & 7] = __builtin_fmaf(sv[q & 7], b, a); asm volatile("" : "+v"(sv[q & 7])); }
                }
                if (MODE == 3 && K >= 1 && K < 100) asm volatile("s_nop %0" :: "n"(K - 1));
                if (MODE == 2 && u < 8) {
#pragma unroll
                    for (int q = 0; q < K; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r4[(u * K + q) & 7]) : "v"(laddr), "n"(q * 1024));
                }
                if (MODE == 4 && u < K) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(r4[u & 7]) : "v"(gp), "n"(0));
                if (MODE == 5 && u < K) asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(gq), "v"(r4[u & 7]));
                if (MODE == 4 && u == 14) { asm volatile("s_waitcnt vmcnt(0)"); for (int q = 0; q < 8; ++q) asm volatile("" :: "v"(r4[q])); }
                if (MODE == 2 && u == 14) { asm volatile("s_waitcnt lgkmcnt(0)"); for (int q = 0; q < 8; ++q) asm volatile("" :: "v"(r4[q])); }
            }
        }
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        for (int i = 0; i < 8; ++i) s += sv[i];
    } else if (MODE == 3) {
        if (K >= 100) __builtin_amdgcn_s_setprio(3);
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = a + i + threadIdx.x;
        for (int it = 0; it < siters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], b, a);
        }
        for (int i = 0; i < 16; ++i) s += v[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 7) tm[wave] = wall_clock64() - t0;
}

template <int MODE, int K>
float run(float* d, int miters, int siters, float* w = nullptr) {
    static long long* tm = nullptr; if (!tm) hipMalloc(&tm, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, K>), dim3(256), dim3(512), 0, 0, d, tm, miters, siters, 1.0f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    if (w) { long long h[8]; hipMemcpy(h, tm, 64, hipMemcpyDeviceToHost); w[0] = h[0] * 1e-5f; w[1] = h[4] * 1e-5f; }
    return best;
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    float* d; hipMalloc(&d, 1 << 24);
    const int M = 8000;     // 8000*16 MFMAs
    const float base = run<0, 0>(d, M, 0);
    printf("bf16 MFMA only: %.3f ms  (%.1f cycles per MFMA at 2.4 GHz nominal)\n", base, base * 2.4e6 / (M * 16.0));
    printf("same wave, k v_fma per MFMA: k=1 %.3f  k=2 %.3f  k=4 %.3f  k=8 %.3f ms\n", run<1, 1>(d, M, 0), run<1, 2>(d, M, 0), run<1, 4>(d, M, 0), run<1, 8>(d, M, 0));
    printf("same wave, ds_read_b128 per MFMA (first 8 of 16): k=1 %.3f  k=2 %.3f ms\n", run<2, 1>(d, M, 0), run<2, 2>(d, M, 0));
    printf("same wave, K global_load_dwordx4 per 16 MFMAs: K=2 %.3f  K=4 %.3f  K=8 %.3f ms\n", run<4, 2>(d, M, 0), run<4, 4>(d, M, 0), run<4, 8>(d, M, 0));
    printf("same wave, K global_store_dwordx4 per 16 MFMAs: K=2 %.3f  K=4 %.3f ms\n", run<5, 2>(d, M, 0), run<5, 4>(d, M, 0));
    float w[2];
    for (int S = 5000; S <= 20000; S *= 2) {
        float t = run<3, 0>(d, M, S, w), ta = run<3, 0>(d, 0, S);
        printf("co-resident VALU wave x%d: %.3f ms (MFMA wave %.3f, side wave %.3f), side alone %.3f ms\n", S, t, w[0], w[1], ta);
    }
    printf("-- MFMA wave yields: s_nop n after every MFMA; side wave VALU x10000 (alone 0.75 ms) --\n");
#define Y(K) { float t0 = run<3, K>(d, M, 0), t = run<3, K>(d, M, 10000, w); printf("s_nop %2d: MFMA only %.3f ms; with side wave %.3f ms (MFMA wave %.3f, side wave %.3f)\n", K - 1, t0, t, w[0], w[1]); }
    Y(1) Y(2) Y(3) Y(4) Y(5) Y(6) Y(8)
    { float t = run<3, 100>(d, M, 10000, w); printf("side wave s_setprio 3: %.3f ms (MFMA wave %.3f, side wave %.3f)\n", t, w[0], w[1]); }
    return 0;
}

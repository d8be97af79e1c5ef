// Does side work of a co-resident wave steal time from the MFMA stream of another wave on the same SIMD?
// One workgroup of 512 threads per CU: waves 0-3 (one per SIMD) run M fp32 MFMAs each; waves 4-7 (their SIMD partners)
// run a side loop of the selected kind for a fixed count.  Reported: time of the launch vs the MFMA-only time.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_share tools/mfma_share.hip && ./mfma_share
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int SWAP, int PRIO, int NOP = 0>
__global__ __launch_bounds__(512) void k(float* out, const float* in, long long* tm, int miters, int siters, float a, float b) {
    __shared__ float lds[8192];
    const int wave = SWAP ? ((threadIdx.x >> 6) ^ 4) : (threadIdx.x >> 6);
    if (PRIO && wave >= 4) __builtin_amdgcn_s_setprio(3);
    lds[threadIdx.x] = a; lds[threadIdx.x + 512] = b;
    __syncthreads();
    const long long t0 = wall_clock64();
    float s = 0.f;
    if (wave < 4) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        float av = a + threadIdx.x * 1e-6f, bv = b;
        float sv[8];
        for (int i = 0; i < 8; ++i) sv[i] = a * i;
        unsigned sc = 0;
        f32x4 lv[4] = {}; float lw[4] = {};
        const f32x4* lp = (const f32x4*)lds + (threadIdx.x & 63);
        const f32x4* gp = (const f32x4*)in + (threadIdx.x & 63);
        for (int it = 0; it < miters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
                    if (NOP >= 1 && NOP < 10) asm volatile("s_nop 15");
                    if (NOP >= 2 && NOP < 10) asm volatile("s_nop 15");
                    if (NOP >= 3 && NOP < 10) asm volatile("s_nop 15");
                    if (NOP >= 30 && NOP < 50) {
#pragma unroll
                        for (int q = 0; q < NOP - 30; ++q) { sc = sc * 3u + 1u; asm volatile("" : "+s"(sc)); }
                    } else if (NOP >= 50 && NOP < 70) {
#pragma unroll
                        for (int q = 0; q < NOP - 50; ++q) { asm volatile("" : "+v"(lp)); lv[q & 3] += lp[q * 64]; }
                    } else if (NOP >= 70 && NOP < 90) {
#pragma unroll
                        for (int q = 0; q < NOP - 70; ++q) { asm volatile("" : "+v"(gp)); lv[q & 3] += gp[q * 64]; }
                    } else if (NOP >= 90) {
#pragma unroll
                        for (int q = 0; q < NOP - 90; ++q) { asm volatile("" : "+v"(lp)); lw[q & 3] += ((const float*)lp)[q * 64]; }
                    } else if (NOP >= 10) {
#pragma unroll
                        for (int q = 0; q < NOP - 10; ++q) { sv[q & 7] = __builtin_fmaf(sv[q & 7], bv, av); asm volatile("" : "+v"(sv[q & 7])); }
                    }
                }
        }
        for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
        for (int i = 0; i < 8; ++i) s += sv[i];
        s += sc; for (int i = 0; i < 4; ++i) s += lv[i].x + lv[i].y + lv[i].z + lv[i].w + lw[i];
    } else if (KIND == 1) {            // VALU: independent fma chains (64 v_fma per iteration)
        float v[16];
        for (int i = 0; i < 16; ++i) v[i] = a + i + threadIdx.x;
        for (int it = 0; it < siters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], b, a);
        }
        for (int i = 0; i < 16; ++i) s += v[i];
    } else if (KIND == 2) {            // LDS reads (64 ds_read_b128 per iteration)
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const f32x4* p = (const f32x4*)lds + (threadIdx.x & 63);
        for (int it = 0; it < siters; ++it) {
#pragma unroll
            for (int u = 0; u < 64; ++u) { v += p[(u & 7) * 64]; asm volatile("" : "+v"(v)); }
        }
        s = v.x + v.y + v.z + v.w;
    } else if (KIND == 3) {            // global loads, L1/L2 resident (64 dwordx4 per iteration)
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const f32x4* p = (const f32x4*)in + (threadIdx.x & 63);
        for (int it = 0; it < siters; ++it) {
#pragma unroll
            for (int u = 0; u < 64; ++u) { v += p[(u & 15) * 64]; asm volatile("" : "+v"(v)); }
        }
        s = v.x + v.y + v.z + v.w;
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 7) tm[wave] = wall_clock64() - t0;      // 100 MHz ticks of this wave
}

template <int KIND, int SWAP = 0, int PRIO = 0, int NOP = 0>
float run(float* d, float* in, int miters, int siters, float* wave_ms = nullptr) {
    static long long* tm = nullptr; if (!tm) hipMalloc(&tm, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND, SWAP, PRIO, NOP>), dim3(256), dim3(512), 0, 0, d, in, tm, miters, siters, 1.0f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    if (wave_ms) { long long h[8]; hipMemcpy(h, tm, 64, hipMemcpyDeviceToHost); wave_ms[0] = h[0] * 1e-5f; wave_ms[1] = h[4] * 1e-5f; }
    return best;
}

int main() {
    float *d, *in; hipMalloc(&d, 1 << 24); hipMalloc(&in, 1 << 24); hipMemset(in, 0, 1 << 24);
    const int M = 2000;                      // 2000 * 64 MFMAs * 64 cycles = 8.2 M cycles = 3.4 ms
    printf("MFMA only              : %.3f ms\n", run<0>(d, in, M, 0));
    const char* names[4] = {"", "VALU fma", "LDS b128 reads", "global loads"};
    for (int s = 0; s <= 3; ++s) {
        const int S = 10000 << s;
        float w[2]; float t1 = run<1>(d, in, M, S, w), t1a = run<1>(d, in, 0, S);
        printf("%-15s x%6d: with MFMA %.3f ms (MFMA wave %.3f, side wave %.3f)   side alone %.3f ms\n", names[1], S, t1, w[0], w[1], t1a);
    }
    for (int s = 0; s <= 2; ++s) {
        const int S = 2000 << s;
        float w[2]; float t2 = run<2>(d, in, M, S, w), t2a = run<2>(d, in, 0, S);
        printf("%-15s x%6d: with MFMA %.3f ms (MFMA wave %.3f, side wave %.3f)   side alone %.3f ms\n", names[2], S, t2, w[0], w[1], t2a);
        float t3 = run<3>(d, in, M, S, w), t3a = run<3>(d, in, 0, S);
        printf("%-15s x%6d: with MFMA %.3f ms (MFMA wave %.3f, side wave %.3f)   side alone %.3f ms\n", names[3], S, t3, w[0], w[1], t3a);
    }
    printf("-- side waves are the OLDER waves of the workgroup --\n");
    { float w[2]; float t = run<1, 1, 0>(d, in, M, 40000, w); printf("VALU x40000 : %.3f ms (MFMA wave %.3f, side wave %.3f)\n", t, w[0], w[1]);
      t = run<2, 1, 0>(d, in, M, 2000, w); printf("LDS  x2000  : %.3f ms (MFMA wave %.3f, side wave %.3f)\n", t, w[0], w[1]);
      t = run<3, 1, 0>(d, in, M, 2000, w); printf("VMEM x2000  : %.3f ms (MFMA wave %.3f, side wave %.3f)\n", t, w[0], w[1]); }
    printf("-- side waves younger, s_setprio 3 --\n");
    { float w[2]; float t = run<1, 0, 1>(d, in, M, 40000, w); printf("VALU x40000 : %.3f ms (MFMA wave %.3f, side wave %.3f)\n", t, w[0], w[1]);
      t = run<2, 0, 1>(d, in, M, 2000, w); printf("LDS  x2000  : %.3f ms (MFMA wave %.3f, side wave %.3f)\n", t, w[0], w[1]);
      t = run<3, 0, 1>(d, in, M, 2000, w); printf("VMEM x2000  : %.3f ms (MFMA wave %.3f, side wave %.3f)\n", t, w[0], w[1]); }
    printf("-- MFMA wave executes s_nop 15 x1/x2/x3 after every MFMA --\n");
    { float w[2]; float t;
      t = run<0, 0, 0, 1>(d, in, M, 0, w); printf("nop1 MFMA only: %.3f ms\n", t);
      t = run<0, 0, 0, 2>(d, in, M, 0, w); printf("nop2 MFMA only: %.3f ms\n", t);
      t = run<0, 0, 0, 3>(d, in, M, 0, w); printf("nop3 MFMA only: %.3f ms\n", t);
      t = run<1, 0, 0, 1>(d, in, M, 40000, w); printf("nop1 VALU x40000 : %.3f ms (MFMA wave %.3f, side wave %.3f)\n", t, w[0], w[1]);
      t = run<1, 0, 0, 2>(d, in, M, 40000, w); printf("nop2 VALU x40000 : %.3f ms (MFMA wave %.3f, side wave %.3f)\n", t, w[0], w[1]);
      t = run<1, 0, 0, 3>(d, in, M, 40000, w); printf("nop3 VALU x40000 : %.3f ms (MFMA wave %.3f, side wave %.3f)\n", t, w[0], w[1]);
      t = run<2, 0, 0, 2>(d, in, M, 2000, w); printf("nop2 LDS  x2000  : %.3f ms (MFMA wave %.3f, side wave %.3f)\n", t, w[0], w[1]);
      t = run<3, 0, 0, 2>(d, in, M, 2000, w); printf("nop2 VMEM x2000  : %.3f ms (MFMA wave %.3f, side wave %.3f)\n", t, w[0], w[1]); }
    printf("-- same wave: k v_fma after every MFMA --\n");
    { float t;
      t = run<0, 0, 0, 12>(d, in, M, 0); printf("k=2 : %.3f ms\n", t);
      t = run<0, 0, 0, 14>(d, in, M, 0); printf("k=4 : %.3f ms\n", t);
      t = run<0, 0, 0, 18>(d, in, M, 0); printf("k=8 : %.3f ms\n", t);
      t = run<0, 0, 0, 26>(d, in, M, 0); printf("k=16: %.3f ms\n", t); }
    printf("-- same wave, per MFMA: k s_add / k ds_read_b128 / k global_load_dwordx4 / k ds_read_b32 --\n");
    { float t;
      t = run<0, 0, 0, 34>(d, in, M, 0); printf("4 s_add      : %.3f ms\n", t);
      t = run<0, 0, 0, 46>(d, in, M, 0); printf("16 s_add     : %.3f ms\n", t);
      t = run<0, 0, 0, 51>(d, in, M, 0); printf("1 ds_b128    : %.3f ms\n", t);
      t = run<0, 0, 0, 54>(d, in, M, 0); printf("4 ds_b128    : %.3f ms\n", t);
      t = run<0, 0, 0, 71>(d, in, M, 0); printf("1 global x4  : %.3f ms\n", t);
      t = run<0, 0, 0, 74>(d, in, M, 0); printf("4 global x4  : %.3f ms\n", t);
      t = run<0, 0, 0, 94>(d, in, M, 0); printf("4 ds_b32     : %.3f ms\n", t);
      t = run<0, 0, 0, 98>(d, in, M, 0); printf("8 ds_b32     : %.3f ms\n", t); }
    return 0;
}

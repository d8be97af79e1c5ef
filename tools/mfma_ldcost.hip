// What does a load cost an fp32 MFMA stream?  Per iteration one wave issues 16 MFMAs (1024 cycles of SIMD time) and K loads of
// one kind right after the first MFMA; their results are waited for before the last MFMA and never touched by the VALU.
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_ldcost tools/mfma_ldcost.hip && ./tools/mfma_ldcost
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int K, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, const float* in, int iters, float a, float b) {
    __shared__ f32x4 lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64 * WAVES) lds[i] = (f32x4){a, b, a, b};
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float av = a + threadIdx.x * 1e-6f, bv = b;
    const unsigned laddr = (threadIdx.x & 63) * 16;
    const f32x4* gp = (const f32x4*)in + (threadIdx.x & 63);
    f32x4 r4[16]; float r1[16]; unsigned sc[8] = {0, 1, 2, 3, 4, 5, 6, 7}; unsigned va[8] = {0, 1, 2, 3, 4, 5, 6, 7};
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)in, 0, 1 << 24, 0x00020000);
    for (int it = 0; it < iters; ++it) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[0], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < K; ++q) {
            if (KIND == 1) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r1[q]) : "v"(laddr), "n"(q * 1024));
            if (KIND == 2) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r4[q]) : "v"(laddr), "n"(q * 1024));
            if (KIND == 3) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(r4[q]) : "v"(gp), "n"(q * 1024 % 4096));
            if (KIND == 4) asm volatile("global_load_dword %0, %1, off offset:%2" : "=v"(r1[q]) : "v"(gp), "n"(q * 1024 % 4096));
            if (KIND == 5) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sc[q & 7]) :: "scc");
            if (KIND == 6) asm volatile("s_mul_i32 %0, %0, 3" : "+s"(sc[q & 7]));
            if (KIND == 7) { unsigned lo = laddr; asm volatile("" : "+v"(lo)); r4[q & 15] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)lo + (q * 256 % 4096), 0, 0)); }
            if (KIND == 8) asm volatile("v_add_u32 %0, %0, %1" : "+v"(va[q & 7]) : "v"(laddr));
            if (KIND == 9) { unsigned lo = laddr; asm volatile("" : "+v"(lo));       // direct-to-LDS: no VGPR write-back
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)((char*)lds + (q & 15) * 1024 + (threadIdx.x >> 6) * 16384 % 32768), 16, (int)lo + (q * 256 % 4096), 0, 0, 0); }
            if (KIND == 10) asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(laddr), "v"(r4[0]), "n"(q * 1024) : "memory");
        }
#pragma unroll
        for (int u = 1; u < 15; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[u & 3], 0, 0, 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
#pragma unroll
        for (int q = 0; q < K && q < 16; ++q) { if (KIND == 1 || KIND == 4) asm volatile("" :: "v"(r1[q])); else if (KIND == 2 || KIND == 3 || KIND == 7) asm volatile("" :: "v"(r4[q])); }
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[3], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += sc[i] + va[i];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
}

template <int KIND, int K, int WAVES>
float run(float* d, float* in, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND, K, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, d, in, iters, 1.0f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

template <int WAVES>
void table(float* d, float* in) {
    const int IT = 8000;
    const float base = run<0, 0, WAVES>(d, in, IT);
    const double cyc = 1024.0 * (WAVES / 4);      // MFMA cycles per iteration per SIMD
    printf("%d waves per CU: 16 MFMAs alone %.3f ms\n", WAVES, base);
#define ROW(KIND, K, name) { float t = run<KIND, K, WAVES>(d, in, IT); printf("  + %2d %-20s %.3f ms  -> %.1f cycles per load\n", K, name, t, (t / base - 1.0) * cyc / K / (WAVES / 4)); }
    ROW(1, 8, "ds_read_b32") ROW(1, 16, "ds_read_b32") ROW(2, 4, "ds_read_b128") ROW(2, 8, "ds_read_b128")
    ROW(4, 8, "global_load_dword") ROW(3, 4, "global_load_dwordx4") ROW(3, 8, "global_load_dwordx4")
    ROW(7, 4, "buffer_load_dwordx4") ROW(7, 8, "buffer_load_dwordx4") ROW(9, 4, "buffer_load_dwordx4 lds") ROW(9, 8, "buffer_load_dwordx4 lds") ROW(10, 4, "ds_write_b128") ROW(10, 8, "ds_write_b128") ROW(5, 32, "s_add_u32") ROW(5, 64, "s_add_u32") ROW(6, 32, "s_mul_i32") ROW(8, 32, "v_add_u32") ROW(8, 64, "v_add_u32")
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    float *d, *in; hipMalloc(&d, 1 << 24); hipMalloc(&in, 1 << 24); hipMemset(in, 0, 1 << 24);
    table<4>(d, in);
    table<8>(d, in);
    return 0;
}

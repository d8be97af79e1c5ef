// Calibration of rocprofv3's FETCH_SIZE (= TCC_EA0_RDREQ x 64 B... or not) on the access patterns the conv kernels stage with:
//   k_wide   : 16 B per lane, contiguous (1 KB per wave instruction)            -- the pattern MI355X_MICROARCH.md's "x2" rule is stated for
//   k_chunk32: 32-B chunks at a 128-B stride (two lanes per chunk)               -- one 16-cin slice of bf16 NDHWC rows (conv64_bf16_kernel)
//   k_chunk64: 64-B chunks at a 256-B stride (four lanes per chunk)              -- one 16-cin slice of fp32 NDHWC rows (conv64_wino_kernel)
//   k_row128 : whole 128-B rows (eight lanes per row), contiguous                -- bf16 rows fetched in one piece
// Every kernel reads `bytes_req` REQUESTED bytes out of a buffer far larger than the 256 MB Infinity Cache, once, and prints that
// number; run under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` and divide (tools/fetch_calib.sh).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_wide(const u32x4* __restrict__ x, unsigned* out, size_t n16) {
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) acc ^= x[i];
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}
// chunk = CH bytes (CH/16 lanes), stride = ST bytes, slice s selects the chunk inside the row
template <int CH, int ST>
__global__ __launch_bounds__(256) void k_chunk(const char* __restrict__ x, unsigned* out, size_t nrows, int slice) {
    constexpr int LPC = CH / 16;
    u32x4 acc = {0, 0, 0, 0};
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (size_t i = t; i < nrows * LPC; i += (size_t)gridDim.x * 256) {
        const size_t row = i / LPC, part = i % LPC;
        acc ^= *(const u32x4*)(x + row * ST + (size_t)slice * CH + part * 16);
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

int main() {
    const size_t bytes = (size_t)3 << 30;                 // 3 GiB: 12x the Infinity Cache
    char* d; unsigned* o;
    if (hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&o, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d, 1, bytes);
    hipDeviceSynchronize();
    const int grid = 256 * 8;
    hipLaunchKernelGGL(k_wide, dim3(grid), dim3(256), 0, 0, (const u32x4*)d, o, bytes / 16);
    printf("k_wide            requested %zu bytes (contiguous 16 B per lane)\n", bytes);
    hipLaunchKernelGGL((k_chunk<32, 128>), dim3(grid), dim3(256), 0, 0, d, o, bytes / 128, 1);
    printf("k_chunk<32, 128>  requested %zu bytes (32-B chunk of every 128-B row; sectors touched = %zu x 64 B)\n", bytes / 4, bytes / 128);
    hipLaunchKernelGGL((k_chunk<64, 256>), dim3(grid), dim3(256), 0, 0, d, o, bytes / 256, 1);
    printf("k_chunk<64, 256>  requested %zu bytes (64-B chunk of every 256-B row)\n", bytes / 4);
    hipLaunchKernelGGL((k_chunk<128, 128>), dim3(grid), dim3(256), 0, 0, d, o, bytes / 128, 0);
    printf("k_chunk<128, 128> requested %zu bytes (whole 128-B rows, 8 lanes per row)\n", bytes);
    hipLaunchKernelGGL((k_chunk<16, 128>), dim3(grid), dim3(256), 0, 0, d, o, bytes / 128, 3);
    printf("k_chunk<16, 128>  requested %zu bytes (16-B chunk of every 128-B row)\n", bytes / 8);
    hipDeviceSynchronize();
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return 0;
}

// On-device input pipeline: the per-sample work of PatchHandler3D.load_patches_from_index_file
// (src/Network/PatchHandler3D.py:49-160) -- slice a P^3 (or (PR)^3) patch out of a resident 4-D volume, np.rot90 it
// in one of three planes, flip the sign of a velocity component, divide by venc / 4095, or threshold the mask --
// as ONE gather kernel per output tensor, driven by a small descriptor table built from the CSV rows.
// HBM-bound and tiny (2.1 MB per sample); its purpose is to take h5py + numpy slicing + H2D copies off the step's
// critical path when training from real data.
#include "fdn_common.h"

struct PatchDesc {          // one per (sample, output tensor); mirrored by data_device.py (8 x int64)
    const float* src;       // volume base, layout (T, X, Y, Z)
    int32_t X, Y;
    int32_t Z, t;
    int32_t x0, y0;
    int32_t z0, plane;      // plane 0 = no rotation, 1:(0,1) 2:(0,2) 3:(1,2)  (np.rot90 axes)
    int32_t k, mode;        // k = rot90 count (1..3); mode 0 = sign * (v / div), 1 = (v >= thr) ? 1 : 0
    float sign, div;        // div = venc or 4095; thr is passed in `div` for mode 1
};

__global__ __launch_bounds__(256) void gather_patches_kernel(const PatchDesc* __restrict__ desc, float* __restrict__ out, int B,
                                                              int S) {
    const int64_t per = (int64_t)S * S * S;
    const int64_t total = per * B;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / per);
        int r = (int)(i - (int64_t)b * per);
        int c[3];
        c[0] = r / (S * S); r -= c[0] * S * S;
        c[1] = r / S;
        c[2] = r - c[1] * S;
        const PatchDesc d = desc[b];
        if (d.plane) {
            // out = np.rot90(m, k, axes=(a,b)):  k=1: out[ia,ib] = m[ib, S-1-ia];  k=2: m[S-1-ia, S-1-ib];  k=3: m[S-1-ib, ia]
            const int a = d.plane == 3 ? 1 : 0, bb = d.plane == 1 ? 1 : 2;
            const int ia = c[a], ib = c[bb];
            if (d.k == 1) { c[a] = ib; c[bb] = S - 1 - ia; }
            else if (d.k == 2) { c[a] = S - 1 - ia; c[bb] = S - 1 - ib; }
            else if (d.k == 3) { c[a] = S - 1 - ib; c[bb] = ia; }
        }
        const float v = d.src[(((int64_t)d.t * d.X + d.x0 + c[0]) * d.Y + d.y0 + c[1]) * d.Z + d.z0 + c[2]];
        out[i] = d.mode ? (v >= d.div ? 1.f : 0.f) : d.sign * (v / d.div);
    }
}

extern "C" int fdn_gather_patches(const void* desc, float* out, int B, int S, void* stream) {
    FDN_REQUIRE(desc && out && B > 0 && S > 0, "fdn_gather_patches: bad argument");
    const int64_t total = (int64_t)B * S * S * S;
    int64_t nb = (total + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(gather_patches_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const PatchDesc*)desc,
                       out, B, S);
    FDN_CHECK_LAUNCH("gather_patches_kernel");
    return FDN_OK;
}

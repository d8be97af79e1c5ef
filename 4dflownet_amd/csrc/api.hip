// extern "C" boundary of lib4dflow_hip.so: argument validation, shape dispatch, error reporting.
// See include/fdn.h for the contract and the reference call sites each entry point replaces.
#include <stdarg.h>
#include <stdio.h>
#include <mutex>
#include <set>
#include <utility>
#include "fdn_common.h"

static thread_local char g_err[512] = "";

void fdn_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int fdn_func_max_lds(const void* fn, int bytes, const char* who) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;       // (device, kernel) pairs whose attribute is set
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) dev = 0;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({dev, fn})) return FDN_OK;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) { fdn_set_error("%s: hipFuncSetAttribute(%d B of LDS): %s", who, bytes, hipGetErrorString(e)); return FDN_ERR_HIP; }
    done.insert({dev, fn});
    return FDN_OK;
}

FDN_HOOK_VAR(int, fdn_wgrad64_force_direct, 0);
#ifdef FDN_TEST_HOOKS
extern "C" int fdn_debug_set_wgrad64_direct(int on) { fdn_wgrad64_force_direct = on; return FDN_OK; }
#endif

extern "C" int fdn_version(void) { return FDN_VERSION; }   // 161: fdn_conv64_dgrad_fused_multi; 160: FDN_CONV64_PACK_FLOATS = 423 * 4096 (+ the bf16 x 3 stream), FDN_ALGO_WINO_BF16X3
extern "C" const char* fdn_last_error(void) { return g_err; }

// small-channel kernels (small_convs.hip), templated on the activation storage type (float / uint16_t = bf16 bits)
template <typename T> int fdn_conv_cin3_fwd_launch(const T* x, const float* w, const float* bias, T* y, int N, int D, int H,
                                                   int W, int act, float alpha, hipStream_t s);
template <typename T> int fdn_conv_cout1_fwd_launch(const T* x, const float* w, const float* bias, float* y, int N, int D,
                                                    int H, int W, int ldy, int y_coff, int act, float alpha, hipStream_t s);
template <typename T> int fdn_conv1x1_fwd_launch(const T* xa, const T* xb, const float* w, const float* bias, T* y,
                                                 int64_t nvox, int act, float alpha, hipStream_t s);
template <typename T> int fdn_conv1x1_dgrad_launch(const T* dz, const float* w, const T* ya, const T* yb, T* dxa, T* dxb,
                                                   int64_t nvox, hipStream_t s);
int fdn_conv_cout1_dgrad_launch(const float* dz, const float* w, float* dxpad, int N, int D, int H, int W, int lddz,
                                int dz_coff, hipStream_t s);
template <typename T> int fdn_wgrad_cin3_launch(const T* x, const T* dz, float* dw, void* ws, size_t ws_bytes, int N, int D,
                                                int H, int W, hipStream_t s);
template <typename T> int fdn_wgrad_cout1_launch(const T* x, const float* dz, float* dw, void* ws, size_t ws_bytes, int N,
                                                 int D, int H, int W, int lddz, int dz_coff, hipStream_t s);
template <typename T> int fdn_conv_cout1_dgrad_folded_launch(const float* dz, const float* w, const T* y_prev, int act,
                                                             float alpha, T* dz_prev, float* dbias_prev, void* workspace,
                                                             size_t workspace_bytes, int N, int D, int H, int W, int lddz,
                                                             int dz_coff, hipStream_t s, const uint16_t* ymask = nullptr);
template <typename T> int fdn_wgrad_1x1_launch(const T* xa, const T* xb, const T* dz, float* dw, void* ws, size_t ws_bytes,
                                               int64_t nvox, hipStream_t s);
template <typename T> int fdn_bias_grad_launch(const T* dz, float* db, void* ws, size_t ws_bytes, int64_t nvox, int C,
                                               int lddz, int dz_coff, hipStream_t s);
size_t fdn_small_wgrad_workspace_bytes(int Cin, int Cout, int K);

extern "C" int fdn_conv3d_fwd(const float* x, const float* x2, const float* w, const float* wpack, const float* bias,
                              const float* residual, float* y, int N, int D, int H, int W, int Cin, int Cout, int K,
                              int ldy, int y_coff, int act, float alpha, int algo, void* stream) {
    FDN_REQUIRE(x && y, "fdn_conv3d_fwd: x/y is NULL");
    FDN_REQUIRE(algo >= FDN_ALGO_AUTO && algo <= FDN_ALGO_LAST, "fdn_conv3d_fwd: bad algo %d", algo);
    FDN_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "fdn_conv3d_fwd: bad dims N=%d D=%d H=%d W=%d", N, D, H, W);
    FDN_REQUIRE(act >= FDN_ACT_NONE && act <= FDN_ACT_LEAKY, "fdn_conv3d_fwd: bad act %d", act);
    hipStream_t s = (hipStream_t)stream;
    if (Cin == 64 && Cout == 64 && K == 3) {
        FDN_REQUIRE(wpack, "fdn_conv3d_fwd: the 64->64 MFMA path needs wpack (fdn_pack_conv64_weights)");
        FDN_REQUIRE(ldy == 64 && y_coff == 0, "fdn_conv3d_fwd: 64->64 path writes dense rows (ldy=64,y_coff=0)");
        FDN_REQUIRE(D <= 1022 && H <= 1022 && W <= 1022, "fdn_conv3d_fwd: dims too large");
        return fdn_conv64_launch(x, wpack, bias, residual, y, N, D, H, W, D, H, W, 0, 0, act, alpha, s, algo);
    }
    FDN_REQUIRE(residual == nullptr, "fdn_conv3d_fwd: residual only on the 64->64 path");
    if (Cin == 3 && Cout == 64 && K == 3) {
        FDN_REQUIRE(w && ldy == 64 && y_coff == 0, "fdn_conv3d_fwd(3->64): needs w, dense output");
        return fdn_conv_cin3_fwd_launch(x, w, bias, y, N, D, H, W, act, alpha, s);
    }
    if (Cin == 64 && Cout == 1 && K == 3) {
        FDN_REQUIRE(w && ldy >= 1 && y_coff >= 0 && y_coff < ldy, "fdn_conv3d_fwd(64->1): bad w/ldy/y_coff");
        return fdn_conv_cout1_fwd_launch(x, w, bias, y, N, D, H, W, ldy, y_coff, act, alpha, s);
    }
    if (Cin == 128 && Cout == 64 && K == 1) {
        FDN_REQUIRE(w && x2 && ldy == 64 && y_coff == 0, "fdn_conv3d_fwd(1x1 128->64): needs w, x2, dense output");
        return fdn_conv1x1_fwd_launch(x, x2, w, bias, y, (int64_t)N * D * H * W, act, alpha, s);
    }
    fdn_set_error("fdn_conv3d_fwd: unsupported (Cin=%d,Cout=%d,K=%d)", Cin, Cout, K);
    return FDN_ERR_UNSUPPORTED;
}

extern "C" int fdn_conv3d_dgrad(const float* dz, const float* w, const float* wpack, float* dxpad, int N, int D,
                                int H, int W, int Cin, int Cout, int K, int lddz, int dz_coff, int algo, void* stream) {
    FDN_REQUIRE(dz && dxpad, "fdn_conv3d_dgrad: dz/dxpad is NULL");
    FDN_REQUIRE(algo >= FDN_ALGO_AUTO && algo <= FDN_ALGO_LAST, "fdn_conv3d_dgrad: bad algo %d", algo);
    FDN_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "fdn_conv3d_dgrad: bad dims");
    hipStream_t s = (hipStream_t)stream;
    if (Cin == 64 && Cout == 64 && K == 3) {
        FDN_REQUIRE(wpack, "fdn_conv3d_dgrad: the 64->64 MFMA path needs wpack = wp_dgrad");
        FDN_REQUIRE(lddz == 64 && dz_coff == 0, "fdn_conv3d_dgrad: 64->64 path reads dense rows");
        FDN_REQUIRE(D <= 1020 && H <= 1020 && W <= 1020, "fdn_conv3d_dgrad: dims too large");
        return fdn_conv64_launch(dz, wpack, nullptr, nullptr, dxpad, N, D, H, W, D + 2, H + 2, W + 2, -1, 1,
                                 FDN_ACT_NONE, 0.f, s, algo);
    }
    if (Cin == 64 && Cout == 1 && K == 3) {
        FDN_REQUIRE(w, "fdn_conv3d_dgrad(64->1): needs w");
        return fdn_conv_cout1_dgrad_launch(dz, w, dxpad, N, D, H, W, lddz, dz_coff, s);
    }
    fdn_set_error("fdn_conv3d_dgrad: unsupported (Cin=%d,Cout=%d,K=%d)", Cin, Cout, K);
    return FDN_ERR_UNSUPPORTED;
}

extern "C" int fdn_conv3d_dgrad_fused(const float* dz, const float* wpack, float* dxpad, const float* skip,
                                      const float* y_prev, int act, float alpha, float* dz_prev, int N, int D, int H,
                                      int W, int algo, void* stream) {
    FDN_REQUIRE(dz && wpack && dxpad && dz_prev, "fdn_conv3d_dgrad_fused: NULL argument");
    FDN_REQUIRE(algo >= FDN_ALGO_AUTO && algo <= FDN_ALGO_LAST, "fdn_conv3d_dgrad_fused: bad algo %d", algo);
    FDN_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && D <= 1020 && H <= 1020 && W <= 1020, "fdn_conv3d_dgrad_fused: bad dims");
    FDN_REQUIRE(act >= FDN_ACT_NONE && act <= FDN_ACT_LEAKY, "fdn_conv3d_dgrad_fused: bad act %d", act);
    return fdn_conv64_launch_ex(dz, wpack, nullptr, nullptr, dxpad, skip, y_prev, dz_prev, N, D, H, W, D + 2, H + 2, W + 2,
                                -1, 1, act, alpha, (hipStream_t)stream, 3, algo);
}

// 64->64 forward that also writes the sign mask of its output, and the fused dgrad that reads the mask instead of y_prev (conv64_wino2d_kernel.h)
extern "C" int fdn_conv64_fwd_mask(const float* x, const float* wpack, const float* bias, const float* residual, float* y, uint16_t* y_mask,
                                   int N, int D, int H, int W, int act, float alpha, int algo, void* stream) {
    FDN_REQUIRE(x && wpack && y && y_mask, "fdn_conv64_fwd_mask: NULL argument");
    FDN_REQUIRE(algo >= FDN_ALGO_AUTO && algo <= FDN_ALGO_LAST, "fdn_conv64_fwd_mask: bad algo %d", algo);
    FDN_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && D <= 1020 && H <= 1020 && W <= 1020, "fdn_conv64_fwd_mask: bad dims");
    FDN_REQUIRE(act >= FDN_ACT_NONE && act <= FDN_ACT_LEAKY, "fdn_conv64_fwd_mask: bad act %d", act);
    return fdn_conv64_launch_ex(x, wpack, bias, residual, y, nullptr, nullptr, nullptr, N, D, H, W, D, H, W, 0, 0, act, alpha,
                                (hipStream_t)stream, 3, algo, nullptr, y_mask, nullptr);
}

extern "C" int fdn_conv64_dgrad_fused_mask(const float* dz, const float* wpack, float* dxpad, const float* skip, const uint16_t* y_mask,
                                           int act, float alpha, float* dz_prev, int N, int D, int H, int W, int algo, void* stream) {
    FDN_REQUIRE(dz && wpack && dxpad && dz_prev && y_mask, "fdn_conv64_dgrad_fused_mask: NULL argument");
    FDN_REQUIRE(algo >= FDN_ALGO_AUTO && algo <= FDN_ALGO_LAST, "fdn_conv64_dgrad_fused_mask: bad algo %d", algo);
    FDN_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && D <= 1020 && H <= 1020 && W <= 1020, "fdn_conv64_dgrad_fused_mask: bad dims");
    FDN_REQUIRE(act == FDN_ACT_RELU || act == FDN_ACT_LEAKY, "fdn_conv64_dgrad_fused_mask: a mask belongs to an activation (act %d)", act);
    return fdn_conv64_launch_ex(dz, wpack, nullptr, nullptr, dxpad, skip, nullptr, dz_prev, N, D, H, W, D + 2, H + 2, W + 2,
                                -1, 1, act, alpha, (hipStream_t)stream, 3, algo, nullptr, nullptr, y_mask);
}

extern "C" int fdn_conv64_dgrad_fused_multi(const float* const* dz, const float* const* wpack, int nsrc, float* dxpad, const float* skip,
                                            const float* y_prev, const uint16_t* y_mask, int act, float alpha, float* dz_prev, int N, int D,
                                            int H, int W, int algo, void* stream) {
    FDN_REQUIRE(dz && wpack && dxpad && dz_prev && nsrc >= 1 && nsrc <= 3, "fdn_conv64_dgrad_fused_multi: NULL argument or nsrc outside 1..3");
    FDN_REQUIRE(algo >= FDN_ALGO_AUTO && algo <= FDN_ALGO_LAST, "fdn_conv64_dgrad_fused_multi: bad algo %d", algo);
    FDN_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && D <= 1020 && H <= 1020 && W <= 1020, "fdn_conv64_dgrad_fused_multi: bad dims");
    FDN_REQUIRE(act >= FDN_ACT_NONE && act <= FDN_ACT_LEAKY, "fdn_conv64_dgrad_fused_multi: bad act %d", act);
    FDN_REQUIRE(!(y_mask && y_prev), "fdn_conv64_dgrad_fused_multi: y_prev OR its sign mask");
    FDN_REQUIRE(!y_mask || act == FDN_ACT_RELU || act == FDN_ACT_LEAKY, "fdn_conv64_dgrad_fused_multi: a mask belongs to an activation (act %d)", act);
    // the kernels address the further sources' weight streams as non-negative byte distances from source 0's: order by pack address
    // (the sum over the sources is formed in that order)
    int ord[3] = {0, 1, 2};
    for (int i = 0; i < nsrc; ++i) FDN_REQUIRE(dz[i] && wpack[i], "fdn_conv64_dgrad_fused_multi: NULL pointer for source %d", i);
    for (int i = 1; i < nsrc; ++i)
        for (int j = i; j > 0 && wpack[ord[j]] < wpack[ord[j - 1]]; --j) { const int t = ord[j]; ord[j] = ord[j - 1]; ord[j - 1] = t; }
    FdnExtraSrc ex{nsrc, nullptr, nullptr, 0, 0};
    for (int i = 1; i < nsrc; ++i) {
        const long long d = (long long)(wpack[ord[i]] - wpack[ord[0]]) * (long long)sizeof(float);
        FDN_REQUIRE(d >= 0 && d < (1ll << 30), "fdn_conv64_dgrad_fused_multi: the packs of the sources must lie within 1 GiB of each other (one fdn_pack_conv64_weights_batch buffer)");
        if (i == 1) { ex.x1 = dz[ord[1]]; ex.wd1 = (int)d; } else { ex.x2 = dz[ord[2]]; ex.wd2 = (int)d; }
    }
    return fdn_conv64_launch_ex(dz[ord[0]], wpack[ord[0]], nullptr, nullptr, dxpad, skip, y_prev, dz_prev, N, D, H, W, D + 2, H + 2, W + 2,
                                -1, 1, act, alpha, (hipStream_t)stream, 3, algo, nullptr, nullptr, y_mask, nsrc > 1 ? &ex : nullptr);
}

extern "C" int fdn_conv3d_dgrad_fused_part(const float* dz, const float* wpack, float* dxpad, const float* skip,
                                           const float* y_prev, int act, float alpha, float* dz_prev, int N, int D, int H,
                                           int W, int parts, int algo, void* stream) {
    FDN_REQUIRE(dz && wpack && dxpad && dz_prev, "fdn_conv3d_dgrad_fused_part: NULL argument");
    FDN_REQUIRE(algo >= FDN_ALGO_AUTO && algo <= FDN_ALGO_LAST, "fdn_conv3d_dgrad_fused_part: bad algo %d", algo);
    FDN_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && D <= 1020 && H <= 1020 && W <= 1020, "fdn_conv3d_dgrad_fused_part: bad dims");
    FDN_REQUIRE(act >= FDN_ACT_NONE && act <= FDN_ACT_LEAKY, "fdn_conv3d_dgrad_fused_part: bad act %d", act);
    FDN_REQUIRE(parts >= 1 && parts <= 3, "fdn_conv3d_dgrad_fused_part: parts must be FDN_DGRAD_INNER | FDN_DGRAD_SHELL");
    return fdn_conv64_launch_ex(dz, wpack, nullptr, nullptr, dxpad, skip, y_prev, dz_prev, N, D, H, W, D + 2, H + 2, W + 2,
                                -1, 1, act, alpha, (hipStream_t)stream, parts, algo);
}

extern "C" int fdn_fold_halo_border(const float* dxpad0, const float* dxpad1, const float* dxpad2, int nsrc,
                                    const float* skip, const float* y_prev, int act, float alpha, float* dz_prev, int N,
                                    int D, int H, int W, void* stream) {
    FDN_REQUIRE(dxpad0 && dz_prev, "fdn_fold_halo_border: NULL argument");
    FDN_REQUIRE(nsrc >= 1 && nsrc <= 3 && (nsrc < 2 || dxpad1) && (nsrc < 3 || dxpad2), "fdn_fold_halo_border: bad nsrc %d", nsrc);
    FDN_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "fdn_fold_halo_border: bad dims");
    return fdn_fold_halo_border_launch(dxpad0, dxpad1, dxpad2, nsrc, skip, y_prev, act, alpha, dz_prev, N, D, H, W,
                                       (hipStream_t)stream);
}

extern "C" size_t fdn_conv3d_wgrad_workspace_bytes(int N, int D, int H, int W, int Cin, int Cout, int K) {
    if (Cin == 64 && Cout == 64 && K == 3) {
        const size_t wd = fdn_wgrad64_workspace_bytes(N, D, H, W), ww = fdn_wgrad64_wino_workspace_bytes(N, D, H, W);
        return wd > ww ? wd : ww;
    }
    return fdn_small_wgrad_workspace_bytes(Cin, Cout, K);
}

extern "C" int fdn_conv3d_wgrad(const float* x, const float* x2, const float* dz, float* dw, float* dbias,
                                void* workspace, size_t workspace_bytes, int N, int D, int H, int W, int Cin, int Cout,
                                int K, int lddz, int dz_coff, int algo, void* stream) {
    FDN_REQUIRE(x && dz && dw, "fdn_conv3d_wgrad: x/dz/dw is NULL");
    FDN_REQUIRE(algo >= FDN_ALGO_AUTO && algo <= FDN_ALGO_LAST, "fdn_conv3d_wgrad: bad algo %d", algo);
    FDN_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "fdn_conv3d_wgrad: bad dims");
    const size_t need = fdn_conv3d_wgrad_workspace_bytes(N, D, H, W, Cin, Cout, K);
    if (workspace_bytes < need || (need && !workspace)) {
        fdn_set_error("fdn_conv3d_wgrad: workspace %zu < %zu bytes", workspace_bytes, need);
        return FDN_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const int64_t nvox = (int64_t)N * D * H * W;
    int rc;
    if (Cin == 64 && Cout == 64 && K == 3) {
        FDN_REQUIRE(lddz == 64 && dz_coff == 0, "fdn_conv3d_wgrad: 64->64 path reads dense dz rows");
        // Winograd F(3,4) along W (wgrad64_wino.hip): half the multiplies of the direct kernel, which FDN_ALGO_DIRECT selects
        rc = (fdn_wgrad64_force_direct || algo == FDN_ALGO_DIRECT)
                 ? fdn_wgrad64_launch(x, dz, dw, workspace, workspace_bytes, N, D, H, W, s)
                 : fdn_wgrad64_wino_launch(x, dz, dw, workspace, workspace_bytes, N, D, H, W, s, algo);
    } else if (Cin == 3 && Cout == 64 && K == 3) {
        FDN_REQUIRE(lddz == 64 && dz_coff == 0, "fdn_conv3d_wgrad(3->64): dense dz rows");
        rc = fdn_wgrad_cin3_launch(x, dz, dw, workspace, workspace_bytes, N, D, H, W, s);
    } else if (Cin == 64 && Cout == 1 && K == 3) {
        rc = fdn_wgrad_cout1_launch(x, dz, dw, workspace, workspace_bytes, N, D, H, W, lddz, dz_coff, s);
    } else if (Cin == 128 && Cout == 64 && K == 1) {
        FDN_REQUIRE(x2 && lddz == 64 && dz_coff == 0, "fdn_conv3d_wgrad(1x1): needs x2, dense dz rows");
        rc = fdn_wgrad_1x1_launch(x, x2, dz, dw, workspace, workspace_bytes, nvox, s);
    } else {
        fdn_set_error("fdn_conv3d_wgrad: unsupported (Cin=%d,Cout=%d,K=%d)", Cin, Cout, K);
        return FDN_ERR_UNSUPPORTED;
    }
    if (rc != FDN_OK) return rc;
    if (dbias) return fdn_bias_grad_launch(dz, dbias, workspace, workspace_bytes, nvox, Cout, lddz, dz_coff, s);
    return FDN_OK;
}

// Weight gradients of n_layers 64->64 3x3x3 layers that share one grid, in ONE launch (+ one reduction launch) where the kernel allows
// it (W % 4 == 0, even D, FDN_ALGO_AUTO / _WINO_H2), else layer by layer through fdn_conv3d_wgrad's path.  The batched kernel gives
// every layer 64 / n_layers splits of the voxel sum where the single-layer launch uses 63: equal to fp32 rounding, not bit for bit.
static bool wgrad_batchable(int n_layers, int D, int W, int algo) {
    return !fdn_wgrad64_force_direct && (W & 3) == 0 && fdn_wgrad64_wino_batch_ok(n_layers, D, algo);
}
extern "C" size_t fdn_conv3d_wgrad_batch_workspace_bytes(int n_layers, int N, int D, int H, int W) {
    if (n_layers <= 0 || N <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    const size_t one = fdn_conv3d_wgrad_workspace_bytes(N, D, H, W, 64, 64, 3);
    size_t all = 0;                       // (the batch may go out in chunks of fewer layers with more splits each: fdn_conv3d_wgrad_batch)
    if (wgrad_batchable(n_layers, D, W, FDN_ALGO_AUTO))
        for (int c = 2; c <= n_layers; ++c) {
            const size_t b = fdn_wgrad64_wino_batch_workspace_bytes(c, N, D, H, W);
            if (b > all) all = b;
        }
    return all > one ? all : one;
}
extern "C" int fdn_conv3d_wgrad_batch(const float* const* x, const float* const* dz, float* const* dw, float* const* dbias, int n_layers,
                                      void* workspace, size_t workspace_bytes, int N, int D, int H, int W, int algo, void* stream) {
    FDN_REQUIRE(x && dz && dw && n_layers > 0, "fdn_conv3d_wgrad_batch: NULL table or n_layers <= 0");
    FDN_REQUIRE(algo >= FDN_ALGO_AUTO && algo <= FDN_ALGO_LAST, "fdn_conv3d_wgrad_batch: bad algo %d", algo);
    FDN_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "fdn_conv3d_wgrad_batch: bad dims");
    const size_t need = fdn_conv3d_wgrad_batch_workspace_bytes(n_layers, N, D, H, W);
    if (workspace_bytes < need || !workspace) {
        fdn_set_error("fdn_conv3d_wgrad_batch: workspace %zu < %zu bytes", workspace_bytes, need);
        return FDN_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    if (wgrad_batchable(n_layers, D, W, algo)) {
        // The batched kernel gives every layer 64 / n splits x 4 coordinates: n = 11 fills 220 of the 256 CUs.  Such a batch goes out in
        // chunks that fill the chip (>= 63 of 64 split slots: 1, 2, 3, 4, 7, 8, 9, 16 layers), a batch at >= 15/16 as it is.
        auto fill64 = [](int c) { return (64 / c) * c; };
        for (int i0 = 0; i0 < n_layers;) {
            const int rem = n_layers - i0;
            int c = rem;
            if (fill64(rem) < 60) { c = rem - 1; while (c > 1 && fill64(c) < 63) --c; }
            if (c >= 2) { if (int rc = fdn_wgrad64_wino_batch_launch(x + i0, dz + i0, dw + i0, c, workspace, workspace_bytes, N, D, H, W, s)) return rc; }
            else if (int rc = fdn_conv3d_wgrad(x[i0], nullptr, dz[i0], dw[i0], nullptr, workspace, workspace_bytes, N, D, H, W, 64, 64, 3, 64, 0, algo, stream)) return rc;
            i0 += c;
        }
    } else {
        for (int i = 0; i < n_layers; ++i) {
            FDN_REQUIRE(x[i] && dz[i] && dw[i], "fdn_conv3d_wgrad_batch: NULL pointer for layer %d", i);
            if (int rc = fdn_conv3d_wgrad(x[i], nullptr, dz[i], dw[i], nullptr, workspace, workspace_bytes, N, D, H, W, 64, 64, 3, 64, 0, algo, stream))
                return rc;
        }
    }
    if (dbias)                           // (stream-ordered behind the reduction: the workspace is free again)
        for (int i = 0; i < n_layers; ++i)
            if (dbias[i])
                if (int rc = fdn_bias_grad_launch(dz[i], dbias[i], workspace, workspace_bytes, (int64_t)N * D * H * W, 64, 64, 0, s)) return rc;
    return FDN_OK;
}

// ---- bf16 activation path ----
extern "C" int fdn_conv64_fwd_bf16_mask(const uint16_t* x, const uint16_t* wpack, const float* bias, const uint16_t* residual,
                                        uint16_t* y, uint16_t* y_mask, int N, int D, int H, int W, int act, float alpha, void* stream) {
    FDN_REQUIRE(x && wpack && y, "fdn_conv64_fwd_bf16: NULL argument");
    FDN_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && D <= 1022 && H <= 1022 && W <= 1022, "fdn_conv64_fwd_bf16: bad dims");
    FDN_REQUIRE(act >= FDN_ACT_NONE && act <= FDN_ACT_LEAKY, "fdn_conv64_fwd_bf16: bad act %d", act);
    return fdn_conv64_bf16_launch(x, wpack, bias, residual, y, nullptr, nullptr, nullptr, nullptr, N, D, H, W, D, H, W, 0, 0,
                                  act, alpha, (hipStream_t)stream, y_mask, nullptr);
}

extern "C" int fdn_conv64_fwd_bf16(const uint16_t* x, const uint16_t* wpack, const float* bias, const uint16_t* residual,
                                   uint16_t* y, int N, int D, int H, int W, int act, float alpha, void* stream) {
    return fdn_conv64_fwd_bf16_mask(x, wpack, bias, residual, y, nullptr, N, D, H, W, act, alpha, stream);
}

extern "C" int fdn_conv64_dgrad_fused_bf16_mask(const uint16_t* dz, const uint16_t* wpack, float* dxpad, const uint16_t* skip,
                                                const uint16_t* y_prev, const uint16_t* y_mask, int act, float alpha,
                                                uint16_t* dz_prev, int N, int D, int H, int W, void* stream) {
    FDN_REQUIRE(dz && wpack && dxpad && dz_prev, "fdn_conv64_dgrad_fused_bf16: NULL argument");
    FDN_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && D <= 1020 && H <= 1020 && W <= 1020, "fdn_conv64_dgrad_fused_bf16: bad dims");
    FDN_REQUIRE(act >= FDN_ACT_NONE && act <= FDN_ACT_LEAKY, "fdn_conv64_dgrad_fused_bf16: bad act %d", act);
    FDN_REQUIRE(!(y_mask && act == FDN_ACT_NONE), "fdn_conv64_dgrad_fused_bf16_mask: a sign mask needs act = RELU or LEAKY");
    return fdn_conv64_bf16_launch(dz, wpack, nullptr, nullptr, nullptr, dxpad, skip, y_prev, dz_prev, N, D, H, W, D + 2, H + 2,
                                  W + 2, -1, 1, act, alpha, (hipStream_t)stream, nullptr, y_mask);
}

extern "C" int fdn_conv64_dgrad_fused_bf16_multi(const uint16_t* const* dz, const uint16_t* const* wpack, int nsrc, float* dxpad,
                                                 const uint16_t* skip, const uint16_t* y_prev, const uint16_t* y_mask, int act, float alpha,
                                                 uint16_t* dz_prev, int N, int D, int H, int W, void* stream) {
    FDN_REQUIRE(dz && wpack && dxpad && dz_prev && nsrc >= 1 && nsrc <= 3, "fdn_conv64_dgrad_fused_bf16_multi: NULL argument or nsrc outside 1..3");
    FDN_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && D <= 1020 && H <= 1020 && W <= 1020, "fdn_conv64_dgrad_fused_bf16_multi: bad dims");
    FDN_REQUIRE(act >= FDN_ACT_NONE && act <= FDN_ACT_LEAKY, "fdn_conv64_dgrad_fused_bf16_multi: bad act %d", act);
    FDN_REQUIRE(!(y_mask && act == FDN_ACT_NONE), "fdn_conv64_dgrad_fused_bf16_multi: a sign mask needs act = RELU or LEAKY");
    for (int i = 0; i < nsrc; ++i) FDN_REQUIRE(dz[i] && wpack[i], "fdn_conv64_dgrad_fused_bf16_multi: NULL pointer for source %d", i);
    const FdnExtraSrcBf ex{nsrc, nsrc > 1 ? dz[1] : nullptr, nsrc > 2 ? dz[2] : nullptr, nsrc > 1 ? wpack[1] : nullptr, nsrc > 2 ? wpack[2] : nullptr};
    return fdn_conv64_bf16_launch(dz[0], wpack[0], nullptr, nullptr, nullptr, dxpad, skip, y_prev, dz_prev, N, D, H, W, D + 2, H + 2,
                                  W + 2, -1, 1, act, alpha, (hipStream_t)stream, nullptr, y_mask, nsrc > 1 ? &ex : nullptr);
}

extern "C" int fdn_conv64_dgrad_fused_bf16(const uint16_t* dz, const uint16_t* wpack, float* dxpad, const uint16_t* skip,
                                           const uint16_t* y_prev, int act, float alpha, uint16_t* dz_prev, int N, int D,
                                           int H, int W, void* stream) {
    return fdn_conv64_dgrad_fused_bf16_mask(dz, wpack, dxpad, skip, y_prev, nullptr, act, alpha, dz_prev, N, D, H, W, stream);
}

extern "C" int fdn_fold_halo_border_bf16(const float* dxpad0, const float* dxpad1, const float* dxpad2, int nsrc,
                                         const uint16_t* skip, const uint16_t* y_prev, int act, float alpha,
                                         uint16_t* dz_prev, int N, int D, int H, int W, void* stream) {
    FDN_REQUIRE(dxpad0 && dz_prev, "fdn_fold_halo_border_bf16: NULL argument");
    FDN_REQUIRE(nsrc >= 1 && nsrc <= 3 && (nsrc < 2 || dxpad1) && (nsrc < 3 || dxpad2), "fdn_fold_halo_border_bf16: bad nsrc %d", nsrc);
    FDN_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "fdn_fold_halo_border_bf16: bad dims");
    return fdn_fold_halo_border_bf16_launch(dxpad0, dxpad1, dxpad2, nsrc, skip, y_prev, act, alpha, dz_prev, N, D, H, W,
                                            (hipStream_t)stream);
}

extern "C" int fdn_conv1x1_dgrad(const float* dz, const float* w, const float* ya, const float* yb, float* dxa, float* dxb,
                                 int64_t nvox, void* stream) {
    FDN_REQUIRE(dz && w && ya && yb && dxa && dxb && nvox > 0, "fdn_conv1x1_dgrad: NULL argument or nvox<=0");
    return fdn_conv1x1_dgrad_launch<float>(dz, w, ya, yb, dxa, dxb, nvox, (hipStream_t)stream);
}

extern "C" int fdn_conv_cout1_dgrad_folded(const float* dz, const float* w, const float* y_prev, int act, float alpha,
                                           float* dz_prev, float* dbias_prev, void* workspace, size_t workspace_bytes, int N,
                                           int D, int H, int W, int lddz, int dz_coff, void* stream) {
    return fdn_conv_cout1_dgrad_folded_launch<float>(dz, w, y_prev, act, alpha, dz_prev, dbias_prev, workspace, workspace_bytes,
                                                     N, D, H, W, lddz, dz_coff, (hipStream_t)stream);
}

extern "C" int fdn_conv_cout1_dgrad_folded_mask(const float* dz, const float* w, const uint16_t* y_mask, int act, float alpha,
                                                float* dz_prev, float* dbias_prev, void* workspace, size_t workspace_bytes, int N,
                                                int D, int H, int W, int lddz, int dz_coff, void* stream) {
    FDN_REQUIRE(y_mask && (act == FDN_ACT_RELU || act == FDN_ACT_LEAKY), "fdn_conv_cout1_dgrad_folded_mask: a sign mask and act = RELU or LEAKY");
    return fdn_conv_cout1_dgrad_folded_launch<float>(dz, w, nullptr, act, alpha, dz_prev, dbias_prev, workspace, workspace_bytes,
                                                     N, D, H, W, lddz, dz_coff, (hipStream_t)stream, y_mask);
}

// ---- bf16 activation path: the thin layers and the generic conv entry points ----
extern "C" int fdn_conv1x1_dgrad_bf16(const uint16_t* dz, const float* w, const uint16_t* ya, const uint16_t* yb,
                                      uint16_t* dxa, uint16_t* dxb, int64_t nvox, void* stream) {
    FDN_REQUIRE(dz && w && ya && yb && dxa && dxb && nvox > 0, "fdn_conv1x1_dgrad_bf16: NULL argument or nvox<=0");
    return fdn_conv1x1_dgrad_launch<uint16_t>(dz, w, ya, yb, dxa, dxb, nvox, (hipStream_t)stream);
}

extern "C" size_t fdn_conv3d_wgrad_bf16_batch_workspace_bytes(int n_layers, int N, int D, int H, int W) {
    if (n_layers <= 0 || N <= 0 || D <= 0 || H <= 0 || W <= 0) return 0;
    const size_t one = fdn_conv3d_wgrad_bf16_workspace_bytes(N, D, H, W, 64, 64, 3);
    const size_t all = fdn_wgrad64_bf16_batch_ok(n_layers, N, D, H, W) ? fdn_wgrad64_bf16_batch_workspace_bytes(n_layers, N, D, H, W) : 0;
    return all > one ? all : one;
}
extern "C" int fdn_conv3d_wgrad_bf16_batch(const uint16_t* const* x, const uint16_t* const* dz, float* const* dw, float* const* dbias,
                                           int n_layers, void* workspace, size_t workspace_bytes, int N, int D, int H, int W, void* stream) {
    FDN_REQUIRE(x && dz && dw && n_layers > 0, "fdn_conv3d_wgrad_bf16_batch: NULL table or n_layers <= 0");
    FDN_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "fdn_conv3d_wgrad_bf16_batch: bad dims");
    const size_t need = fdn_conv3d_wgrad_bf16_batch_workspace_bytes(n_layers, N, D, H, W);
    if (workspace_bytes < need || !workspace) {
        fdn_set_error("fdn_conv3d_wgrad_bf16_batch: workspace %zu < %zu bytes", workspace_bytes, need);
        return FDN_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    if (fdn_wgrad64_bf16_batch_ok(n_layers, N, D, H, W)) {
        if (int rc = fdn_wgrad64_bf16_batch_launch(x, dz, dw, n_layers, workspace, workspace_bytes, N, D, H, W, s)) return rc;
    } else {
        for (int i = 0; i < n_layers; ++i) {
            FDN_REQUIRE(x[i] && dz[i] && dw[i], "fdn_conv3d_wgrad_bf16_batch: NULL pointer for layer %d", i);
            if (int rc = fdn_wgrad64_bf16_launch(x[i], dz[i], dw[i], workspace, workspace_bytes, N, D, H, W, s)) return rc;
        }
    }
    if (dbias)                           // (stream-ordered behind the reduction: the workspace is free again)
        for (int i = 0; i < n_layers; ++i)
            if (dbias[i])
                if (int rc = fdn_bias_grad_launch<uint16_t>(dz[i], dbias[i], workspace, workspace_bytes, (int64_t)N * D * H * W, 64, 64, 0, s)) return rc;
    return FDN_OK;
}

extern "C" int fdn_conv_cout1_dgrad_folded_bf16_mask(const float* dz, const float* w, const uint16_t* y_prev, const uint16_t* y_mask,
                                                     int act, float alpha, uint16_t* dz_prev, float* dbias_prev, void* workspace,
                                                     size_t workspace_bytes, int N, int D, int H, int W, int lddz, int dz_coff,
                                                     void* stream) {
    FDN_REQUIRE(!(y_mask && act == FDN_ACT_NONE), "fdn_conv_cout1_dgrad_folded_bf16_mask: a sign mask needs act = RELU or LEAKY");
    return fdn_conv_cout1_dgrad_folded_launch<uint16_t>(dz, w, y_prev, act, alpha, dz_prev, dbias_prev, workspace,
                                                        workspace_bytes, N, D, H, W, lddz, dz_coff, (hipStream_t)stream, y_mask);
}

extern "C" int fdn_conv_cout1_dgrad_folded_bf16(const float* dz, const float* w, const uint16_t* y_prev, int act, float alpha,
                                                uint16_t* dz_prev, float* dbias_prev, void* workspace, size_t workspace_bytes,
                                                int N, int D, int H, int W, int lddz, int dz_coff, void* stream) {
    return fdn_conv_cout1_dgrad_folded_bf16_mask(dz, w, y_prev, nullptr, act, alpha, dz_prev, dbias_prev, workspace, workspace_bytes, N, D, H, W,
                                                 lddz, dz_coff, stream);
}

extern "C" int fdn_conv3d_fwd_bf16(const uint16_t* x, const uint16_t* x2, const float* w, const uint16_t* wpack,
                                   const float* bias, const uint16_t* residual, void* y, int N, int D, int H, int W, int Cin,
                                   int Cout, int K, int ldy, int y_coff, int act, float alpha, void* stream) {
    FDN_REQUIRE(x && y, "fdn_conv3d_fwd_bf16: x/y is NULL");
    FDN_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "fdn_conv3d_fwd_bf16: bad dims N=%d D=%d H=%d W=%d", N, D, H, W);
    FDN_REQUIRE(act >= FDN_ACT_NONE && act <= FDN_ACT_LEAKY, "fdn_conv3d_fwd_bf16: bad act %d", act);
    hipStream_t s = (hipStream_t)stream;
    if (Cin == 64 && Cout == 64 && K == 3) {
        FDN_REQUIRE(ldy == 64 && y_coff == 0, "fdn_conv3d_fwd_bf16: 64->64 path writes dense rows");
        return fdn_conv64_fwd_bf16(x, wpack, bias, residual, (uint16_t*)y, N, D, H, W, act, alpha, stream);
    }
    FDN_REQUIRE(residual == nullptr, "fdn_conv3d_fwd_bf16: residual only on the 64->64 path");
    if (Cin == 3 && Cout == 64 && K == 3) {
        FDN_REQUIRE(w && ldy == 64 && y_coff == 0, "fdn_conv3d_fwd_bf16(3->64): needs w, dense output");
        return fdn_conv_cin3_fwd_launch<uint16_t>(x, w, bias, (uint16_t*)y, N, D, H, W, act, alpha, s);
    }
    if (Cin == 64 && Cout == 1 && K == 3) {
        FDN_REQUIRE(w && ldy >= 1 && y_coff >= 0 && y_coff < ldy, "fdn_conv3d_fwd_bf16(64->1): bad w/ldy/y_coff");
        return fdn_conv_cout1_fwd_launch<uint16_t>(x, w, bias, (float*)y, N, D, H, W, ldy, y_coff, act, alpha, s);
    }
    if (Cin == 128 && Cout == 64 && K == 1) {
        FDN_REQUIRE(w && x2 && ldy == 64 && y_coff == 0, "fdn_conv3d_fwd_bf16(1x1 128->64): needs w, x2, dense output");
        return fdn_conv1x1_fwd_launch<uint16_t>(x, x2, w, bias, (uint16_t*)y, (int64_t)N * D * H * W, act, alpha, s);
    }
    fdn_set_error("fdn_conv3d_fwd_bf16: unsupported (Cin=%d,Cout=%d,K=%d)", Cin, Cout, K);
    return FDN_ERR_UNSUPPORTED;
}

extern "C" size_t fdn_conv3d_wgrad_bf16_workspace_bytes(int N, int D, int H, int W, int Cin, int Cout, int K) {
    if (Cin == 64 && Cout == 64 && K == 3) return fdn_wgrad64_bf16_workspace_bytes(N, D, H, W);
    return fdn_small_wgrad_workspace_bytes(Cin, Cout, K);
}

extern "C" int fdn_conv3d_wgrad_bf16(const uint16_t* x, const uint16_t* x2, const void* dz, float* dw, float* dbias,
                                     void* workspace, size_t workspace_bytes, int N, int D, int H, int W, int Cin, int Cout,
                                     int K, int lddz, int dz_coff, void* stream) {
    FDN_REQUIRE(x && dz && dw, "fdn_conv3d_wgrad_bf16: x/dz/dw is NULL");
    FDN_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "fdn_conv3d_wgrad_bf16: bad dims");
    const size_t need = fdn_conv3d_wgrad_bf16_workspace_bytes(N, D, H, W, Cin, Cout, K);
    if (workspace_bytes < need || (need && !workspace)) {
        fdn_set_error("fdn_conv3d_wgrad_bf16: workspace %zu < %zu bytes", workspace_bytes, need);
        return FDN_ERR_WORKSPACE;
    }
    hipStream_t s = (hipStream_t)stream;
    const int64_t nvox = (int64_t)N * D * H * W;
    const uint16_t* dzb = (const uint16_t*)dz;
    int rc;
    if (Cin == 64 && Cout == 64 && K == 3) {
        FDN_REQUIRE(lddz == 64 && dz_coff == 0, "fdn_conv3d_wgrad_bf16: 64->64 path reads dense dz rows");
        rc = fdn_wgrad64_bf16_launch(x, dzb, dw, workspace, workspace_bytes, N, D, H, W, s);
    } else if (Cin == 3 && Cout == 64 && K == 3) {
        FDN_REQUIRE(lddz == 64 && dz_coff == 0, "fdn_conv3d_wgrad_bf16(3->64): dense dz rows");
        rc = fdn_wgrad_cin3_launch<uint16_t>(x, dzb, dw, workspace, workspace_bytes, N, D, H, W, s);
    } else if (Cin == 64 && Cout == 1 && K == 3) {      // the head's dz is the fp32 prediction gradient
        rc = fdn_wgrad_cout1_launch<uint16_t>(x, (const float*)dz, dw, workspace, workspace_bytes, N, D, H, W, lddz, dz_coff, s);
        if (rc != FDN_OK) return rc;
        if (dbias) return fdn_bias_grad_launch<float>((const float*)dz, dbias, workspace, workspace_bytes, nvox, 1, lddz, dz_coff, s);
        return FDN_OK;
    } else if (Cin == 128 && Cout == 64 && K == 1) {
        FDN_REQUIRE(x2 && lddz == 64 && dz_coff == 0, "fdn_conv3d_wgrad_bf16(1x1): needs x2, dense dz rows");
        rc = fdn_wgrad_1x1_launch<uint16_t>(x, x2, dzb, dw, workspace, workspace_bytes, nvox, s);
    } else {
        fdn_set_error("fdn_conv3d_wgrad_bf16: unsupported (Cin=%d,Cout=%d,K=%d)", Cin, Cout, K);
        return FDN_ERR_UNSUPPORTED;
    }
    if (rc != FDN_OK) return rc;
    if (dbias) return fdn_bias_grad_launch<uint16_t>(dzb, dbias, workspace, workspace_bytes, nvox, Cout, lddz, dz_coff, s);
    return FDN_OK;
}

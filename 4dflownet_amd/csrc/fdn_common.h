// Shared helpers for the lib4dflow_hip.so translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "fdn.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

typedef unsigned fdn_u32x2 __attribute__((ext_vector_type(2)));

// Activation storage: float, or bfloat16 held as its uint16_t bit pattern (the bf16 path of BASELINE.json configs[3]).
// Arithmetic is always fp32; a bf16 store rounds to nearest even.
__device__ __forceinline__ float fdn_ld1(const float* p) { return *p; }
__device__ __forceinline__ float fdn_ld1(const uint16_t* p) { return __builtin_bit_cast(float, (unsigned)*p << 16); }
__device__ __forceinline__ void fdn_st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void fdn_st1(uint16_t* p, float v) { const __bf16 b = (__bf16)v; *p = __builtin_bit_cast(uint16_t, b); }
__device__ __forceinline__ f32x4 fdn_ld4(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ f32x4 fdn_ld4(const uint16_t* p) {
    const fdn_u32x2 r = *(const fdn_u32x2*)p;
    return (f32x4){__builtin_bit_cast(float, r.x << 16), __builtin_bit_cast(float, r.x & 0xffff0000u),
                   __builtin_bit_cast(float, r.y << 16), __builtin_bit_cast(float, r.y & 0xffff0000u)};
}
// raw (still packed) 4-element loads for register prefetch rings: convert at the point of use, not at the load
__device__ __forceinline__ f32x4 fdn_ld4_raw(const float* p) { return *(const f32x4*)p; }
__device__ __forceinline__ fdn_u32x2 fdn_ld4_raw(const uint16_t* p) { return *(const fdn_u32x2*)p; }
__device__ __forceinline__ f32x4 fdn_cvt4(f32x4 r) { return r; }
__device__ __forceinline__ f32x4 fdn_cvt4(fdn_u32x2 r) {
    return (f32x4){__builtin_bit_cast(float, r.x << 16), __builtin_bit_cast(float, r.x & 0xffff0000u),
                   __builtin_bit_cast(float, r.y << 16), __builtin_bit_cast(float, r.y & 0xffff0000u)};
}
__device__ __forceinline__ void fdn_st4(float* p, f32x4 v) { *(f32x4*)p = v; }
__device__ __forceinline__ void fdn_st4(uint16_t* p, f32x4 v) {
    typedef __bf16 bf4 __attribute__((ext_vector_type(4)));
    const bf4 b = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    *(fdn_u32x2*)p = __builtin_bit_cast(fdn_u32x2, b);
}

// one 16-B vector = FdnVec<T>::E consecutive elements (4 fp32 / 8 bf16) as fp32 values
template <typename T> struct FdnVec;
template <> struct FdnVec<float> {
    static constexpr int E = 4;
    __device__ __forceinline__ static void ld(const float* p, float (&v)[4]) {
        const f32x4 t = *(const f32x4*)p; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    __device__ __forceinline__ static void st(float* p, const float (&v)[4]) { *(f32x4*)p = (f32x4){v[0], v[1], v[2], v[3]}; }
};
template <> struct FdnVec<uint16_t> {
    static constexpr int E = 8;
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    __device__ __forceinline__ static void ld(const uint16_t* p, float (&v)[8]) {
        const u4 r = *(const u4*)p;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __builtin_bit_cast(float, r[i] << 16);
            v[2 * i + 1] = __builtin_bit_cast(float, r[i] & 0xffff0000u);
        }
    }
    __device__ __forceinline__ static void st(uint16_t* p, const float (&v)[8]) {
        typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
        bf8 b;
#pragma unroll
        for (int i = 0; i < 8; ++i) b[i] = (__bf16)v[i];
        *(u4*)p = __builtin_bit_cast(u4, b);
    }
};

// fp32 -> three bf16 pieces, v = hi + mid + lo EXACTLY (each piece the round-to-nearest-even bf16 of what the pieces before it leave: the
// residual of a round-to-nearest is an fp32 number, the last one has at most 8 significant bits).  A bf16 x bf16 product is exact in fp32,
// so the six cross terms hi.hi, mid.hi, lo.hi, hi.mid, mid.mid, hi.lo accumulated in fp32 on the bf16 matrix pipe reproduce the fp32
// product to 2^-25 |u||v| (the three dropped terms) -- under the half ulp an fp32 multiply rounds away -- at 6/16 of the fp32-MFMA time.
typedef __bf16 fdn_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 fdn_bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned fdn_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned fdn_pk_bf16(float a, float b) {       // two round-to-nearest-even bf16 in one dword (v_cvt_pk_bf16_f32)
    const fdn_bf16x2 t = {(__bf16)a, (__bf16)b};
    return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ f32x4 fdn_unpk_bf16(fdn_u32x2 h) {
    return (f32x4){__builtin_bit_cast(float, h.x << 16), __builtin_bit_cast(float, h.x & 0xffff0000u),
                   __builtin_bit_cast(float, h.y << 16), __builtin_bit_cast(float, h.y & 0xffff0000u)};
}
__device__ __forceinline__ void fdn_split3(const f32x4 v, fdn_u32x2& hi, fdn_u32x2& mid, fdn_u32x2& lo) {
    hi = (fdn_u32x2){fdn_pk_bf16(v.x, v.y), fdn_pk_bf16(v.z, v.w)};
    f32x4 r = v - fdn_unpk_bf16(hi);
    mid = (fdn_u32x2){fdn_pk_bf16(r.x, r.y), fdn_pk_bf16(r.z, r.w)};
    r = r - fdn_unpk_bf16(mid);
    lo = (fdn_u32x2){fdn_pk_bf16(r.x, r.y), fdn_pk_bf16(r.z, r.w)};
}

// Direct-to-LDS load (gfx950 `buffer_load_dwordx4 ... lds`): every lane reads 16 B at rsrc + voff + soff and the wave writes the
// 64 x 16 B to LDS at lds + lane * 16 (lds is wave-uniform: it travels in M0).  Out-of-range lanes store zeros.  The compiler counts
// it in vmcnt like any other vector-memory load.  (A __device__ function, not a lambda inside the kernel: the builtin does not exist
// in the host pass, and a lambda that uses it silently takes the kernel's host stub with it.)
__device__ __forceinline__ void fdn_lds_dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds, unsigned voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

// The same instruction issued OUTSIDE the compiler's bookkeeping (inline asm).  hipcc tracks a builtin LDS-DMA as a pending LDS
// write and puts s_waitcnt vmcnt(0) in front of the next LDS read it cannot prove disjoint (every ds_read_b64_tr_b16, for one):
// a ring of tiles requested two iterations ahead then waits for its newest request before the first operand read of every
// iteration.  With this form the CALLER owns the ordering: s_waitcnt vmcnt(n) by hand (vector-memory loads complete in order)
// and a barrier before another wave reads the data.  rsrc = fdn_raw_rsrc(...), lds_addr = fdn_lds_addr(pointer), wave-uniform.
typedef int fdn_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ fdn_i32x4 fdn_raw_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;
    return (fdn_i32x4){(int)(unsigned)a, (int)(unsigned)((a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ unsigned fdn_lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ void fdn_lds_dma16_untracked(fdn_i32x4 rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                 :: "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory", "m0");
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a fence + s_barrier and hipcc drains EVERY counter in front of
// it -- s_waitcnt vmcnt(0) included, so global loads a software pipeline issued a few instructions earlier (register prefetches of
// a later tile, LDS-DMA into an area only the issuing wave reads back) stall all waves of the workgroup for a full memory latency.
// Here only the LDS queue is drained; vector-memory loads stay in flight (the compiler's own waitcnt bookkeeping for their
// destination registers is unaffected).  Use it ONLY where nothing another wave reads after the barrier comes from a pending
// vector-memory operation.
__device__ __forceinline__ void fdn_barrier_lds() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

void fdn_set_error(const char* fmt, ...);

// hipFuncAttributeMaxDynamicSharedMemorySize, set once per (device, kernel) under a lock: the C-ABI is re-entrant per
// (device, stream) from any host thread (api.hip).  Returns FDN_OK or FDN_ERR_HIP (message set).
int fdn_func_max_lds(const void* fn, int bytes, const char* who);

// Variant-forcing / ablation hooks (fdn_debug_*) exist only in the TEST build of the library (lib4dflow_hip_test.so,
// compiled with -DFDN_TEST_HOOKS by 4dflownet_amd/build.py); the product library has no mutable global state.
#ifdef FDN_TEST_HOOKS
#define FDN_HOOK_VAR(type, name, init) static type name = init
#define FDN_DBG_BITS(args) ((args).dbg)          // kernel-side ablation bits: a runtime argument in the test build ...
#else
#define FDN_HOOK_VAR(type, name, init) static constexpr type name = init
#define FDN_DBG_BITS(args) 0                     // ... and compiled out of the product library
#endif

#define FDN_CHECK_LAUNCH(name)                                                         \
    do {                                                                               \
        hipError_t e_ = hipGetLastError();                                             \
        if (e_ != hipSuccess) {                                                        \
            fdn_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));      \
            return FDN_ERR_HIP;                                                        \
        }                                                                              \
    } while (0)

#define FDN_REQUIRE(cond, ...)                                                         \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            fdn_set_error(__VA_ARGS__);                                                \
            return FDN_ERR_BAD_ARG;                                                    \
        }                                                                              \
    } while (0)

// floor(r/d) for 0 <= r < 1024 and 1 <= d <= 1024, with magic = ceil(2^20/d).
__host__ __device__ inline unsigned fdn_magic20(unsigned d) { return ((1u << 20) + d - 1) / d; }
__device__ __forceinline__ int fdn_div20(int r, unsigned magic) { return (int)(((unsigned)r * magic) >> 20); }
// floor(n/d) for n*d < 2^40 with M = ceil(2^40/d) = hi*2^32 + lo:  (n*M) >> 40 = (n*hi + mulhi(n, lo)) >> 8.
// For wave-uniform n this is four scalar instructions.
inline void fdn_magic40(unsigned d, unsigned* hi, unsigned* lo) {
    const unsigned long long m = ((1ull << 40) + d - 1) / d;
    *hi = (unsigned)(m >> 32); *lo = (unsigned)m;
}
__device__ __forceinline__ int fdn_udiv40(int n, unsigned hi, unsigned lo) {
    return (int)(((unsigned)n * hi + __umulhi((unsigned)n, lo)) >> 8);
}

__device__ __forceinline__ float fdn_act(float z, int act, float alpha) {
    if (act == FDN_ACT_RELU) return z > 0.f ? z : 0.f;
    if (act == FDN_ACT_LEAKY) return z > 0.f ? z : alpha * z;
    return z;
}
// act'(z) recovered from the stored output y = act(z)  (y > 0 <=> z > 0 for relu / leaky-relu)
__device__ __forceinline__ float fdn_act_grad(float y, int act, float alpha) {
    if (act == FDN_ACT_RELU) return y > 0.f ? 1.f : 0.f;
    if (act == FDN_ACT_LEAKY) return y > 0.f ? 1.f : alpha;
    return 1.f;
}

// Tile planner shared by host code of the MFMA kernels: choose (td,th,tw) with td*th*tw <= max_vox and
// halo rows <= max_rows minimising the number of workgroup-rounds over `ncu` CUs.
struct FdnTile { int td, th, tw, ntd, nth, ntw; };
FdnTile fdn_plan_tile(int N, int OD, int OH, int OW, int max_vox, int max_halo_rows, int halo_d);

// entry points implemented in the per-kernel translation units
int fdn_conv64_launch(const float* x, const float* wpack, const float* bias, const float* residual, float* y,
                      int N, int ID, int IH, int IW, int OD, int OH, int OW, int off, int zero_mode, int act,
                      float alpha, hipStream_t s, int algo = FDN_ALGO_AUTO);
// Further sources of a multi-source fused dgrad (fdn_conv64_dgrad_fused_multi: dz_prev = fold(sum_s conv_T(dz_s, W_s))): rows and the byte
// distance of the source's PACK from source 0's (>= 0: the caller orders the sources by pack address; every stream of a pack lies at
// the same offset inside it, so one distance serves the 2-D and the 1-D kernel)
struct FdnExtraSrc { int nsrc; const float* x1; const float* x2; int wd1, wd2; };
int fdn_conv64_launch_ex(const float* x, const float* wpack, const float* bias, const float* residual, float* y,
                         const float* fskip, const float* fy, float* fout, int N, int ID, int IH, int IW, int OD, int OH,
                         int OW, int off, int zero_mode, int act, float alpha, hipStream_t s, int parts = 3,
                         int algo = FDN_ALGO_AUTO, unsigned* probe = nullptr, uint16_t* ymask = nullptr, const uint16_t* fmask = nullptr,
                         const FdnExtraSrc* extra = nullptr);
// Winograd F(4,3)-along-W variant of the 64->64 conv (conv64_wino.hip): one output box with all 27 taps
// output box + its non-zero (kd, kh) tap ranges.  wface = 1: the pair of w faces of a fused dgrad's shell (box = the (d,h) range of the
// padded grid, ow = 0, ew = 4: one "group" per (d,h) position; see conv64_wino.hip)
struct FdnWino2dPrepared;
struct FdnWinoBox { int od, oh, ow, ed, eh, ew, ta0, ta1, tb0, tb1, wface; };
bool fdn_conv64_wino_ok(int ebd, int ebh, int ebw);
int fdn_conv64_wino_launch_boxes(const float* x, const float* upack, const float* bias, const float* residual, float* y,
                                 const float* fskip, const float* fy, float* fout, int N, int ID, int IH, int IW, int OD, int OH,
                                 int OW, const FdnWinoBox* boxes, int nbox, int off, int zero_mode, int act, float alpha,
                                 hipStream_t s, const struct FdnWino2dPrepared* inner = nullptr, const FdnExtraSrc* extra = nullptr);
int fdn_conv64_wino_launch(const float* x, const float* upack, const float* bias, const float* residual, float* y,
                           const float* fskip, const float* fy, float* fout, int N, int ID, int IH, int IW, int OD, int OH,
                           int OW, int obd, int obh, int obw, int ebd, int ebh, int ebw, int off, int zero_mode, int act,
                           float alpha, hipStream_t s);
int fdn_pack_conv64_wino_launch(const float* w, float* uf, float* ud, hipStream_t s);
// 2-D Winograd variant (conv64_wino2d.hip): F(2,3) along H x F(4,3) along W; one output box with all 27 taps
// (hm = output rows per cell: 2 = F(2,3) along H, stream at pack + 81*4096; 4 = F(4,3) along H, stream at pack + 153*4096)
bool fdn_conv64_wino2d_ok(int ebd, int ebh, int ebw, int ID, int IH, int IW, int hm);
// A planned, not yet launched 2-D launch (the kernel's argument block, opaque outside conv64_wino2d_kernel.h): fdn_conv64_wino_launch_boxes
// takes one as `inner` and issues it together with its own regions as ONE launch (conv64_wino2d_shell_kernel: the fused dgrad).
struct FdnWino2dPrepared { alignas(8) unsigned char args[384]; int blocks; int lds; };
int fdn_conv64_wino2d_prepare(const float* x, const float* upack2, const float* bias, const float* residual, float* y,
                              const float* fskip, const float* fy, float* fout, int N, int ID, int IH, int IW, int OD, int OH,
                              int OW, int obd, int obh, int obw, int ebd, int ebh, int ebw, int off, int zero_mode, int act,
                              float alpha, int hm, FdnWino2dPrepared* out, uint16_t* ymask = nullptr, const uint16_t* fmask = nullptr,
                              const FdnExtraSrc* extra = nullptr);
int fdn_conv64_wino2d_launch(const float* x, const float* upack2, const float* bias, const float* residual, float* y,
                             const float* fskip, const float* fy, float* fout, int N, int ID, int IH, int IW, int OD, int OH,
                             int OW, int obd, int obh, int obw, int ebd, int ebh, int ebw, int off, int zero_mode, int act,
                             float alpha, int hm, hipStream_t s, uint16_t* ymask = nullptr, const uint16_t* fmask = nullptr);
int fdn_pack_conv64_wino2d_launch(const float* w, float* uf, float* ud, hipStream_t s);
int fdn_fold_halo_border_launch(const float* s0, const float* s1, const float* s2, int nsrc, const float* skip,
                                const float* yprev, int act, float alpha, float* out, int N, int D, int H, int W,
                                hipStream_t s);
int fdn_wgrad64_launch(const float* x, const float* dz, float* dw, void* ws, size_t ws_bytes, int N, int D, int H,
                       int W, hipStream_t s);
size_t fdn_wgrad64_workspace_bytes(int N, int D, int H, int W);
int fdn_wgrad64_wino_launch(const float* x, const float* dz, float* dw, void* ws, size_t ws_bytes, int N, int D, int H,
                            int W, hipStream_t s, int algo = FDN_ALGO_AUTO);
size_t fdn_wgrad64_wino_workspace_bytes(int N, int D, int H, int W);
// several 64->64 layers of one grid in ONE launch (wgrad64_wino.hip); _ok: applicable (else the caller loops over fdn_wgrad64_wino_launch)
bool fdn_wgrad64_wino_batch_ok(int n_layers, int D, int algo);
size_t fdn_wgrad64_wino_batch_workspace_bytes(int n_layers, int N, int D, int H, int W);
int fdn_wgrad64_wino_batch_launch(const float* const* x, const float* const* dz, float* const* dw, int n_layers, void* ws, size_t ws_bytes,
                                  int N, int D, int H, int W, hipStream_t s);
int fdn_wgrad64_reduce_launch(const float* partial, float* dw, int S, hipStream_t s);

// bf16 activation path (conv64_bf16.hip)
// further sources of a multi-source fused dgrad in bf16 mode (fdn_conv64_dgrad_fused_bf16_multi): rows and packs of sources 1, 2
struct FdnExtraSrcBf { int nsrc; const uint16_t* x1; const uint16_t* x2; const uint16_t* wp1; const uint16_t* wp2; };
int fdn_conv64_bf16_launch(const uint16_t* x, const uint16_t* wpack, const float* bias, const uint16_t* residual, uint16_t* y,
                           float* ypad, const uint16_t* fskip, const uint16_t* fy, uint16_t* fout, int N, int ID, int IH,
                           int IW, int OD, int OH, int OW, int off, int zero_mode, int act, float alpha, hipStream_t s,
                           uint16_t* ymask = nullptr, const uint16_t* fmask = nullptr, const struct FdnExtraSrcBf* extra = nullptr);
int fdn_fold_halo_border_bf16_launch(const float* s0, const float* s1, const float* s2, int nsrc, const uint16_t* skip,
                                     const uint16_t* yprev, int act, float alpha, uint16_t* out, int N, int D, int H, int W,
                                     hipStream_t s);
size_t fdn_wgrad64_bf16_workspace_bytes(int N, int D, int H, int W);
int fdn_wgrad64_bf16_launch(const uint16_t* x, const uint16_t* dz, float* dw, void* ws, size_t ws_bytes, int N, int D, int H,
                            int W, hipStream_t s);
// several 64->64 layers of one grid per launch (wgrad64_bf16.hip)
bool fdn_wgrad64_bf16_batch_ok(int n_layers, int N, int D, int H, int W);
size_t fdn_wgrad64_bf16_batch_workspace_bytes(int n_layers, int N, int D, int H, int W);
int fdn_wgrad64_bf16_batch_launch(const uint16_t* const* x, const uint16_t* const* dz, float* const* dw, int n_layers, void* ws, size_t ws_bytes,
                                  int N, int D, int H, int W, hipStream_t s);

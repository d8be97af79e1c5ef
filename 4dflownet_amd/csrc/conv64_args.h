// Argument block of the 64->64 3x3x3 conv kernel (conv64_mfma.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// One launch covers up to 7 REGIONS of the output grid.  A region is an output box [ob, ob+eb) with the tap range
// [t0, t1] per dimension that can be non-zero for it, tiled with its own tile shape; workgroups
// [first_block, first_block + N*ntd*nth*ntw) belong to it.
//   forward            : 1 region  = the whole grid, all 27 taps.
//   fused dgrad        : 7 regions = the inner D^3 box with all taps + six 1-voxel shell slabs of the padded grid that only
//                        have 9 non-zero taps each (the other taps would read rows that are zero by the boundary rule)
//                        and need a 1-deep staging box in the slab's normal direction.  The slab tiles cost a third of
//                        an inner tile and are dispatched last, where they also fill the launch's tail.
struct Conv64Region {
    int first_block;
    int obd, obh, obw, ebd, ebh, ebw;
    int ta0, ta1, tb0, tb1, tc0, tc1;
    int td, th, tw, ntd, nth, ntw;
    int hh, hw;                 // staged box dims: th + (tb1-tb0), tw + (tc1-tc0)   (depth: td + (ta1-ta0))
    int rows;                   // staged rows
    unsigned mg_hhhw, mg_hw;    // magic divisors for staged-row decomposition
    unsigned mg_thtw, mg_tw;    // ... for tile-row -> (d,h,w) (fp32 kernel)
    unsigned mg_tpn_hi, mg_tpn_lo, mg_thw_hi, mg_thw_lo, mg_ntw_hi, mg_ntw_lo;   // 40-bit magics: block -> (n, td, th, tw)
    // bf16 kernel only: LDS image geometry.  Staged voxel (zd,zh,zw) lives in LDS row (zd*hh + zh)*hs + zw (hs >= hw,
    // hs = 4 mod 8 so that the two 4-row runs a ds_read_b128 lane group takes from rows zh, zh+1 fall on the same bank
    // rows), its two 16-B chunks swapped when ((zh >> swz_hs) + ((zw >> 2) & swz_wm)) & 1.
    int hs, lrows, lrows_p;     // lrows_p = lrows rounded up to 32 (one LDS buffer = whole 1-KB pieces)
    int swz_hs, swz_wm;
    unsigned mg_hhhs, mg_hs;    // magic divisors for LDS-row -> (zd, zh, zw') (direct-to-LDS staging of the FAST kernel)
    // bf16 kernel, general variant only: 1 = the region's first two axes are (h, d) instead of (d, h) -- every d / h field above is
    // then in KERNEL order (ob"d" = first h, t"a" = height taps, ...).  Used for the two d faces of a fused dgrad's shell: their
    // single tap is the depth tap, and the kernel slides its 8 accumulator planes along the first axis, which must have 3 taps.
    int swap_dh;
};

struct Conv64Args {
    const float* x;
    const float* wp;
    const float* bias;
    const float* res;
    float* y;
    // fused fold (dgrad mode only): interior voxels are finished in the epilogue
    const float* fskip;     // gradient to add (N,ID,IH,IW,64) or null
    const float* fy;        // producer output for act' or null
    float* fout;            // dz_prev (N,ID,IH,IW,64) or null (= plain padded-grid dgrad)
    int N, ID, IH, IW, OD, OH, OW;
    int off, zero_mode;
    int act;
    float alpha;
    int dbg;                // ablation bits (bench only): 1 = B stream stride 0, 4 = no staging loads, 8 = no epilogue
    int nreg;
    Conv64Region reg[7];
};

// bf16-activation variant (conv64_bf16.hip): activations / gradients are bf16 NDHWC, accumulation fp32.
struct Conv64BfArgs {
    const uint16_t* x;      // (N,ID,IH,IW,64) bf16
    const uint16_t* wp;     // packed bf16 operand stream (fdn_pack_conv64_weights_bf16)
    const float* bias;
    const uint16_t* res;    // forward: residual (N,OD,OH,OW,64) bf16 or null
    uint16_t* y;            // forward output bf16
    float* ypad;            // dgrad: fp32 padded-grid scratch (only rows that still need the border fold are written)
    const uint16_t* fskip;  // fused fold (dgrad): gradient to add, producer output for act', finished dz_prev
    const uint16_t* fy;
    uint16_t* fout;
    // sign masks (round 5): 64 bits per voxel, bit c = (the stored bf16 y[c] > 0), as four uint16_t words [voxel][cout / 16].  The forward
    // writes one beside y (ymask, or null); the fused dgrad reads it INSTEAD of y for act' (fmask, or null: then fy): an eighth of a
    // 16-B load per lane and plane instead of two 16-B loads -- at (4,128^3) an epilogue operand costs what streaming its 1.07 GB costs
    uint16_t* ymask;
    const uint16_t* fmask;
    // multi-source fused dgrad (round 6; the fp32 twin is Wino2Args'): dz_prev = fold(sum_s conv_T(x_s, W_s)) for up to three layers that
    // share their input -- the slice loop of the one-launch fused kernel runs over 2 (MODE 2) resp. 4 (MODE 0) slices x nsrc sources
    const uint16_t* x1;
    const uint16_t* x2;
    const uint16_t* wp1;
    const uint16_t* wp2;
    int nsrc;
    int N, ID, IH, IW, OD, OH, OW;
    int off, zero_mode;
    int act;
    float alpha;
    int dbg;
    int nreg;
    Conv64Region reg[7];
};

// Argument block of the 64->64 3x3x3 conv kernel (conv64_mfma.hip).
#pragma once
#include <hip/hip_runtime.h>

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

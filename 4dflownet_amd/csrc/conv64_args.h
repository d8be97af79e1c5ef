// Argument block of the 64->64 3x3x3 conv kernel (conv64_mfma.hip).
#pragma once
#include <hip/hip_runtime.h>

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
    int td, th, tw, ntd, nth, ntw;
    int hh, hw;                 // halo dims th+2, tw+2 (hd = td+2)
    int rows;                   // hd*hh*hw
    unsigned mg_hhhw, mg_hw;    // magic divisors for halo-row decomposition
    int act;
    float alpha;
    int dbg;                    // ablation bits (bench only): 1 = B stream stride 0, 4 = no staging loads, 8 = no epilogue
};

// Internal shared declarations for the spkm HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SPKM_WAVE 64

// Row-broadcast inside a 16-lane DPP row: every lane of the row receives lane
// N's value (gfx90a+ `row_newbcast:N`, DPP ctrl 0x150+N).  All lanes must be
// active.  The compiler folds this into the consuming VOP2 where it can.
template <int N>
__device__ __forceinline__ int row_bcast_i32(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, 0x150 + N, 0xf, 0xf, false);
}

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

// Launch-geometry record for one workgroup of the tiled assignment kernel.
struct spkm_blockmap {
    int tile;     // centroid tile (k-tile) this workgroup keeps in LDS, -1 = idle
    int stream;   // first chunk this workgroup processes
    int nstreams; // chunk stride
    int pad;
};

#define SCREEN_KT 32 // centroids per tile of the f32 screen (screen.hip, screen_quad.hip)

#include "policy.h" // quad_split / quad_split_late (the compile-time splits of the two-phase screen) live with the policy

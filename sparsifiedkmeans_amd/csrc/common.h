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

// Two-phase forms of the 4-lanes-per-point screen (screen_quad.hip, TWO): the first A = quad_split(NR) rounds for all
// centroids, the rest only for each tile's leader (or, hinted, for all again when a step's points do not clear their
// hints).  The split is a compile-time constant: with a run-time split every round sits behind its own branch and the
// finish's LDS reads are waited for one by one.
// Late split of the HINTED form (columns of >= 37 entries): half of the rounds.  In the first iterations of a run the
// hints are loose (the own centroid has just moved a long way) and the competition's partial sums clear them only after
// about half of the rounds; measured on the headline run (s = 51, 13 rounds) a split at 7 is best in iterations 2-4
// (28.4 / 27.9 / 26.9 ms against 33.9 / 32.6 / 29.7 at 3), the early one from the fifth on.  The host picks per call from
// the early-finish count of the previous call (policy.h).
__host__ __device__ constexpr int quad_split_late(int nr) { return nr >= 10 ? (nr + 1) / 2 : 0; } // 0: none
__host__ __device__ constexpr int quad_split(int nr) { return nr >= 3 ? ((nr + 2) / 4 > 2 ? (nr + 2) / 4 : 2) : nr; }

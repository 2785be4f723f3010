// Micro-benchmark (GPU box): the inner rounds of the 4-lanes-per-point f32 screen, broadcast form (quad_round: every
// lane keeps a quarter of its point's entries, DPP broadcasts) against the own-register form (quad_round_own: every lane
// keeps all entries, no broadcast) at 4 / 3 / 2 waves per SIMD.  13 rounds of 4 entries on a 32-centroid tile of 1025 rows.
// Build: hipcc --offload-arch=gfx950 -O3 -I sparsifiedkmeans_amd/csrc tools/ubench_quad.hip -o tools/ubench_quad
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "quad_steps.inc"
#include "ubench_quad_variants.inc"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int P = 1024, NR = 13;
constexpr size_t TILE = (size_t)(P + 1) * 128;

__device__ __forceinline__ unsigned hash(unsigned a) { a ^= a >> 16; a *= 0x7feb352dU; a ^= a >> 15; a *= 0x846ca68bU; a ^= a >> 16; return a; }
// row of entry e of point q: even point slots take even rows in the first half of their column, odd slots odd rows
__device__ __forceinline__ int row_of(unsigned q, int e, int ps) { unsigned r = hash(q * 64u + e) % P; const int par = ((e < 26) ? 0 : 1) ^ (ps & 1); return (int)((r & ~1u) | par); }
__device__ __forceinline__ int enc(int row) { return (row << 7) | (((row >> 1) & 3) << 4); }

template <int MODE, int THREADS, int VAR>
__global__ __launch_bounds__(THREADS) void k(float* out, int iters, unsigned long long* clk)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (size_t i = threadIdx.x; i < TILE / 4; i += THREADS) reinterpret_cast<float*>(smem)[i] = (float)(i % 97) * 0.01f;
    __syncthreads();
    const int lane = threadIdx.x & 63, ps = lane >> 2, l4 = lane & 3;
    const bool swp = (ps & 2) != 0;
    const int off0 = l4 * 16 + (swp ? 64 : 0), off1 = off0 ^ 64, delta = off1 - off0;
    const unsigned q = (blockIdx.x * THREADS + threadIdx.x) >> 2;
    const unsigned long long t0c = __builtin_readcyclecounter(), t0w = wall_clock64();
    double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
    if constexpr (MODE == 0) {
        int o[NR]; float x[NR];
#pragma unroll
        for (int r = 0; r < NR; r++) { o[r] = enc(row_of(q, 4 * r + l4, ps)); x[r] = 0.001f * (float)(hash(q + r) & 1023); }
        float acc4 = 0.f;
        if (VAR == 4) {   // f16 tile of 64 centroids: 16 per lane, v_fma_mix_f32 + v_fma_f32 per centroid (half the LDS bytes per centroid)
            float c[16];
#pragma unroll
            for (int j = 0; j < 16; j++) c[j] = 0.f;
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int r = 0; r < NR; r++) bc_mix16(__builtin_bit_cast(int, x[r]), o[r], off0, delta, c);
            }
#pragma unroll
            for (int j = 0; j < 16; j++) acc4 += c[j];
            acc0 = (double)acc4;
        } else if (VAR == 5) {   // packed f16: 32 centroids in 64-B rows, one ds_read_b128 per entry, v_pk_add_f16 + v_pk_fma_f16
            int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
            const int off16 = l4 * 16;
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int r = 0; r < NR; r++) bc_pk16((int)(0x3c003c00u ^ (hash(q + r) & 0x00ff00ffu)), (o[r] >> 1) & ~63, off16, c0, c1, c2, c3);
                c0 &= 0x3bff3bff; c1 &= 0x3bff3bff; c2 &= 0x3bff3bff; c3 &= 0x3bff3bff; // (keep the f16 sums finite)
            }
            acc0 = (double)(c0 ^ c1 ^ c2 ^ c3);
        } else
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < NR; r++) {
                if (VAR == 0) bc_full(__builtin_bit_cast(int, x[r]), o[r], off0, delta, 0, acc0, acc1, acc2, acc3, acc4);
                else if (VAR == 1) bc_noreads(__builtin_bit_cast(int, x[r]), o[r], off0, delta, 0, acc0, acc1, acc2, acc3, acc4);
                else if (VAR == 2) bc_nomath(__builtin_bit_cast(int, x[r]), o[r], off0, delta, 0, acc0, acc1, acc2, acc3, acc4);
                else bc_noaddr(__builtin_bit_cast(int, x[r]), (o[r] ^ off0), (o[r] ^ off0 ^ 64), delta, 0, acc0, acc1, acc2, acc3, acc4);
            }
        }
    } else {
        int o[4 * NR]; double xp[2 * NR];
#pragma unroll
        for (int e = 0; e < 4 * NR; e++) o[e] = enc(row_of(q, e, ps));
#pragma unroll
        for (int e = 0; e < 2 * NR; e++) { float2 v = {0.001f * (float)(hash(q + e) & 1023), 0.002f * (float)(hash(q - e) & 1023)}; xp[e] = __builtin_bit_cast(double, v); }
        double acc4 = 0;
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < NR; r++)
            {
                if (VAR == 3) {   // software-pipelined: the next round's reads are in flight during this round's arithmetic
                    if (r == 0) own_pipe_0_1_0(xp[0], xp[1], off0, off1, acc0, acc1, acc2, acc3, o[0], o[1], o[2], o[3], o[4], o[5], o[6], o[7]);
                    else if (r == NR - 1) { if (r & 1) own_pipe_1_0_1(xp[2 * r], xp[2 * r + 1], off0, off1, acc0, acc1, acc2, acc3); else own_pipe_0_0_1(xp[2 * r], xp[2 * r + 1], off0, off1, acc0, acc1, acc2, acc3); }
                    else if (r & 1) own_pipe_1_0_0(xp[2 * r], xp[2 * r + 1], off0, off1, acc0, acc1, acc2, acc3, o[4 * r + 4], o[4 * r + 5], o[4 * r + 6], o[4 * r + 7]);
                    else own_pipe_0_0_0(xp[2 * r], xp[2 * r + 1], off0, off1, acc0, acc1, acc2, acc3, o[4 * r + 4], o[4 * r + 5], o[4 * r + 6], o[4 * r + 7]);
                } else
                if (VAR == 0) own_full(xp[2 * r], xp[2 * r + 1], o[4 * r], o[4 * r + 1], o[4 * r + 2], o[4 * r + 3], off0, off1, 0, acc0, acc1, acc2, acc3, acc4);
                else if (VAR == 1) own_noreads(xp[2 * r], xp[2 * r + 1], o[4 * r], o[4 * r + 1], o[4 * r + 2], o[4 * r + 3], off0, off1, 0, acc0, acc1, acc2, acc3, acc4);
                else own_nomath(xp[2 * r], xp[2 * r + 1], o[4 * r], o[4 * r + 1], o[4 * r + 2], o[4 * r + 3], off0, off1, 0, acc0, acc1, acc2, acc3, acc4);
            }
        }
    }
    const float2 a = __builtin_bit_cast(float2, acc0), b = __builtin_bit_cast(float2, acc1), c = __builtin_bit_cast(float2, acc2), d = __builtin_bit_cast(float2, acc3);
    out[(size_t)blockIdx.x * THREADS + threadIdx.x] = a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
    if (blockIdx.x == 7 && threadIdx.x == 0) { clk[0] = __builtin_readcyclecounter() - t0c; clk[1] = wall_clock64() - t0w; }
}

static unsigned long long* d_clk;
template <int MODE, int THREADS, int VAR = 0> void run(const char* name, float* d_out, int iters)
{
    auto kern = k<MODE, THREADS, VAR>;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)TILE + 64));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = 256;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), TILE + 64, 0, d_out, 4, d_clk);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), TILE + 64, 0, d_out, iters, d_clk);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long hc[2]; CHECK(hipMemcpy(hc, d_clk, 16, hipMemcpyDeviceToHost));
    const double ghz = (double)hc[0] / ((double)hc[1] * 10.0);   // core cycles per ns (wall clock: 100 MHz)
    const double steps = (double)grid * (THREADS / 64) * iters;      // wave-steps (16 points x 32 centroids x 52 entries)
    // headline: 1e8 points x 3 tiles (+4 % remainder) = 1.95e7 wave-steps
    printf("%-34s threads=%4d  %.3f ms  %.1f ns per wave-step per CU-slot  -> rounds of a 1e8-point K=100 launch: %.1f ms | core clock %.2f GHz, %.0f core cycles per wave-step per SIMD\n", name, THREADS, ms,
           ms * 1e6 / ((double)(THREADS / 64) * iters), ms / steps * 1.95e7, ghz, ms * 1e6 / ((double)(THREADS / 64) * iters) * 4.0 * ghz);
}

int main()
{
    float* d_out; 
    CHECK(hipMalloc(&d_out, 256 * 1024 * 4));
    CHECK(hipMalloc(&d_clk, 64));
    const int iters = 2000;
    run<0, 1024>("broadcast form (quad_round)", d_out, iters);
    run<0, 1024, 1>("broadcast, no LDS reads", d_out, iters);
    run<0, 1024, 2>("broadcast, no packed math", d_out, iters);
    run<0, 1024, 3>("broadcast math + reads, no addr ops", d_out, iters);
    run<0, 1024, 4>("f16 tile, 64 centroids (x0.5 steps)", d_out, iters);
    run<0, 1024, 5>("packed f16, 32 centroids, 64-B rows", d_out, iters);
    run<0, 768, 5>("packed f16, 32 centroids, 64-B rows", d_out, iters);
    run<0, 768>("broadcast form (quad_round)", d_out, iters);
    run<0, 768, 1>("broadcast, no LDS reads", d_out, iters);
    run<0, 768, 2>("broadcast, no packed math", d_out, iters);
    run<0, 512>("broadcast form (quad_round)", d_out, iters);
    run<1, 768>("own-register form (quad_round_own)", d_out, iters);
    run<1, 768, 1>("own, no LDS reads", d_out, iters);
    run<1, 768, 2>("own, no packed math", d_out, iters);
    run<1, 512>("own-register form (quad_round_own)", d_out, iters);
    run<1, 512, 3>("own, software-pipelined reads", d_out, iters);
    return 0;
}

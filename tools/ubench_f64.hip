// Micro-benchmarks for the f64 VALU / LDS rates the assignment kernel depends on (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_f64.hip -o ubench_f64 ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
template <int MODE>
__global__ __launch_bounds__(1024) void k(double* out, int iters, double seed)
{
    extern __shared__ double lds[];
    for (int i = threadIdx.x; i < 16 * 1025; i += blockDim.x) lds[i] = seed * i;
    __syncthreads();
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double one = seed;
    int addr = (threadIdx.x & 15) * 8 + ((threadIdx.x >> 4) & 3) * 128 * 7;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
            if (MODE == 0) { // independent v_fma_f64
                asm volatile("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n"
                             "v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(one));
            } else if (MODE == 1) { // v_mul_f64 / v_add_f64 alternating, independent
                asm volatile("v_mul_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                             "v_mul_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_add_f64 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(one));
            } else if (MODE == 2) { // v_fmac_f64_dpp row_newbcast
                asm volatile("v_fmac_f64_dpp %0, %8, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %1, %8, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                             "v_fmac_f64_dpp %2, %8, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %8, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                             "v_fmac_f64_dpp %4, %8, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %5, %8, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n"
                             "v_fmac_f64_dpp %6, %8, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %7, %8, %8 row_newbcast:8 row_mask:0xf bank_mask:0xf"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(one));
            } else if (MODE == 3) { // dependent chain v_fmac_dpp -> v_mul -> v_add (one accumulator), the kernel's critical path
                asm volatile("v_fmac_f64_dpp %0, %8, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mul_f64 %0, %0, %0\n v_add_f64 %1, %1, %0\n"
                             "v_fmac_f64_dpp %2, %8, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mul_f64 %2, %2, %2\n v_add_f64 %1, %1, %2\n"
                             "v_fmac_f64_dpp %3, %8, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mul_f64 %3, %3, %3"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(one));
            } else if (MODE == 4) { // ds_read_b64 x8, conflict-free-ish pattern (4 rows per wave)
                double t0, t1, t2, t3, t4, t5, t6, t7;
                asm volatile("ds_read_b64 %0, %8\n ds_read_b64 %1, %8 offset:128\n ds_read_b64 %2, %8 offset:256\n ds_read_b64 %3, %8 offset:384\n"
                             "ds_read_b64 %4, %8 offset:512\n ds_read_b64 %5, %8 offset:640\n ds_read_b64 %6, %8 offset:768\n ds_read_b64 %7, %8 offset:896\n s_waitcnt lgkmcnt(0)"
                             : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3), "=&v"(t4), "=&v"(t5), "=&v"(t6), "=&v"(t7) : "v"(addr));
                a0 += t0; a1 += t1; a2 += t2; a3 += t3; a4 += t4; a5 += t5; a6 += t6; a7 += t7;
            } else if (MODE == 5) { // v_add_u32_dpp x8
                int b0, b1, b2, b3;
                asm volatile("v_add_u32_dpp %0, %4, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %4, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                             "v_add_u32_dpp %2, %4, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %4, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                             "v_add_u32_dpp %0, %4, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %4, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                             "v_add_u32_dpp %2, %4, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %4, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf"
                             : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3) : "v"(addr));
                addr ^= (b0 ^ b1 ^ b2 ^ b3) & 0;
            } else if (MODE == 6) { // plain v_add_u32 x8
                int b0, b1, b2, b3;
                asm volatile("v_add_u32 %0, %4, %4\n v_add_u32 %1, %4, %4\n v_add_u32 %2, %4, %4\n v_add_u32 %3, %4, %4\n"
                             "v_add_u32 %0, %4, %4\n v_add_u32 %1, %4, %4\n v_add_u32 %2, %4, %4\n v_add_u32 %3, %4, %4"
                             : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3) : "v"(addr));
                addr ^= (b0 ^ b1 ^ b2 ^ b3) & 0;
            } else if (MODE == 7) { // the kernel's step mix: add_dpp + fmac_dpp + mul + add  (x2), independent regs
                int b0, b1;
                asm volatile("v_add_u32_dpp %8, %10, %10 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %0, %11, %11 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mul_f64 %1, %1, %11\n v_add_f64 %2, %2, %11\n"
                             "v_add_u32_dpp %9, %10, %10 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp %3, %11, %11 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_mul_f64 %4, %4, %11\n v_add_f64 %5, %5, %11"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(b0), "=&v"(b1) : "v"(addr), "v"(one));
                addr ^= (b0 ^ b1) & 0;
            } else if (MODE == 8) { // step mix with plain add instead of DPP add
                int b0, b1;
                asm volatile("v_add_u32 %8, %10, %10\n v_fmac_f64_dpp %0, %11, %11 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mul_f64 %1, %1, %11\n v_add_f64 %2, %2, %11\n"
                             "v_add_u32 %9, %10, %10\n v_fmac_f64_dpp %3, %11, %11 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_mul_f64 %4, %4, %11\n v_add_f64 %5, %5, %11"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(b0), "=&v"(b1) : "v"(addr), "v"(one));
                addr ^= (b0 ^ b1) & 0;
            } else if (MODE == 9) { // v_mov_b32_dpp x8
                int b0, b1, b2, b3;
                asm volatile("v_mov_b32_dpp %0, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %4 row_newbcast:2 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %2, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %2, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
                             : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3) : "v"(addr));
                addr ^= (b0 ^ b1 ^ b2 ^ b3) & 0;
            } else if (MODE == 11) { // f32 screen step, packed: mov_b64_dpp + add_u32_dpp + pk_add_f32 + pk_fma_f32 (x2)
                int b0, b1;
                asm volatile("v_mov_b64_dpp %0, %11 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %8, %10, %10 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_pk_add_f32 %1, %1, %11\n v_pk_fma_f32 %2, %2, %11, %2\n"
                             "v_mov_b64_dpp %3, %11 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %9, %10, %10 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_pk_add_f32 %4, %4, %11\n v_pk_fma_f32 %5, %5, %11, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(b0), "=&v"(b1) : "v"(addr), "v"(one));
                addr ^= (b0 ^ b1) & 0;
            } else if (MODE == 12) { // f32 screen step, scalar f32 with DPP operand: add_u32_dpp + 2x v_sub_f32_dpp + 2x v_fmac_f32 (8 instr = 1.6 steps)
                int b0, b1; float f0 = (float)seed, f1 = f0, f2 = f0, f3 = f0, g = f0;
                asm volatile("v_add_u32_dpp %0, %6, %6 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_sub_f32_dpp %2, %7, %7 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_sub_f32_dpp %3, %7, %7 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_fmac_f32 %4, %7, %7\n v_fmac_f32 %5, %7, %7\n"
                             "v_add_u32_dpp %1, %6, %6 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_sub_f32_dpp %2, %7, %7 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_sub_f32_dpp %3, %7, %7 row_newbcast:2 row_mask:0xf bank_mask:0xf"
                             : "=&v"(b0), "=&v"(b1), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(addr), "v"(g));
                addr ^= (b0 ^ b1) & 0; a0 += f0 + f1 + f2 + f3;
            } else if (MODE == 13) { // v_pk_fma_f32 x8 independent
                asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n v_pk_fma_f32 %1, %1, %8, %8\n v_pk_fma_f32 %2, %2, %8, %8\n v_pk_fma_f32 %3, %3, %8, %8\n"
                             "v_pk_fma_f32 %4, %4, %8, %8\n v_pk_fma_f32 %5, %5, %8, %8\n v_pk_fma_f32 %6, %6, %8, %8\n v_pk_fma_f32 %7, %7, %8, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(one));
            } else if (MODE == 14) { // v_mov_b64_dpp x8
                asm volatile("v_mov_b64_dpp %0, %8 row_newbcast:1 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %1, %8 row_newbcast:2 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %2, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %3, %8 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b64_dpp %4, %8 row_newbcast:5 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %5, %8 row_newbcast:6 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %6, %8 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp %7, %8 row_newbcast:8 row_mask:0xf bank_mask:0xf"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(one));
            } else if (MODE == 15) { // v_pk_fma_f16 x8 independent (32-bit registers)
                int h0 = addr, h1 = addr + 1, h2 = addr + 2, h3 = addr + 3, h4 = addr + 4, h5 = addr + 5, h6 = addr + 6, h7 = addr + 7;
                asm volatile("v_pk_fma_f16 %0, %0, %8, %8\n v_pk_fma_f16 %1, %1, %8, %8\n v_pk_fma_f16 %2, %2, %8, %8\n v_pk_fma_f16 %3, %3, %8, %8\n"
                             "v_pk_fma_f16 %4, %4, %8, %8\n v_pk_fma_f16 %5, %5, %8, %8\n v_pk_fma_f16 %6, %6, %8, %8\n v_pk_fma_f16 %7, %7, %8, %8"
                             : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7) : "v"(addr));
                addr ^= (h0 ^ h1 ^ h2 ^ h3 ^ h4 ^ h5 ^ h6 ^ h7) & 0;
            } else if (MODE == 16) { // v_pk_add_f16 x8 independent
                int h0 = addr, h1 = addr + 1, h2 = addr + 2, h3 = addr + 3, h4 = addr + 4, h5 = addr + 5, h6 = addr + 6, h7 = addr + 7;
                asm volatile("v_pk_add_f16 %0, %0, %8\n v_pk_add_f16 %1, %1, %8\n v_pk_add_f16 %2, %2, %8\n v_pk_add_f16 %3, %3, %8\n"
                             "v_pk_add_f16 %4, %4, %8\n v_pk_add_f16 %5, %5, %8\n v_pk_add_f16 %6, %6, %8\n v_pk_add_f16 %7, %7, %8"
                             : "+v"(h0), "+v"(h1), "+v"(h2), "+v"(h3), "+v"(h4), "+v"(h5), "+v"(h6), "+v"(h7) : "v"(addr));
                addr ^= (h0 ^ h1 ^ h2 ^ h3 ^ h4 ^ h5 ^ h6 ^ h7) & 0;
            } else if (MODE == 17) { // v_pk_add_f32 x8 independent
                asm volatile("v_pk_add_f32 %0, %0, %8\n v_pk_add_f32 %1, %1, %8\n v_pk_add_f32 %2, %2, %8\n v_pk_add_f32 %3, %3, %8\n"
                             "v_pk_add_f32 %4, %4, %8\n v_pk_add_f32 %5, %5, %8\n v_pk_add_f32 %6, %6, %8\n v_pk_add_f32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(one));
            } else if (MODE == 18) { // v_dot2_f32_f16 x8 (f16 pairs, f32 accumulate)
                float g0 = 1.f, g1 = 2.f, g2 = 3.f, g3 = 4.f;
                asm volatile("v_dot2_f32_f16 %0, %4, %4, %0\n v_dot2_f32_f16 %1, %4, %4, %1\n v_dot2_f32_f16 %2, %4, %4, %2\n v_dot2_f32_f16 %3, %4, %4, %3\n"
                             "v_dot2_f32_f16 %0, %4, %4, %0\n v_dot2_f32_f16 %1, %4, %4, %1\n v_dot2_f32_f16 %2, %4, %4, %2\n v_dot2_f32_f16 %3, %4, %4, %3"
                             : "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3) : "v"(addr));
                a0 += g0 + g1 + g2 + g3;
            } else if (MODE == 19) { // ds_read_b128 x4 in flight, 4 lanes per 64-B segment
                float4 t0, t1, t2, t3;
                int ad = (threadIdx.x & 3) * 16 + ((threadIdx.x >> 2) & 15) * 128 * 5 + ((threadIdx.x >> 3) & 1) * 64;
                asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n ds_read_b128 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)"
                             : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(ad));
                a0 += t0.x + t1.x + t2.x + t3.x;
            } else if (MODE == 10) { // v_fma_f32 x8 independent
                float f = (float)seed;
                asm volatile("v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n"
                             "v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1\n v_fma_f32 %0, %0, %1, %1"
                             : "+v"(addr) : "v"(f));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE>
void run(const char* name, int threads, int blocks_per_cu)
{
    int dev = 0; hipDeviceProp_t prop; hipGetDeviceProperties(&prop, dev);
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * blocks_per_cu, iters = 2000;
    double* out; hipMalloc(&out, (size_t)blocks * threads * 8);
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 1025 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, threads, 16 * 1025 * 8>>>(out, 10, 1.0);
    hipEventRecord(e0);
    k<MODE><<<blocks, threads, 16 * 1025 * 8>>>(out, iters, 1.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = (double)threads / 64 * blocks_per_cu / 4;
    const double insts_per_wave = (double)iters * REP;
    // wave-instructions issued per SIMD / time -> cycles per wave-instruction per SIMD at 2.4 GHz
    const double cyc = ms * 1e-3 * 2.4e9 / (insts_per_wave * waves_per_simd);
    printf("%-34s threads=%4d waves/SIMD=%.0f  %.3f ms  -> %.2f cycles per wave-instr per SIMD (@2.4GHz nominal)\n", name,
           threads, waves_per_simd, ms, cyc);
    hipFree(out);
}

__global__ void k_clock(long long* out, int iters)
{
    long long t0 = __builtin_amdgcn_s_memtime();
    long long w0 = wall_clock64();
    double a = threadIdx.x;
    for (int i = 0; i < iters; i++) asm volatile("v_fma_f64 %0, %0, %0, %0" : "+v"(a));
    long long t1 = __builtin_amdgcn_s_memtime();
    long long w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = w1 - w0; out[2] = (long long)a; }
}

int main()
{
    {
        long long* d; hipMalloc(&d, 64); long long h[3];
        k_clock<<<1024, 1024>>>(d, 200000);
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("s_memtime ticks=%lld  wall_clock64 ticks (100MHz)=%lld -> s_memtime runs at %.1f MHz under f64 load\n", h[0], h[1], h[0] / (h[1] / 100.0));
    }
    for (int t : {1024}) {
        run<0>("v_fma_f64 independent", t, 1);
        run<1>("v_mul_f64/v_add_f64 independent", t, 1);
        run<2>("v_fmac_f64_dpp independent", t, 1);
        run<3>("fmac_dpp->mul->add dependent", t, 1);
        run<4>("ds_read_b64 (8 in flight)", t, 1);
        run<5>("v_add_u32_dpp", t, 1);
        run<6>("v_add_u32 plain", t, 1);
        run<7>("step mix (dpp add)", t, 1);
        run<8>("step mix (plain add)", t, 1);
        run<9>("v_mov_b32_dpp", t, 1);
        run<10>("v_fma_f32 dependent chain", t, 1);
        run<11>("f32 screen step packed (4 instr)", t, 1);
        run<12>("f32 screen step scalar+dpp (8 instr)", t, 1);
        run<13>("v_pk_fma_f32 independent", t, 1);
        run<14>("v_mov_b64_dpp", t, 1);
        run<15>("v_pk_fma_f16 independent", t, 1);
        run<16>("v_pk_add_f16 independent", t, 1);
        run<17>("v_pk_add_f32 independent", t, 1);
        run<18>("v_dot2_f32_f16", t, 1);
        run<19>("ds_read_b128 (4 in flight; counts 8 per REP)", t, 1);
    }
    return 0;
}

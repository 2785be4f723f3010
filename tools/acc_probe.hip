// Stand-alone probe (GPU box): how fast can one workgroup-per-CU kernel stream the record layout of the exact
// accumulation pass (512-B records, gathered through a permutation) with different load shapes?  No LDS work, no
// arithmetic beyond a checksum: the ceiling of each access pattern.
//   hipcc --offload-arch=gfx950 -O3 tools/acc_probe.hip -o tools/acc_probe && tools/acc_probe [npoints] [shuffled]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int S = 51, R = 512;

// A: lanes <-> entries, 8-B + 2-B loads per lane, U points in flight (the shape k_exact_accumulate uses)
template <int U, bool NT>
__global__ __launch_bounds__(1024) void k_a(const char* __restrict__ rec, const int* __restrict__ perm, long long n,
                                            double* __restrict__ out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long nw = (long long)gridDim.x * 16;
    const long long per = (n + nw - 1) / nw;
    const long long w = (long long)blockIdx.x * 16 + wave;
    const long long lo = w * per, hi = (lo + per < n) ? lo + per : n;
    const int lanec = lane < S ? lane : S - 1;
    const unsigned offx = lanec * 8u, offr = S * 8u + lanec * 2u;
    double acc = 0.0;
    int racc = 0;
    for (long long q = lo; q < hi; q += 64) {
        const long long my_i = (q + lane < hi) ? perm[q + lane] : 0;
        const int have = (hi - q < 64) ? (int)(hi - q) : 64;
        for (int u = 0; u < have; u += U) {
            double xv[U];
            int rv[U];
#pragma unroll
            for (int v = 0; v < U; v++) {
                const int src = (u + v < have) ? u + v : u;
                const long long i = (long long)(unsigned)__builtin_amdgcn_readlane((int)my_i, src);
                const char* b = rec + (size_t)i * R;
                if (NT) {
                    xv[v] = __builtin_nontemporal_load(reinterpret_cast<const double*>(b + offx));
                    rv[v] = __builtin_nontemporal_load(reinterpret_cast<const unsigned short*>(b + offr));
                } else {
                    xv[v] = *reinterpret_cast<const double*>(b + offx);
                    rv[v] = *reinterpret_cast<const unsigned short*>(b + offr);
                }
            }
#pragma unroll
            for (int v = 0; v < U; v++) { acc += xv[v]; racc ^= rv[v]; }
        }
    }
    if (acc == 1.2345 || racc == 0x7fffffff) out[0] = acc;
}

// B: 16 B per lane, 32 lanes per record -> two records per wave instruction, U instruction pairs in flight
typedef double d2 __attribute__((ext_vector_type(2)));
template <int U, bool NT>
__global__ __launch_bounds__(1024) void k_b(const char* __restrict__ rec, const int* __restrict__ perm, long long n,
                                            double* __restrict__ out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long nw = (long long)gridDim.x * 16;
    const long long per = (n + nw - 1) / nw;
    const long long w = (long long)blockIdx.x * 16 + wave;
    const long long lo = w * per, hi = (lo + per < n) ? lo + per : n;
    const int half = lane >> 5;
    const unsigned off = (lane & 31) * 16u;
    double acc = 0.0;
    for (long long q = lo; q < hi; q += 64) {
        const long long my_i = (q + lane < hi) ? perm[q + lane] : 0;
        const int have = (hi - q < 64) ? (int)(hi - q) : 64;
        for (int u = 0; u < have; u += 2 * U) {
            d2 xv[U];
#pragma unroll
            for (int v = 0; v < U; v++) {
                const int s0 = (u + 2 * v < have) ? u + 2 * v : u, s1 = (u + 2 * v + 1 < have) ? u + 2 * v + 1 : u;
                const long long i0 = (long long)(unsigned)__builtin_amdgcn_readlane((int)my_i, s0);
                const long long i1 = (long long)(unsigned)__builtin_amdgcn_readlane((int)my_i, s1);
                const long long i = half ? i1 : i0;
                const char* b = rec + (size_t)i * R + off;
                if (NT) xv[v] = __builtin_nontemporal_load(reinterpret_cast<const d2*>(b));
                else xv[v] = *reinterpret_cast<const d2*>(b);
            }
#pragma unroll
            for (int v = 0; v < U; v++) acc += xv[v].x + xv[v].y;
        }
    }
    if (acc == 1.2345) out[0] = acc;
}

// D: the accumulation kernel's work distribution with pattern A's loads: segments of SEG points handed to workgroups
// round-robin, the 16 waves of a workgroup interleaved at 16-point batches, three workgroup barriers per segment
template <int U, int SEG, int NBAR, int FEAT = 0> // FEAT: 1 per-point stores (8 B + 4 B by id), 2 LDS gather + 2 f64 ops per entry, 4 slab init per segment
__global__ __launch_bounds__(1024) void k_d(const char* __restrict__ rec, const int* __restrict__ perm, long long n,
                                            double* __restrict__ out, double* __restrict__ mind = nullptr,
                                            float* __restrict__ ub = nullptr, const double* __restrict__ Ccol = nullptr)
{
    __shared__ double negc[1024];
    __shared__ double ssum[1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lanec = lane < S ? lane : S - 1;
    const unsigned offx = lanec * 8u, offr = S * 8u + lanec * 2u;
    double acc = 0.0;
    int racc = 0;
    if (FEAT & 2) { negc[threadIdx.x] = (double)threadIdx.x * 0.001; }
    __syncthreads();
    const long long nseg = (n + SEG - 1) / SEG;
    for (long long seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
        const long long start = seg * SEG;
        const int len = (int)((n - start < SEG) ? n - start : SEG);
        if (FEAT & 4) { negc[threadIdx.x] = -Ccol[(seg % 100) * 1024 + threadIdx.x] / 0.05; ssum[threadIdx.x] = 0.0; }
        if (NBAR > 0) __syncthreads();
        for (int qb = wave * U; qb < len; qb += 16 * U) {
            const int have = (len - qb < U) ? len - qb : U;
            const int lp = lane < U ? lane : U - 1;
            const int my_i = perm[start + min(qb + lp, len - 1)];
            double xv[U];
            int rv[U];
#pragma unroll
            for (int v = 0; v < U; v++) {
                const int src = (v < have) ? v : 0;
                const long long i = (long long)(unsigned)__builtin_amdgcn_readlane(my_i, src);
                const char* b = rec + (size_t)i * R;
                xv[v] = __builtin_nontemporal_load(reinterpret_cast<const double*>(b + offx));
                rv[v] = __builtin_nontemporal_load(reinterpret_cast<const unsigned short*>(b + offr));
            }
#pragma unroll
            for (int v = 0; v < U; v++) {
                if (FEAT & 2) { const double d = xv[v] + negc[rv[v] & 1023]; acc += d * d; }
                else { acc += xv[v]; racc ^= rv[v]; }
            }
            if ((FEAT & 1) && lane < have) {
                if (FEAT & 32) { // nontemporal stores
                    __builtin_nontemporal_store(acc, &mind[(unsigned)my_i]);
                    if (!(FEAT & 8)) __builtin_nontemporal_store((float)acc, &ub[(unsigned)my_i]);
                } else {
                    if (!(FEAT & 16)) mind[(unsigned)my_i] = acc;
                    if (!(FEAT & 8)) ub[(unsigned)my_i] = (float)acc;
                }
            }
        }
        if (NBAR > 1) __syncthreads();
        if (NBAR > 2) __syncthreads();
    }
    if (acc == 1.2345 || racc == 0x7fffffff) out[0] = acc + ssum[3];
}

// C: plain sequential float4-style stream over the whole array (no permutation): the box's streaming ceiling
__global__ __launch_bounds__(1024) void k_c(const char* __restrict__ rec, long long bytes, double* __restrict__ out)
{
    const d2* p = reinterpret_cast<const d2*>(rec);
    const long long nv = bytes / 16;
    double acc = 0.0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < nv; i += 4 * stride) {
        const d2 a = __builtin_nontemporal_load(p + i), b = __builtin_nontemporal_load(p + i + stride),
                 c = __builtin_nontemporal_load(p + i + 2 * stride), d = __builtin_nontemporal_load(p + i + 3 * stride);
        acc += a.x + a.y + b.x + b.y + c.x + c.y + d.x + d.y;
    }
    if (acc == 1.2345) out[0] = acc;
}

// fills the records with what a shard holds: random doubles (values) and random row ids -- memory-bound kernels on
// this part run measurably faster on constant data (fewer bit toggles), so a memset buffer flatters the numbers
__global__ void k_fill(char* rec, long long n, int constant)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n * (R / 8); i += (long long)gridDim.x * blockDim.x) {
        unsigned long long z = (unsigned long long)i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
        z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 27; z *= 0x94D049BB133111EBull; z ^= z >> 31;
        const int w = (int)(i % (R / 8));
        double v;
        if (w < S) v = ((double)(long long)(z >> 11) / 9007199254740992.0 - 0.5) * 40.0;   // a value
        else { unsigned long long ids = z & 0x03ff03ff03ff03ffull; v = __longlong_as_double((long long)ids); } // four row ids < 1024
        if (constant) v = __longlong_as_double(0x0101010101010101LL);
        reinterpret_cast<double*>(rec)[i] = v;
    }
}

template <typename F>
static float timeit(F f, int reps = 5)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    f();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int r = 0; r < reps; r++) f();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms;
    CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / reps;
}

int main(int argc, char** argv)
{
    const long long n = argc > 1 ? atoll(argv[1]) : 100000000LL;
    const bool shuffled = argc > 2 && atoi(argv[2]) != 0;
    const int constant = argc > 3 ? atoi(argv[3]) : 0;
    char* rec;
    int* perm;
    double* out;
    CHECK(hipMalloc(&rec, (size_t)n * R + 4096));
    CHECK(hipMemset(rec, 1, (size_t)n * R + 4096));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, rec, n, constant);
    CHECK(hipDeviceSynchronize());
    CHECK(hipMalloc(&perm, (size_t)n * 4));
    CHECK(hipMalloc(&out, 64));
    {
        std::vector<int> h(n);
        for (long long i = 0; i < n; i++) h[i] = (int)i;
        if (shuffled) { std::mt19937_64 g(1); std::shuffle(h.begin(), h.end(), g); }
        CHECK(hipMemcpy(perm, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    }
    const double gb = (double)n * (S * 10 + 4) / 1e9;
    printf("%s data; ", constant ? "constant" : "random");
    printf("n = %lld, %s order, useful bytes %.1f GB (records %.1f GB)\n", n, shuffled ? "shuffled" : "sequential", gb, (double)n * R / 1e9);
    auto rep = [&](const char* name, float ms) { printf("%-44s %7.2f ms  %6.2f TB/s useful\n", name, ms, gb / ms); fflush(stdout); };
    rep("A  8B+2B per lane, U=16", timeit([&] { hipLaunchKernelGGL((k_a<16, false>), dim3(256), dim3(1024), 0, 0, rec, perm, n, out); }));
    rep("A  8B+2B per lane, U=16, nt", timeit([&] { hipLaunchKernelGGL((k_a<16, true>), dim3(256), dim3(1024), 0, 0, rec, perm, n, out); }));
    rep("A  8B+2B per lane, U=8, nt", timeit([&] { hipLaunchKernelGGL((k_a<8, true>), dim3(256), dim3(1024), 0, 0, rec, perm, n, out); }));
    rep("A  8B+2B per lane, U=16, nt, 2 WG/CU", timeit([&] { hipLaunchKernelGGL((k_a<16, true>), dim3(512), dim3(1024), 0, 0, rec, perm, n, out); }));
    rep("B  16B per lane (2 records/instr), U=8", timeit([&] { hipLaunchKernelGGL((k_b<8, false>), dim3(256), dim3(1024), 0, 0, rec, perm, n, out); }));
    rep("B  16B per lane, U=8, nt", timeit([&] { hipLaunchKernelGGL((k_b<8, true>), dim3(256), dim3(1024), 0, 0, rec, perm, n, out); }));
    rep("B  16B per lane, U=4, nt", timeit([&] { hipLaunchKernelGGL((k_b<4, true>), dim3(256), dim3(1024), 0, 0, rec, perm, n, out); }));
    rep("B  16B per lane, U=8, nt, 2 WG/CU", timeit([&] { hipLaunchKernelGGL((k_b<8, true>), dim3(512), dim3(1024), 0, 0, rec, perm, n, out); }));
    rep("B  16B per lane, U=16, nt", timeit([&] { hipLaunchKernelGGL((k_b<16, true>), dim3(256), dim3(1024), 0, 0, rec, perm, n, out); }));
    rep("D  kernel-like: seg 8192, 16 waves interleaved, 3 barriers", timeit([&] { hipLaunchKernelGGL((k_d<16, 8192, 3>), dim3(256), dim3(1024), 0, 0, rec, perm, n, out); }));
    {
        double *mind, *Cc; float* ub;
        CHECK(hipMalloc(&mind, (size_t)n * 8)); CHECK(hipMalloc(&ub, (size_t)n * 4)); CHECK(hipMalloc(&Cc, 100 * 1024 * 8));
        CHECK(hipMemset(Cc, 0, 100 * 1024 * 8));
        rep("D1 + per-point stores (8 B + 4 B)", timeit([&] { hipLaunchKernelGGL((k_d<16, 8192, 3, 1>), dim3(256), dim3(1024), 0, 0, rec, perm, n, out, mind, ub, Cc); }));
        rep("D2 + LDS gather and 2 f64 ops per entry", timeit([&] { hipLaunchKernelGGL((k_d<16, 8192, 3, 2>), dim3(256), dim3(1024), 0, 0, rec, perm, n, out, mind, ub, Cc); }));
        rep("D4 + slab init per segment", timeit([&] { hipLaunchKernelGGL((k_d<16, 8192, 3, 4>), dim3(256), dim3(1024), 0, 0, rec, perm, n, out, mind, ub, Cc); }));
        rep("D7 all three", timeit([&] { hipLaunchKernelGGL((k_d<16, 8192, 3, 7>), dim3(256), dim3(1024), 0, 0, rec, perm, n, out, mind, ub, Cc); }));
        rep("D1 only the 8-B store", timeit([&] { hipLaunchKernelGGL((k_d<16, 8192, 3, 1 | 8>), dim3(256), dim3(1024), 0, 0, rec, perm, n, out, mind, ub, Cc); }));
        rep("D1 only the 4-B store", timeit([&] { hipLaunchKernelGGL((k_d<16, 8192, 3, 1 | 16>), dim3(256), dim3(1024), 0, 0, rec, perm, n, out, mind, ub, Cc); }));
        rep("D1 both stores, nontemporal", timeit([&] { hipLaunchKernelGGL((k_d<16, 8192, 3, 1 | 32>), dim3(256), dim3(1024), 0, 0, rec, perm, n, out, mind, ub, Cc); }));
        rep("D1 8-B store, nontemporal", timeit([&] { hipLaunchKernelGGL((k_d<16, 8192, 3, 1 | 32 | 8>), dim3(256), dim3(1024), 0, 0, rec, perm, n, out, mind, ub, Cc); }));
        rep("D1 U=64-point batches (512-B + 256-B stores)", timeit([&] { hipLaunchKernelGGL((k_d<64, 8192, 3, 1>), dim3(256), dim3(1024), 0, 0, rec, perm, n, out, mind, ub, Cc); }));
        rep("D3 stores + LDS/f64", timeit([&] { hipLaunchKernelGGL((k_d<16, 8192, 3, 3>), dim3(256), dim3(1024), 0, 0, rec, perm, n, out, mind, ub, Cc); }));
    }
    rep("D  seg 8192, no barriers", timeit([&] { hipLaunchKernelGGL((k_d<16, 8192, 0>), dim3(256), dim3(1024), 0, 0, rec, perm, n, out); }));
    rep("D  seg 65536, 3 barriers", timeit([&] { hipLaunchKernelGGL((k_d<16, 65536, 3>), dim3(256), dim3(1024), 0, 0, rec, perm, n, out); }));
    rep("D  seg 2048, 3 barriers", timeit([&] { hipLaunchKernelGGL((k_d<16, 2048, 3>), dim3(256), dim3(1024), 0, 0, rec, perm, n, out); }));
    rep("D  seg 8192, 3 barriers, 512 WGs", timeit([&] { hipLaunchKernelGGL((k_d<16, 8192, 3>), dim3(512), dim3(1024), 0, 0, rec, perm, n, out); }));
    {
        const long long bytes = n * R;
        float ms = timeit([&] { hipLaunchKernelGGL(k_c, dim3(2048), dim3(1024), 0, 0, rec, bytes, out); });
        printf("%-44s %7.2f ms  %6.2f TB/s (all %.1f GB)\n", "C  sequential 16B stream, nt", ms, (double)bytes / 1e9 / ms, (double)bytes / 1e9);
    }
    return 0;
}

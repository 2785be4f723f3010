// Verifies on hardware that v_pk_add_f32 honours op_sel / op_sel_hi on the dwords of a 64-bit source.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(float* out)
{
    double t = __builtin_bit_cast(double, ((unsigned long long)__builtin_bit_cast(unsigned, 10.0f) << 32) | __builtin_bit_cast(unsigned, 1.0f)); // (lo=1, hi=10)
    double x = __builtin_bit_cast(double, ((unsigned long long)__builtin_bit_cast(unsigned, 200.0f) << 32) | __builtin_bit_cast(unsigned, 100.0f)); // (lo=100, hi=200)
    double r0, r1;
    asm volatile("v_pk_add_f32 %0, %2, %3 op_sel:[0,0] op_sel_hi:[1,0]\n v_pk_add_f32 %1, %2, %3 op_sel:[0,1] op_sel_hi:[1,1]"
                 : "=&v"(r0), "=&v"(r1) : "v"(t), "v"(x));
    unsigned long long b0 = __builtin_bit_cast(unsigned long long, r0), b1 = __builtin_bit_cast(unsigned long long, r1);
    out[0] = __builtin_bit_cast(float, (unsigned)b0); out[1] = __builtin_bit_cast(float, (unsigned)(b0 >> 32));
    out[2] = __builtin_bit_cast(float, (unsigned)b1); out[3] = __builtin_bit_cast(float, (unsigned)(b1 >> 32));
}
int main()
{
    float* d; hipMalloc(&d, 16); float h[4];
    k<<<1, 64>>>(d); hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("op_sel lo-broadcast: (%g, %g) expect (101, 110); hi-broadcast: (%g, %g) expect (201, 210)\n", h[0], h[1], h[2], h[3]);
    return 0;
}

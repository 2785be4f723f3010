#!/usr/bin/env python3
"""Generates assign_steps.inc: hand-scheduled gfx950 inner blocks of the tiled
assignment kernel (assign.hip), one per batch length N = 1..16.

A block consumes N consecutive stored entries of TWO independent groups of points (A and B);
within a group, each 16-lane DPP row owns one point.  Per entry j of a point, strictly in
order (reference arithmetic, private/SparseMatrixMinusCluster.c:173-176):

    a     = koff + roff[row lane j]            v_add_u32_dpp   row_newbcast:j
    t_j   = LDS[a]     ( = -C[r_j][k]/gamma )  ds_read_b64
    t_j   = x[row lane j] * 1.0 + t_j          v_fmac_f64_dpp  row_newbcast:j   == RN(x - c)
    t_j   = t_j * t_j                          v_mul_f64
    acc   = acc + t_j                          v_add_f64

Schedule: all N reads of group A are issued first; group A's math is interleaved with the
issue of group B's reads (so B's LDS latency is fully hidden and A's mostly, behind the 16
address adds), then group B's math.  The three dependent f64 ops of one entry are skewed by
two entries so back-to-back instructions are independent.  LDS returns in order, so the wait
before A_j is lgkmcnt(N-1) throughout phase 1 (N-1-j of A still pending + j of B issued).

steps2p<N, IRBYTES> additionally issues, right after group A's reads, the global loads of the
NEXT batch's entries (x: 8 B, row id: 2 or 4 B per lane; SGPR base + 32-bit lane offset) and
waits for them at the very end of the block -- a whole block of latency hiding with nothing in
flight across the statement boundary, which hipcc could not track
(cdna_hip_programming.md §5.7: an asm load must be waited for inside its own statement).
"""
import os

DPP = "row_mask:0xf bank_mask:0xf"
SK = 2  # skew (in entries) between the dependent fmac -> mul -> add of one entry


def block(n: int, prefetch: bool, irbytes: int) -> str:
    L = []
    L.append("s_waitcnt lgkmcnt(0)")  # SMEM returns out of order: start the LDS count clean
    # roff / x come from VALU ops; VALU-write -> DPP-read needs 2 wait states that hipcc
    # cannot see inside an asm statement
    L.append("s_nop 1")
    for j in range(n):
        L.append(f"v_add_u32_dpp %[a{j % 4}], %[roffA], %[koff] row_newbcast:{j} {DPP}")
        L.append(f"ds_read_b64 %[tA{j}], %[a{j % 4}]")
    if prefetch:
        ld = "global_load_ushort" if irbytes == 2 else "global_load_dword"
        L.append("global_load_dwordx2 %[xAn], %[voxA], %[xbase]")
        L.append("global_load_dwordx2 %[xBn], %[voxB], %[xbase]")
        L.append(f"{ld} %[rAn], %[vorA], %[rbase]")
        L.append(f"{ld} %[rBn], %[vorB], %[rbase]")
    # phase 1: math(A) + issue(B)
    for j in range(n + 2 * SK):
        if j < n:
            L.append(f"s_waitcnt lgkmcnt({n - 1})")
            L.append(f"v_fmac_f64_dpp %[tA{j}], %[xA], %[one] row_newbcast:{j} {DPP}")
            L.append(f"v_add_u32_dpp %[a{j % 4}], %[roffB], %[koff] row_newbcast:{j} {DPP}")
            L.append(f"ds_read_b64 %[tB{j}], %[a{j % 4}]")
        if 0 <= j - SK < n:
            L.append(f"v_mul_f64 %[tA{j-SK}], %[tA{j-SK}], %[tA{j-SK}]")
        if 0 <= j - 2 * SK < n:
            L.append(f"v_add_f64 %[accA], %[accA], %[tA{j-2*SK}]")
    # phase 2: math(B)
    for j in range(n + 2 * SK):
        if j < n:
            L.append(f"s_waitcnt lgkmcnt({n - 1 - j})")
            L.append(f"v_fmac_f64_dpp %[tB{j}], %[xB], %[one] row_newbcast:{j} {DPP}")
        if 0 <= j - SK < n:
            L.append(f"v_mul_f64 %[tB{j-SK}], %[tB{j-SK}], %[tB{j-SK}]")
        if 0 <= j - 2 * SK < n:
            L.append(f"v_add_f64 %[accB], %[accB], %[tB{j-2*SK}]")
    if prefetch:
        L.append("s_waitcnt vmcnt(0)")
    return "\\n\\t".join(L)


def func(n: int, prefetch: bool, irbytes: int) -> str:
    outs = ['[accA] "+v"(accA)', '[accB] "+v"(accB)']
    outs += [f'[a{j}] "=&v"(a{j})' for j in range(4)]
    outs += [f'[tA{j}] "=&v"(tA{j})' for j in range(n)]
    outs += [f'[tB{j}] "=&v"(tB{j})' for j in range(n)]
    ins = ['[xA] "v"(xA)', '[xB] "v"(xB)', '[one] "v"(one)', '[roffA] "v"(roffA)', '[roffB] "v"(roffB)',
           '[koff] "v"(koff)']
    decl_t = ", ".join([f"tA{j}" for j in range(n)] + [f"tB{j}" for j in range(n)])
    if prefetch:
        outs += ['[xAn] "=&v"(xAn)', '[xBn] "=&v"(xBn)', '[rAn] "=&v"(rAn)', '[rBn] "=&v"(rBn)']
        ins += ['[xbase] "s"(xbase)', '[rbase] "s"(rbase)', '[voxA] "v"(voxA)', '[voxB] "v"(voxB)',
                '[vorA] "v"(vorA)', '[vorB] "v"(vorB)']
        sig = (f"template <>\n__device__ __forceinline__ void steps2p<{n}, {irbytes}>(int koff, int roffA, double xA, "
               "int roffB, double xB, double one,\n    double& accA, double& accB, const void* xbase, const void* rbase, "
               "unsigned voxA, unsigned voxB, unsigned vorA, unsigned vorB,\n    double& xAn, double& xBn, int& rAn, int& rBn)")
    else:
        sig = (f"template <>\n__device__ __forceinline__ void steps2<{n}>(int koff, int roffA, double xA, int roffB, "
               "double xB, double one,\n                                            double& accA, double& accB)")
    return f"""{sig}
{{
    int a0, a1, a2, a3;
    double {decl_t};
    asm volatile("{block(n, prefetch, irbytes)}"
                 : {", ".join(outs)}
                 : {", ".join(ins)});
}}
"""


def screen_block(n: int, irbytes: int) -> str:
    """f32 SCREEN block (not the reference arithmetic -- a certified pre-filter, see screen.hip):
    each lane owns two centroids (2*kk, 2*kk+1) of a 32-wide f32 tile, so a 16-lane DPP row
    covers 32 centroids of one point.  Per entry j:
        a    = koff + roff[row lane j]                 v_add_u32_dpp
        t    = LDS[a]        (-c0, -c1 as f32)         ds_read_b64
        xb   = (x, x)[row lane j]                      v_mov_b64_dpp
        t    = t + xb        (x-c0, x-c1)              v_pk_add_f32
        acc  = t*t + acc                               v_pk_fma_f32
    Same two-group schedule and in-statement prefetch as the exact blocks."""
    L = ["s_waitcnt lgkmcnt(0)", "s_nop 1"]
    for j in range(n):
        L.append(f"v_add_u32_dpp %[a{j % 4}], %[roffA], %[koff] row_newbcast:{j} {DPP}")
        L.append(f"ds_read_b64 %[tA{j}], %[a{j % 4}]")
    ld = "global_load_ushort" if irbytes == 2 else "global_load_dword"
    L.append("global_load_dword %[xAn], %[voxA], %[xbase]")   # f32 copy of x
    L.append("global_load_dword %[xBn], %[voxB], %[xbase]")
    L.append(f"{ld} %[rAn], %[vorA], %[rbase]")
    L.append(f"{ld} %[rBn], %[vorB], %[rbase]")
    for j in range(n + 2):
        if j < n:
            L.append(f"s_waitcnt lgkmcnt({n - 1})")
            L.append(f"v_mov_b64_dpp %[xb{j % 4}], %[xA] row_newbcast:{j} {DPP}")
            L.append(f"v_add_u32_dpp %[a{j % 4}], %[roffB], %[koff] row_newbcast:{j} {DPP}")
            L.append(f"ds_read_b64 %[tB{j}], %[a{j % 4}]")
        if 0 <= j - 1 < n:
            L.append(f"v_pk_add_f32 %[tA{j-1}], %[tA{j-1}], %[xb{(j-1) % 4}]")
        if 0 <= j - 2 < n:
            L.append(f"v_pk_fma_f32 %[accA], %[tA{j-2}], %[tA{j-2}], %[accA]")
    for j in range(n + 2):
        if j < n:
            L.append(f"s_waitcnt lgkmcnt({n - 1 - j})")
            L.append(f"v_mov_b64_dpp %[xb{j % 4}], %[xB] row_newbcast:{j} {DPP}")
        if 0 <= j - 1 < n:
            L.append(f"v_pk_add_f32 %[tB{j-1}], %[tB{j-1}], %[xb{(j-1) % 4}]")
        if 0 <= j - 2 < n:
            L.append(f"v_pk_fma_f32 %[accB], %[tB{j-2}], %[tB{j-2}], %[accB]")
    L.append("s_waitcnt vmcnt(0)")
    return "\\n\\t".join(L)


def screen_func(n: int, irbytes: int) -> str:
    outs = ['[accA] "+v"(accA)', '[accB] "+v"(accB)']
    outs += [f'[a{j}] "=&v"(a{j})' for j in range(4)]
    outs += [f'[xb{j}] "=&v"(xb{j})' for j in range(4)]
    outs += [f'[tA{j}] "=&v"(tA{j})' for j in range(n)]
    outs += [f'[tB{j}] "=&v"(tB{j})' for j in range(n)]
    outs += ['[xAn] "=&v"(xAn)', '[xBn] "=&v"(xBn)', '[rAn] "=&v"(rAn)', '[rBn] "=&v"(rBn)']
    ins = ['[xA] "v"(xA)', '[xB] "v"(xB)', '[roffA] "v"(roffA)', '[roffB] "v"(roffB)', '[koff] "v"(koff)',
           '[xbase] "s"(xbase)', '[rbase] "s"(rbase)', '[voxA] "v"(voxA)', '[voxB] "v"(voxB)',
           '[vorA] "v"(vorA)', '[vorB] "v"(vorB)']
    decl_t = ", ".join([f"tA{j}" for j in range(n)] + [f"tB{j}" for j in range(n)])
    return f"""template <>
__device__ __forceinline__ void screen2p<{n}, {irbytes}>(int koff, int roffA, double xA, int roffB, double xB,
    double& accA, double& accB, const void* xbase, const void* rbase, unsigned voxA, unsigned voxB, unsigned vorA,
    unsigned vorB, float& xAn, float& xBn, int& rAn, int& rBn)
{{
    int a0, a1, a2, a3;
    double xb0, xb1, xb2, xb3;
    double {decl_t};
    asm volatile("{screen_block(n, irbytes)}"
                 : {", ".join(outs)}
                 : {", ".join(ins)});
}}
"""


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    out = ["// GENERATED by gen_assign_steps.py -- do not edit.\n",
           "template <int N>\n__device__ __forceinline__ void steps2(int koff, int roffA, double xA, int roffB, "
           "double xB, double one,\n                                       double& accA, double& accB);\n",
           "template <int N, int IRBYTES>\n__device__ __forceinline__ void steps2p(int koff, int roffA, double xA, "
           "int roffB, double xB, double one,\n    double& accA, double& accB, const void* xbase, const void* rbase, "
           "unsigned voxA, unsigned voxB, unsigned vorA, unsigned vorB,\n    double& xAn, double& xBn, int& rAn, "
           "int& rBn);\n\n"]
    out += [func(n, False, 0) for n in range(1, 17)]
    for irb in (2, 4):
        out += [func(n, True, irb) for n in range(1, 17)]
    out.append("// f32 screen blocks: xA / xB carry the pair (float(x), float(x)) in a 64-bit register, accA / accB\n"
               "// the two f32 accumulators of the lane's centroid pair.\n"
               "template <int N, int IRBYTES>\n__device__ __forceinline__ void screen2p(int koff, int roffA, double xA, "
               "int roffB, double xB,\n    double& accA, double& accB, const void* xbase, const void* rbase, unsigned voxA, "
               "unsigned voxB, unsigned vorA,\n    unsigned vorB, float& xAn, float& xBn, int& rAn, int& rBn);\n\n")
    for irb in (2, 4):
        out += [screen_func(n, irb) for n in range(1, 17)]
    with open(os.path.join(here, "assign_steps.inc"), "w") as f:
        f.write("".join(out))


if __name__ == "__main__":
    main()

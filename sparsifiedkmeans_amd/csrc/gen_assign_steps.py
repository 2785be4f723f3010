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

def screen32_block(n: int, irbytes: int) -> str:
    """f32 SCREEN block, second generation: up to 32 entries of a point per block, two entries per DPP
    broadcast.  Lane l of a 16-lane row holds the pairs (x_2l, x_2l+1) [two f32 in one 64-bit register] and
    (roff_2l, roff_2l+1) [two LDS row offsets in one 64-bit register] of its point.  Per pair i:
        v[R:R+1] = rp[row lane i]                      v_mov_b64_dpp      (two row offsets at once)
        a = v[R] + koff ; t_2i   = LDS[a]              v_add_u32 (plain, 2 cycles) ; ds_read_b64
        a = v[R+1]+koff ; t_2i+1 = LDS[a]              v_add_u32 ; ds_read_b64
        xb = xp[row lane i]                            v_mov_b64_dpp      (x_2i, x_2i+1)
        t_2i   += (xb.lo, xb.lo)                       v_pk_add_f32 op_sel_hi:[1,0]
        t_2i+1 += (xb.hi, xb.hi)                       v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,1]
        acc = t*t + acc  (x2)                          v_pk_fma_f32
    i.e. 2 DPP + 2 plain + 4 packed instructions per 2 entries per 32 centroids (28 issue cycles instead of
    35).  The broadcast row offsets need their two halves as separate 32-bit operands, which inline-asm operands
    cannot express: they live in v[124:127], named literally and declared as clobbers (they never hold anything
    across the statement).  Entries 0..15 come from lanes 0..7, entries 16..31 from lanes 8..15; each half
    follows the two-group schedule of the first generation with at most 16 LDS reads in flight (waits are
    derived from the issue order below).  The next batch's global loads are issued after the first reads and
    waited for at the end of the statement."""
    L = ["s_waitcnt lgkmcnt(0)", "s_nop 1"]
    issued = []  # LDS reads in issue order (names)

    def wait_for(name):
        younger = len(issued) - 1 - issued.index(name)
        assert younger <= 15, (n, name, younger)
        L.append(f"s_waitcnt lgkmcnt({younger})")

    def issue_pair(grp, i, lane, nent, tbase):
        R = 124 + 2 * (i & 1)
        L.append(f"v_mov_b64_dpp v[{R}:{R+1}], %[rp{grp}] row_newbcast:{lane} {DPP}")
        for h in range(2):
            e = 2 * i + h
            if e < nent:
                L.append(f"v_add_u32 %[a{(2*i+h) % 4}], v{R+h}, %[koff]")
                L.append(f"ds_read_b64 %[t{grp}{e}], %[a{(2*i+h) % 4}]")
                issued.append(f"{grp}{tbase}{e}")

    first = True
    for half in range(2):
        nent = max(0, min(16, n - 16 * half))
        if nent == 0:
            break
        P = (nent + 1) // 2
        tb = f"h{half}_"
        for i in range(P):
            issue_pair("A", i, 8 * half + i, nent, tb)
        if first:
            ld2 = "global_load_ushort" if irbytes == 2 else "global_load_dword"
            L.append("global_load_dwordx2 %[xAn], %[voxA], %[xbase]")
            L.append("global_load_dwordx2 %[xBn], %[voxB], %[xbase]")
            if irbytes == 2:
                L.append(f"{ld2} %[rA0n], %[vorA], %[rbase]")
                L.append(f"{ld2} %[rA1n], %[vorA], %[rbase] offset:2")
                L.append(f"{ld2} %[rB0n], %[vorB], %[rbase]")
                L.append(f"{ld2} %[rB1n], %[vorB], %[rbase] offset:2")
            else:
                L.append(f"{ld2} %[rA0n], %[vorA], %[rbase]")
                L.append(f"{ld2} %[rA1n], %[vorA], %[rbase] offset:4")
                L.append(f"{ld2} %[rB0n], %[vorB], %[rbase]")
                L.append(f"{ld2} %[rB1n], %[vorB], %[rbase] offset:4")
            first = False
        for grp, other in (("A", "B"), ("B", None)):
            for i in range(P + 2):
                if i < P:
                    last_e = min(2 * i + 1, nent - 1)
                    wait_for(f"{grp}{tb}{last_e}")
                    L.append(f"v_mov_b64_dpp %[xb{i % 4}], %[xp{grp}] row_newbcast:{8 * half + i} {DPP}")
                    if other:
                        issue_pair(other, i, 8 * half + i, nent, tb)
                if 0 <= i - 1 < P:
                    e = 2 * (i - 1)
                    L.append(f"v_pk_add_f32 %[t{grp}{e}], %[t{grp}{e}], %[xb{(i-1) % 4}] op_sel_hi:[1,0]")
                    if e + 1 < nent:
                        L.append(f"v_pk_add_f32 %[t{grp}{e+1}], %[t{grp}{e+1}], %[xb{(i-1) % 4}] op_sel:[0,1] op_sel_hi:[1,1]")
                if 0 <= i - 2 < P:
                    e = 2 * (i - 2)
                    L.append(f"v_pk_fma_f32 %[acc{grp}], %[t{grp}{e}], %[t{grp}{e}], %[acc{grp}]")
                    if e + 1 < nent:
                        L.append(f"v_pk_fma_f32 %[acc{grp}], %[t{grp}{e+1}], %[t{grp}{e+1}], %[acc{grp}]")
    L.append("s_waitcnt vmcnt(0)")
    return "\\n\\t".join(L)


def screen32_func(n: int, irbytes: int) -> str:
    nt = min(n, 16)
    outs = ['[accA] "+v"(accA)', '[accB] "+v"(accB)']
    outs += [f'[a{j}] "=&v"(a{j})' for j in range(4)]
    outs += [f'[xb{j}] "=&v"(xb{j})' for j in range(4)]
    outs += [f'[tA{j}] "=&v"(tA{j})' for j in range(nt)]
    outs += [f'[tB{j}] "=&v"(tB{j})' for j in range(nt)]
    outs += ['[xAn] "=&v"(xAn)', '[xBn] "=&v"(xBn)', '[rA0n] "=&v"(rA0n)', '[rA1n] "=&v"(rA1n)',
             '[rB0n] "=&v"(rB0n)', '[rB1n] "=&v"(rB1n)']
    ins = ['[xpA] "v"(xpA)', '[xpB] "v"(xpB)', '[rpA] "v"(rpA)', '[rpB] "v"(rpB)', '[koff] "v"(koff)',
           '[xbase] "s"(xbase)', '[rbase] "s"(rbase)', '[voxA] "v"(voxA)', '[voxB] "v"(voxB)',
           '[vorA] "v"(vorA)', '[vorB] "v"(vorB)']
    decl_t = ", ".join([f"tA{j}" for j in range(nt)] + [f"tB{j}" for j in range(nt)])
    return f"""template <>
__device__ __forceinline__ void screen32p<{n}, {irbytes}>(int koff, double rpA, double xpA, double rpB, double xpB,
    double& accA, double& accB, const void* xbase, const void* rbase, unsigned voxA, unsigned voxB, unsigned vorA,
    unsigned vorB, double& xAn, double& xBn, int& rA0n, int& rA1n, int& rB0n, int& rB1n)
{{
    int a0, a1, a2, a3;
    double xb0, xb1, xb2, xb3;
    double {decl_t};
    asm volatile("{screen32_block(n, irbytes)}"
                 : {", ".join(outs)}
                 : {", ".join(ins)}
                 : "v124", "v125", "v126", "v127");
}}
"""



QX = 56   # first literal register of the quad rounds: x broadcasts v[QX:QX+7], read results v[QT:QT+31]
QT = QX + 8


def quad_round_block(nv: int, pl: int) -> str:
    """f32 SCREEN round of the 4-lanes-per-point kernel (screen.hip, k_screen_quad): nv <= 4 consecutive
    entries of a point, owned by lanes 0..nv-1 of its quad; pl = centroid PAIRS per lane (4: full tile of 32
    centroids, two 16-B reads; 2: 16 centroids, one 16-B read; 1: 8 centroids, one 8-B read; 5: a full tile
    plus ONE extra centroid per lane held in a second LDS region with 16-B rows).  Per entry m:
        a_m = off0 ^ ro[quad lane m]                   v_xor_b32_dpp quad_perm:[m,m,m,m]
                                                       (ro = row*128 | swizzle*16: the lane's 16-B piece l4 of a
                                                        half-row sits at piece l4 ^ ((row >> 1) & 3), which spreads
                                                        the COLUMN reads of the two-phase finish over 8 banks)
        b_m = a_m + delta                              v_add_u32            (pl >= 4: the other half-row,
                                                                             +64 or -64 by point slot)
        e_m = (a_m >> 7) * 16 + ce                     v_bfe_u32, v_lshl_add_u32  (pl = 5: row*16 + lane const)
        T   = LDS[a_m] (, LDS[b_m]) (, LDS[e_m])       ds_read_b128 / b64 (/ b32)
        x_m = xi[quad lane m]                          v_mov_b32_dpp
        T  += (x_m, x_m)   (pl pairs)                  v_pk_add_f32 op_sel_hi:[1,0]
        acc_j = T_j * T_j + acc_j   (j < pl)           v_pk_fma_f32
        E += x_m ; acc4 = E * E + acc4                 v_add_f32, v_fma_f32  (pl = 5)
    All reads of the round are issued first (up to 8 x 16 B + 4 x 4 B per lane in flight), then consumed in
    order.  The 128-bit read results and the broadcast x need their halves as separate operands, which
    inline-asm operands cannot express: they live in v[QX:QT+35], named literally and declared as clobbers."""
    L = ["s_nop 1"]
    two = pl >= 4
    for m in range(nv):
        L.append(f"v_xor_b32_dpp %[a{m}], %[ro], %[off0] quad_perm:[{m},{m},{m},{m}] row_mask:0xf bank_mask:0xf")
        T = QT + 8 * m
        if two:
            L.append(f"v_add_u32 %[b{m}], %[a{m}], %[delta]")
            L.append(f"ds_read_b128 v[{T}:{T+3}], %[a{m}]")
            L.append(f"ds_read_b128 v[{T+4}:{T+7}], %[b{m}]")
            if pl == 5:
                L.append(f"v_bfe_u32 %[e{m}], %[a{m}], 7, 16")
                L.append(f"v_lshl_add_u32 %[e{m}], %[e{m}], 4, %[ce]")
                L.append(f"ds_read_b32 v{QT + 32 + m}, %[e{m}]")
        elif pl == 2:
            L.append(f"ds_read_b128 v[{T}:{T+3}], %[a{m}]")
        else:
            L.append(f"ds_read_b64 v[{T}:{T+1}], %[a{m}]")
    for m in range(nv):
        L.append(f"v_mov_b32_dpp v{QX + 2 * m}, %[xi] quad_perm:[{m},{m},{m},{m}] row_mask:0xf bank_mask:0xf")
    per = (3 if pl == 5 else 2) if two else 1
    nreads = per * nv
    for m in range(nv):
        X = QX + 2 * m
        for h in range(2 if two else 1):
            T = QT + 8 * m + 4 * h
            L.append(f"s_waitcnt lgkmcnt({nreads - 1 - (per * m + h)})")
            L.append(f"v_pk_add_f32 v[{T}:{T+1}], v[{T}:{T+1}], v[{X}:{X+1}] op_sel_hi:[1,0]")
            if pl >= 2:
                L.append(f"v_pk_add_f32 v[{T+2}:{T+3}], v[{T+2}:{T+3}], v[{X}:{X+1}] op_sel_hi:[1,0]")
            L.append(f"v_pk_fma_f32 %[acc{2*h}], v[{T}:{T+1}], v[{T}:{T+1}], %[acc{2*h}]")
            if pl >= 2:
                L.append(f"v_pk_fma_f32 %[acc{2*h+1}], v[{T+2}:{T+3}], v[{T+2}:{T+3}], %[acc{2*h+1}]")
        if pl == 5:
            E = QT + 32 + m
            L.append(f"s_waitcnt lgkmcnt({nreads - 1 - (per * m + 2)})")
            L.append(f"v_add_f32 v{E}, v{E}, v{X}")
            L.append(f"v_fma_f32 %[acc4], v{E}, v{E}, %[acc4]")
    return "\\n\\t".join(L)


def quad_round_func(nv: int, pl: int) -> str:
    nacc = pl if pl < 4 else 4
    outs = [f'[acc{j}] "+v"(acc{j})' for j in range(nacc)]
    if pl == 5:
        outs.append('[acc4] "+v"(acc4)')
    outs += [f'[a{m}] "=&v"(a{m})' for m in range(nv)]
    if pl >= 4:
        outs += [f'[b{m}] "=&v"(b{m})' for m in range(nv)]
    if pl == 5:
        outs += [f'[e{m}] "=&v"(e{m})' for m in range(nv)]
    ins = ['[xi] "v"(xi)', '[ro] "v"(ro)', '[off0] "v"(off0)']
    if pl >= 4:
        ins.append('[delta] "v"(delta)')
    if pl == 5:
        ins.append('[ce] "v"(ce)')
    clob = ", ".join(f'"v{r}"' for r in range(QX, QT + 36))
    names = [f"a{m}" for m in range(nv)]
    if pl >= 4:
        names += [f"b{m}" for m in range(nv)]
    if pl == 5:
        names += [f"e{m}" for m in range(nv)]
    return f"""template <>
__device__ __forceinline__ void quad_round<{nv}, {pl}>(int xi, int ro, int off0, int delta, int ce,
    double& acc0, double& acc1, double& acc2, double& acc3, float& acc4)
{{
    int {", ".join(names)};
    asm volatile("{quad_round_block(nv, pl)}"
                 : {", ".join(outs)}
                 : {", ".join(ins)}
                 : {clob});
}}
"""


def own_round_block(nv: int, pl: int) -> str:
    """f32 SCREEN round of the OWN-REGISTER form of the 4-lanes-per-point kernel (screen_quad.hip, k_screen_own): every
    lane of a quad holds ALL entries of its point -- values as aligned pairs (x0, x1), (x2, x3), row offsets o_m =
    row * 128 | swizzle * 16 -- so nothing is broadcast: per entry two plain address ops instead of a DPP xor, an add
    and a DPP move (4 issue cycles instead of 10 next to the 32 of the packed arithmetic):
        a_m = o_m ^ off0 ; b_m = o_m ^ off1            v_xor_b32 x2           (pl >= 4; off1 = off0 ^ 64)
        T   = LDS[a_m], LDS[b_m]                       ds_read_b128 x2
        T  += (x_m, x_m)                               v_pk_add_f32, x_m picked out of its pair by op_sel
        acc_j = T_j * T_j + acc_j                      v_pk_fma_f32
    pl = 5: the lane's extra centroid for entries m, m + 1 is handled as ONE packed pair -- (c_m, c_m+1) + (x_m, x_m+1),
    squared into a two-sum accumulator -- so the remainder costs one packed add + fma per two entries.
    Temporaries and read results in literal registers v[QX:QT+39] (declared as clobbers)."""
    L = []
    two = pl >= 4
    for m in range(nv):
        T = QT + 8 * m
        A, B = QX + 2 * m, QX + 2 * m + 1
        L.append(f"v_xor_b32 v{A}, %[o{m}], %[off0]")
        if two:
            L.append(f"v_xor_b32 v{B}, %[o{m}], %[off1]")
            L.append(f"ds_read_b128 v[{T}:{T+3}], v{A}")
            L.append(f"ds_read_b128 v[{T+4}:{T+7}], v{B}")
        elif pl == 2:
            L.append(f"ds_read_b128 v[{T}:{T+3}], v{A}")
        else:
            L.append(f"ds_read_b64 v[{T}:{T+1}], v{A}")
    ne = 0
    if pl == 5:
        ne = (nv + 1) // 2 * 2  # entries of the extra centroid, in pairs (a slot past the column: x = 0 on the zero row)
        for m in range(ne):
            E = QT + 36 + m
            L.append(f"v_bfe_u32 v{E}, %[o{m}], 7, 16")
            L.append(f"v_lshl_add_u32 v{E}, v{E}, 4, %[ce]")
            L.append(f"ds_read_b32 v{QT + 32 + m}, v{E}")
    per = 2 if two else 1
    nreads = per * nv + ne
    done = 0
    for m in range(nv):
        xp = "%[xp01]" if m < 2 else "%[xp23]"
        sel = "op_sel_hi:[1,0]" if m % 2 == 0 else "op_sel:[0,1] op_sel_hi:[1,1]"
        for h in range(2 if two else 1):
            T = QT + 8 * m + 4 * h
            done += 1
            L.append(f"s_waitcnt lgkmcnt({nreads - done})")
            L.append(f"v_pk_add_f32 v[{T}:{T+1}], v[{T}:{T+1}], {xp} {sel}")
            if pl >= 2:
                L.append(f"v_pk_add_f32 v[{T+2}:{T+3}], v[{T+2}:{T+3}], {xp} {sel}")
            L.append(f"v_pk_fma_f32 %[acc{2*h}], v[{T}:{T+1}], v[{T}:{T+1}], %[acc{2*h}]")
            if pl >= 2:
                L.append(f"v_pk_fma_f32 %[acc{2*h+1}], v[{T+2}:{T+3}], v[{T+2}:{T+3}], %[acc{2*h+1}]")
    if pl == 5:
        L.append("s_waitcnt lgkmcnt(0)")
        for m in range(0, ne, 2):
            E = QT + 32 + m
            xp = "%[xp01]" if m < 2 else "%[xp23]"
            L.append(f"v_pk_add_f32 v[{E}:{E+1}], v[{E}:{E+1}], {xp}")
            L.append(f"v_pk_fma_f32 %[acc4], v[{E}:{E+1}], v[{E}:{E+1}], %[acc4]")
    return "\\n\\t".join(L)


def own_round_func(nv: int, pl: int) -> str:
    nacc = pl if pl < 4 else 4
    outs = [f'[acc{j}] "+v"(acc{j})' for j in range(nacc)]
    if pl == 5:
        outs.append('[acc4] "+v"(acc4)')
    nused = nv if pl != 5 else (nv + 1) // 2 * 2
    ins = ['[xp01] "v"(xp01)']
    if nused > 2:
        ins.append('[xp23] "v"(xp23)')
    ins += [f'[o{m}] "v"(o{m})' for m in range(nused)]
    ins.append('[off0] "v"(off0)')
    if pl >= 4:
        ins.append('[off1] "v"(off1)')
    if pl == 5:
        ins.append('[ce] "v"(ce)')
    clob = ", ".join(f'"v{r}"' for r in range(QX, QT + 40))
    return f"""template <>
__device__ __forceinline__ void quad_round_own<{nv}, {pl}>(double xp01, double xp23, int o0, int o1, int o2, int o3,
    int off0, int off1, int ce, double& acc0, double& acc1, double& acc2, double& acc3, double& acc4)
{{
    asm volatile("{own_round_block(nv, pl)}"
                 : {", ".join(outs)}
                 : {", ".join(ins)}
                 : {clob});
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
    out.append("// second-generation f32 screen blocks: up to 32 entries, two per DPP broadcast (see gen_assign_steps.py)\n"
               "template <int N, int IRBYTES>\n__device__ __forceinline__ void screen32p(int koff, double rpA, double xpA, "
               "double rpB, double xpB,\n    double& accA, double& accB, const void* xbase, const void* rbase, "
               "unsigned voxA, unsigned voxB, unsigned vorA,\n    unsigned vorB, double& xAn, double& xBn, int& rA0n, "
               "int& rA1n, int& rB0n, int& rB1n);\n\n")
    for irb in (2, 4):
        out += [screen32_func(n, irb) for n in range(1, 33)]
    with open(os.path.join(here, "assign_steps.inc"), "w") as f:
        f.write("".join(out))
    # the quad rounds go to their own file: screen_quad.hip is compiled as separate translation units
    out = ["// GENERATED by gen_assign_steps.py -- do not edit.\n",
           "// rounds of the 4-lanes-per-point f32 screen (see gen_assign_steps.py, quad_round_block)\n"
           "template <int NV, int PL>\n__device__ __forceinline__ void quad_round(int xi, int ro, int off0, "
           "int delta, int ce,\n    double& acc0, double& acc1, double& acc2, double& acc3, float& acc4);\n\n"]
    for pl in (1, 2, 4, 5):
        out += [quad_round_func(nv, pl) for nv in range(1, 5)]
    out.append("// rounds of the own-register form (see gen_assign_steps.py, own_round_block)\n"
               "template <int NV, int PL>\n__device__ __forceinline__ void quad_round_own(double xp01, double xp23, int o0, "
               "int o1, int o2, int o3,\n    int off0, int off1, int ce, double& acc0, double& acc1, double& acc2, "
               "double& acc3, double& acc4);\n\n")
    for pl in (1, 2, 4, 5):
        out += [own_round_func(nv, pl) for nv in range(1, 5)]
    with open(os.path.join(here, "quad_steps.inc"), "w") as f:
        f.write("".join(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Writes tools/ubench_quad_variants.inc: the round of the 4-lanes-per-point f32 screen (csrc/gen_assign_steps.py,
quad_round_block / own_round_block) with parts left out or rearranged, for tools/ubench_quad.hip:
  bc_full / bc_noreads / bc_nomath / bc_noaddr      the broadcast form (what k_screen_quad runs), without its LDS reads,
                                                    without its packed arithmetic, without its address / broadcast ops
  own_full / own_noreads / own_nomath               every lane keeps all entries of its point: no DPP broadcast
  own_pipe_*                                        the same, the next round's reads in flight during this round's arithmetic
  bc_mix16                                          f16 tile of 64 centroids, v_fma_mix_f32 + v_fma_f32 per centroid
  bc_pk16                                           f16 tile of 32 centroids in 64-B rows, v_pk_add_f16 + v_pk_fma_f16 (round 5)
Run from the repo root:  python tools/gen_ubench_quad_variants.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "sparsifiedkmeans_amd", "csrc"))
import gen_assign_steps as G  # noqa: E402

QX, QT = G.QX, G.QT
SEP = "\\n\\t"


def variant(block, drop):
    return SEP.join(l for l in block.split(SEP) if not any(l.startswith(d) for d in drop))


def func(name, block, own):
    clob = ", ".join(f'"v{r}"' for r in range(QX, QT + 40))
    if own:
        return (f"__device__ __forceinline__ void {name}(double xp01, double xp23, int o0, int o1, int o2, int o3, int off0, "
                f"int off1, int ce, double& acc0, double& acc1, double& acc2, double& acc3, double& acc4)\n"
                f'{{ asm volatile("{block}" : [acc0] "+v"(acc0), [acc1] "+v"(acc1), [acc2] "+v"(acc2), [acc3] "+v"(acc3) : '
                f'[xp01] "v"(xp01), [xp23] "v"(xp23), [o0] "v"(o0), [o1] "v"(o1), [o2] "v"(o2), [o3] "v"(o3), [off0] "v"(off0), '
                f'[off1] "v"(off1) : {clob}); }}\n')
    return (f"__device__ __forceinline__ void {name}(int xi, int ro, int off0, int delta, int ce, double& acc0, double& acc1, "
            f"double& acc2, double& acc3, float& acc4)\n"
            f'{{ int a0, a1, a2, a3, b0, b1, b2, b3; asm volatile("{block}" : [acc0] "+v"(acc0), [acc1] "+v"(acc1), '
            f'[acc2] "+v"(acc2), [acc3] "+v"(acc3), [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3), '
            f'[b0] "=&v"(b0), [b1] "=&v"(b1), [b2] "=&v"(b2), [b3] "=&v"(b3) : [xi] "v"(xi), [ro] "v"(ro), [off0] "v"(off0), '
            f'[delta] "v"(delta) : {clob}); }}\n')


SET = [QT, 104]


def pipe_block(ph, first, last):
    L = []
    cur, nxt = SET[ph], SET[1 - ph]

    def issue(base, opfx):
        out = []
        for m in range(4):
            T = base + 8 * m
            A, B = QX + 2 * m, QX + 2 * m + 1
            out += [f"v_xor_b32 v{A}, %[{opfx}{m}], %[off0]", f"v_xor_b32 v{B}, %[{opfx}{m}], %[off1]",
                    f"ds_read_b128 v[{T}:{T+3}], v{A}", f"ds_read_b128 v[{T+4}:{T+7}], v{B}"]
        return out
    if first:
        L += issue(cur, "c")
    n_next = 0
    if not last:
        L += issue(nxt, "o")
        n_next = 8
    done = 0
    for m in range(4):
        xp = "%[xp01]" if m < 2 else "%[xp23]"
        sel = "op_sel_hi:[1,0]" if m % 2 == 0 else "op_sel:[0,1] op_sel_hi:[1,1]"
        for h in range(2):
            T = cur + 8 * m + 4 * h
            done += 1
            L += [f"s_waitcnt lgkmcnt({min(15, n_next + 8 - done)})",
                  f"v_pk_add_f32 v[{T}:{T+1}], v[{T}:{T+1}], {xp} {sel}", f"v_pk_add_f32 v[{T+2}:{T+3}], v[{T+2}:{T+3}], {xp} {sel}",
                  f"v_pk_fma_f32 %[acc{2*h}], v[{T}:{T+1}], v[{T}:{T+1}], %[acc{2*h}]",
                  f"v_pk_fma_f32 %[acc{2*h+1}], v[{T+2}:{T+3}], v[{T+2}:{T+3}], %[acc{2*h+1}]"]
    return SEP.join(L)


def pipe_func(ph, first, last):
    clob = ", ".join(f'"v{r}"' for r in list(range(QX, QT + 32)) + list(range(104, 136)))
    ins = ['[xp01] "v"(xp01)', '[xp23] "v"(xp23)', '[off0] "v"(off0)', '[off1] "v"(off1)']
    args = "double xp01, double xp23, int off0, int off1, double& acc0, double& acc1, double& acc2, double& acc3"
    if first:
        ins += [f'[c{m}] "v"(c{m})' for m in range(4)]
        args += ", int c0, int c1, int c2, int c3"
    if not last:
        ins += [f'[o{m}] "v"(o{m})' for m in range(4)]
        args += ", int o0, int o1, int o2, int o3"
    return (f"__device__ __forceinline__ void own_pipe_{ph}_{int(first)}_{int(last)}({args})\n"
            f'{{ asm volatile("{pipe_block(ph, first, last)}" : [acc0] "+v"(acc0), [acc1] "+v"(acc1), [acc2] "+v"(acc2), '
            f'[acc3] "+v"(acc3) : {", ".join(ins)} : {clob}); }}\n')


def mix_block():
    L = ["s_nop 1"]
    for m in range(4):
        T = QT + 8 * m
        L += [f"v_xor_b32_dpp %[a{m}], %[ro], %[off0] quad_perm:[{m},{m},{m},{m}] row_mask:0xf bank_mask:0xf",
              f"v_add_u32 %[b{m}], %[a{m}], %[delta]", f"ds_read_b128 v[{T}:{T+3}], %[a{m}]", f"ds_read_b128 v[{T+4}:{T+7}], %[b{m}]"]
    for m in range(4):
        L.append(f"v_mov_b32_dpp v{QX + 2 * m}, %[xi] quad_perm:[{m},{m},{m},{m}] row_mask:0xf bank_mask:0xf")
    done = 0
    for m in range(4):
        X = QX + 2 * m
        for h in range(2):
            done += 1
            L.append(f"s_waitcnt lgkmcnt({8 - done})")
            for w in range(4):
                R = QT + 8 * m + 4 * h + w
                for half in range(2):
                    j = h * 8 + w * 2 + half
                    t = R if half else 104 + (m * 8 + h * 4 + w) % 8
                    L += [f"v_fma_mix_f32 v{t}, v{R}, 1.0, v{X} op_sel:[{half},0,0] op_sel_hi:[1,0,0]", f"v_fma_f32 %[c{j}], v{t}, v{t}, %[c{j}]"]
    return SEP.join(L)


def pk16_block():
    """packed-f16 round (round 5's capped experiment): tile rows of 32 f16 centroids = 64 B, ONE ds_read_b128 per entry and lane
    (8 centroids), v_pk_add_f16 + v_pk_fma_f16 on packed halves, f16 accumulators"""
    L = ["s_nop 1"]
    for m in range(4):
        T = QT + 4 * m
        L += [f"v_xor_b32_dpp %[a{m}], %[ro], %[off0] quad_perm:[{m},{m},{m},{m}] row_mask:0xf bank_mask:0xf",
              f"ds_read_b128 v[{T}:{T+3}], %[a{m}]"]
    for m in range(4):
        L.append(f"v_mov_b32_dpp v{QX + m}, %[xi] quad_perm:[{m},{m},{m},{m}] row_mask:0xf bank_mask:0xf")
    for m in range(4):
        T = QT + 4 * m
        L.append(f"s_waitcnt lgkmcnt({3 - m})")
        for w in range(4):
            L.append(f"v_pk_add_f16 v{T + w}, v{T + w}, v{QX + m}")
        for w in range(4):
            L.append(f"v_pk_fma_f16 %[c{w}], v{T + w}, v{T + w}, %[c{w}]")
    return SEP.join(L)


def main():
    b, o = G.quad_round_block(4, 4), G.own_round_block(4, 4)
    noaddr = variant(b, ["v_xor_b32_dpp", "v_add_u32", "v_mov_b32_dpp"])
    for m in range(4):
        noaddr = noaddr.replace(f"%[a{m}]", "%[ro]").replace(f"%[b{m}]", "%[off0]")
    out = ["// GENERATED by tools/gen_ubench_quad_variants.py for tools/ubench_quad.hip -- do not edit.\n",
           func("bc_full", b, False), func("bc_noreads", variant(b, ["ds_read", "s_waitcnt"]), False),
           func("bc_nomath", variant(b, ["v_pk_"]), False), func("bc_noaddr", noaddr, False),
           func("own_full", o, True), func("own_noreads", variant(o, ["ds_read", "s_waitcnt"]), True),
           func("own_nomath", variant(o, ["v_pk_"]), True),
           pipe_func(0, True, False), pipe_func(1, False, False), pipe_func(0, False, False), pipe_func(0, False, True),
           pipe_func(1, False, True)]
    clob = ", ".join(f'"v{r}"' for r in list(range(QX, QT + 32)) + list(range(104, 112)))
    accs = ", ".join(f'[c{j}] "+v"(c[{j}])' for j in range(16))
    out.append(f"__device__ __forceinline__ void bc_mix16(int xi, int ro, int off0, int delta, float (&c)[16])\n"
               f'{{ int a0, a1, a2, a3, b0, b1, b2, b3; asm volatile("{mix_block()}" : {accs}, [a0] "=&v"(a0), [a1] "=&v"(a1), '
               f'[a2] "=&v"(a2), [a3] "=&v"(a3), [b0] "=&v"(b0), [b1] "=&v"(b1), [b2] "=&v"(b2), [b3] "=&v"(b3) : [xi] "v"(xi), '
               f'[ro] "v"(ro), [off0] "v"(off0), [delta] "v"(delta) : {clob}); }}\n')
    clob16 = ", ".join(f'"v{r}"' for r in range(QX, QT + 16))
    out.append(f"__device__ __forceinline__ void bc_pk16(int xi, int ro, int off0, int& c0, int& c1, int& c2, int& c3)\n"
               f'{{ int a0, a1, a2, a3; asm volatile("{pk16_block()}" : [c0] "+v"(c0), [c1] "+v"(c1), [c2] "+v"(c2), [c3] "+v"(c3), '
               f'[a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3) : [xi] "v"(xi), [ro] "v"(ro), [off0] "v"(off0) : {clob16}); }}\n')
    with open(os.path.join(ROOT, "tools", "ubench_quad_variants.inc"), "w") as f:
        f.write("".join(out))


if __name__ == "__main__":
    main()

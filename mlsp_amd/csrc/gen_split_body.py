#!/usr/bin/env python3
"""Emits the hand-placed K-tile body of gemm_split_kernel (gemm.hip): straight-line code, one MFMA per issue slot with its fillers, every
slot closed by sched_barrier(0) so the compiler keeps the placement.

    python3 gen_split_body.py 2 > gemm_split_body_wm2.inc        (128 x 128 tile)
    python3 gen_split_body.py 1 > gemm_split_body_wm1.inc        ( 64 x 128 tile)

Per K-tile (32 deep) a wave issues 24 * WM MFMAs (2 k16 steps x WM x 2 accumulators x 6 piece products).  The MFMA shadow is 32 cycles = 8
issue slots of 4; between two MFMAs sit one (WM = 1: one or two) micro-steps of the three-way bf16 split of the NEXT tile's operands
(16 / 12 element pairs x 3 micro-steps), and after each finished quad its three image writes and the global load of the quad after next.
Names used: acc, a[s2][i][q], b[s2][j][q], raw[8], pk0/pk1/pk2[2], r0, r1, a1, wa, wb, WQA, WQB, SX_LOAD_A/B, SX_XF_A/B (see the kernel)."""
import sys
WM = int(sys.argv[1])
QA = [1, 0, 2, 0, 1, 0]          # smallest products first, the leading one last
QB = [1, 2, 0, 1, 0, 0]
NQA = 2 * WM                      # A quads of a tile per thread (B: 4)
out = []
def micro(m):
    pr, ms = m // 3, m % 3
    qd, hh = pr >> 1, pr & 1
    x0, x1 = f"raw[{qd}][{2 * hh}]", f"raw[{qd}][{2 * hh + 1}]"
    L = []
    if ms == 0:
        # operand transform (gemm_split_kernel<.., XF, XD>): this pair of the staged quad becomes the previous layer's activated output
        # before it is split; SX_XF_A / SX_XF_B expand to nothing in the instantiations without a transform on that operand
        L += [f"SX_XF_A({qd}, {hh});" if qd < NQA else f"SX_XF_B({qd - NQA}, {hh});"]
        L += [f"pk0[{hh}] = sx_cvt_pk({x0}, {x1});",
              f"a1 = __uint_as_float(pk0[{hh}] & 0xffff0000u);",
              f"r0 = {x0} - __uint_as_float(pk0[{hh}] << 16);"]
    elif ms == 1:
        L += [f"r1 = {x1} - a1;", f"pk1[{hh}] = sx_cvt_pk(r0, r1);"]
    else:
        L += [f"pk2[{hh}] = sx_cvt_pk(r0 - __uint_as_float(pk1[{hh}] << 16), r1 - __uint_as_float(pk1[{hh}] & 0xffff0000u));"]
        if hh == 1:
            d = f"wa + {qd} * WQA" if qd < NQA else f"wb + {qd - NQA} * WQB"
            for q in range(3):
                L.append(f"*(u32x2*)({d} + {q} * SX_PLANE) = (u32x2){{pk{q}[0], pk{q}[1]}};")
            L.append(f"raw[{qd}] = SX_LOAD_A({qd});" if qd < NQA else f"raw[{qd}] = SX_LOAD_B({qd - NQA});")
    return L
S = 24 * WM
MS = 3 * 2 * (NQA + 4)
for c in range(S):
    s2, rest = c // (12 * WM), c % (12 * WM)
    ij, p6 = rest // 6, rest % 6
    i, j = ij >> 1, ij & 1
    out.append(f"// slot {c}")
    out.append(f"acc[{i}][{j}] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[{s2}][{i}][{QA[p6]}], b[{s2}][{j}][{QB[p6]}], acc[{i}][{j}], 0, 0, 0);")
    for m in range(c * MS // S, (c + 1) * MS // S):
        out += micro(m)
    out.append("__builtin_amdgcn_sched_barrier(0);")
print("\n".join(out))

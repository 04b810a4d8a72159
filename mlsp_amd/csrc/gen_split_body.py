#!/usr/bin/env python3
"""Emits the hand-placed K-tile bodies of gemm_split_kernel (gemm.hip): straight-line code, one MFMA per issue slot with its fillers, every
slot closed by sched_barrier(0) so the compiler keeps the placement.

    python3 gen_split_body.py WM [VARIANT] > gemm_split_body_wm<WM>[_<VARIANT>].inc      (WM = 2: 128 x 128 tile, 1: 64 x 128)

Per K-tile (32 deep) a wave issues 24 * WM MFMAs (2 k16 steps x WM x 2 accumulators x 6 piece products).  The MFMA shadow is 32 cycles = 8
issue slots of 4; between two MFMAs sits vector work of the NEXT tile's staging: per staged quad (16-byte load) two element pairs x three
micro-steps of the three-way bf16 split, then its three image writes and the global load of the quad after next.

VARIANT (operand transform instantiations, gemm_split_kernel<.., XF, XD, DY>): dy / dyxb / dyxbd = the A quads are (d', y) pairs turned into
the layer's output gradient (SX_DY_A) [+ the xb / xbd transform on B]; xa / xad = the A quads are transformed before their split
(SX_XF_A: scale * x + shift, activation as one max; `d`: + dropout, one hash per quad), xb / xbd = the same on the B quads (SX_XF_B).
The transform roughly doubles the vector work of the quads it applies to; placing it with the split steps of those quads overflows the MFMA
shadow of a third of the slots (measured: +27 % per launch).  In the variants the work items keep their ORDER but are spread over the slots
by WEIGHT (approximate vector instructions), so every slot carries the same load whichever operand is transformed.
The plain body (no VARIANT) is the round-3 placement, unchanged.

Names used: acc, a[s2][i][q], b[s2][j][q], raw[8], pk0/pk1/pk2[2], r0, r1, a1, wa, wb, WQA, WQB, SX_LOAD_A/B, SX_XF_A/B, SX_XF_HASH_A/B (see the kernel)."""
import sys
WM = int(sys.argv[1])
VAR = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else ""
HALF = len(sys.argv) > 3 and sys.argv[3] == "h"      # two f16 pieces, three products (gemm_split_kernel<.., NPC = 2>): see items()
QA = [1, 0, 2, 0, 1, 0]          # smallest products first, the leading one last
QB = [1, 2, 0, 1, 0, 0]
if HALF:
    QA, QB = [1, 0, 0], [0, 1, 0]
NP6 = len(QA)
NQA = 2 * WM                      # A quads of a tile per thread (B: 4)
S = 4 * NP6 * WM
DYA = VAR.startswith("dy")        # dy / dyxb / dyxbd: the A quads are d' + y pairs (SX_DY_A: the BatchNorm backward, 3 operations per value; SX_DY_LOAD)
XA = VAR.startswith("xa")
XB = VAR.startswith("xb") or VAR.startswith("dyxb")
DROP = VAR.endswith("d")
# HALF: the K-tile body is cut in two phases by a barrier.  The fragments of BOTH k16 steps are requested before the body; phase 1 (the
# MFMAs of step 0) only waits for step 0's and stages the first Q1 quads in registers while step 1's fragments arrive; at the barrier every
# wave holds all its fragments, and phase 2 (the MFMAs of step 1) writes the images -- the deferred quads' first.  (Before: all 16 fragment
# reads were waited for in front of the first MFMA.)
Q1 = (3 * (NQA + 4) + 4) // 5 if HALF else 0
WRITES = []


def items():
    """ordered work items (weight, [statements]) of one K-tile's staging"""
    out = []
    for qd in range(NQA + 4):
        isA = qd < NQA
        xf = (XA and isA) or (XB and not isA)
        q = qd if isA else qd - NQA
        op = "A" if isA else "B"
        if xf and DROP:
            out.append((10, [f"SX_XF_HASH_{op}({q});"], qd))
        for hh in range(2):
            x0, x1 = f"raw[{qd}][{2 * hh}]", f"raw[{qd}][{2 * hh + 1}]"
            if DYA and isA:
                out.append((6, [f"SX_DY_A({q}, {hh});"], qd))
            if xf:
                out.append((10 if DROP else 6, [f"SX_XF_{op}({q}, {hh});"], qd))
            if HALF:
                # x * s = h0 + h1 (+ < 2^-22 |x s|): h0 = f16(x s), h1 = f16(x s - h0) (the remainder is exact in fp32); s: the operand's
                # power-of-two scale (sxs_a / sxs_b).  Per pair: v_pk_mul (or 2 v_mul), v_cvt_pk_f16_f32, 2 v_fma_mix_f32 (x s - h0, reading the f16
                # half in place), v_cvt_pk_f16_f32
                sc = "sxs_a" if isA else "sxs_b"
                # quads staged in phase 1 (before the mid-body barrier: see the placement below) keep their pieces in pkd[qd] and are
                # written to the images in phase 2; the others use the scratch pair pk0 / pk1 and are written at once
                early = qd < Q1
                P0, P1 = (f"pkd[{qd}][0]", f"pkd[{qd}][1]") if early else ("pk0", "pk1")
                out.append((3, [f"{P0}[{hh}] = sx_cvt_pk_h({x0} * {sc}, {x1} * {sc});"], qd))
                L = [f"{P1}[{hh}] = sx_cvt_pk_h(sx_rem_lo({x0}, {sc}, {P0}[{hh}]), sx_rem_hi({x1}, {sc}, {P0}[{hh}]));"]
                w = 3
                if hh == 1:
                    d = f"wa + {qd} * WQA" if isA else f"wb + {q} * WQB"
                    W = [f"*(u32x2*)({d} + {p} * SX_PLANE) = (u32x2){{{(P0, P1)[p]}[0], {(P0, P1)[p]}[1]}};" for p in range(2)]
                    if early:
                        WRITES.append((2, W))
                    else:
                        L += W
                        w += 2
                    L.append(f"raw[{qd}] = SX_LOAD_{op}({q});")
                    if DYA and isA:
                        L.append(f"SX_DY_LOAD({q});")
                    w += 1
                out.append((w, L, qd))
                continue
            out.append((4, [f"pk0[{hh}] = sx_cvt_pk({x0}, {x1});", f"a1 = __uint_as_float(pk0[{hh}] & 0xffff0000u);",
                            f"r0 = {x0} - __uint_as_float(pk0[{hh}] << 16);"], qd))
            out.append((2, [f"r1 = {x1} - a1;", f"pk1[{hh}] = sx_cvt_pk(r0, r1);"], qd))
            L = [f"pk2[{hh}] = sx_cvt_pk(r0 - __uint_as_float(pk1[{hh}] << 16), r1 - __uint_as_float(pk1[{hh}] & 0xffff0000u));"]
            w = 5
            if hh == 1:
                d = f"wa + {qd} * WQA" if isA else f"wb + {q} * WQB"
                for p in range(3):
                    L.append(f"*(u32x2*)({d} + {p} * SX_PLANE) = (u32x2){{pk{p}[0], pk{p}[1]}};")
                L.append(f"raw[{qd}] = SX_LOAD_{op}({q});")
                if DYA and isA:
                    L.append(f"SX_DY_LOAD({q});")
                w += 4
            out.append((w, L, qd))
    return out


def mfma(c):
    s2, rest = c // (2 * NP6 * WM), c % (2 * NP6 * WM)
    ij, p6 = rest // NP6, rest % NP6
    i, j = ij >> 1, ij & 1
    if HALF:
        return (f"acc[{i}][{j}] = __builtin_amdgcn_mfma_f32_32x32x16_f16(SXH(a[{s2}][{i}][{QA[p6]}]), SXH(b[{s2}][{j}][{QB[p6]}]), "
                f"acc[{i}][{j}], 0, 0, 0);")
    return f"acc[{i}][{j}] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[{s2}][{i}][{QA[p6]}], b[{s2}][{j}][{QB[p6]}], acc[{i}][{j}], 0, 0, 0);"


out = []
if not VAR and not HALF:
    # the round-3 placement: 3 * 2 * (NQA + 4) micro-steps spread evenly BY COUNT (one per slot at WM = 2, one or two at WM = 1);
    # SX_XF_A / SX_XF_B (which expand to nothing in the plain kernel) ride with the first micro-step of their pair
    def micro(m):
        pr, ms = m // 3, m % 3
        qd, hh = pr >> 1, pr & 1
        x0, x1 = f"raw[{qd}][{2 * hh}]", f"raw[{qd}][{2 * hh + 1}]"
        L = []
        if ms == 0:
            L += [f"pk0[{hh}] = sx_cvt_pk({x0}, {x1});", f"a1 = __uint_as_float(pk0[{hh}] & 0xffff0000u);",
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
    MS = 3 * 2 * (NQA + 4)
    for c in range(S):
        out.append(f"// slot {c}")
        out.append(mfma(c))
        for m in range(c * MS // S, (c + 1) * MS // S):
            out += micro(m)
        out.append("__builtin_amdgcn_sched_barrier(0);")
elif HALF:
    its = items()
    ph = [[(w, L) for w, L, qd in its if qd < Q1], WRITES + [(w, L) for w, L, qd in its if qd >= Q1]]
    H = S // 2
    for phase in range(2):
        if phase == 1:
            out.append("__syncthreads();     // every wave holds the fragments of both k16 steps: the images may be overwritten")
        lst = ph[phase]
        total = sum(w for w, _ in lst)
        k, done = 0, 0
        for c in range(H):
            out.append(f"// slot {phase * H + c}")
            out.append(mfma(phase * H + c))
            while k < len(lst) and (done + lst[k][0] / 2.0) * H < (c + 1) * total:
                out += lst[k][1]
                done += lst[k][0]
                k += 1
            out.append("__builtin_amdgcn_sched_barrier(0);")
        assert k == len(lst)
else:
    its = items()
    total = sum(w for w, _, _ in its)
    k, done = 0, 0
    for c in range(S):
        out.append(f"// slot {c}")
        out.append(mfma(c))
        # items whose midpoint falls inside this slot's share of the total weight
        while k < len(its) and (done + its[k][0] / 2.0) * S < (c + 1) * total:
            out += its[k][1]
            done += its[k][0]
            k += 1
        out.append("__builtin_amdgcn_sched_barrier(0);")
    assert k == len(its)
print("\n".join(out))

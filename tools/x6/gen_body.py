#!/usr/bin/env python3
"""Emits the hand-placed K-tile body of the double-buffered split GEMM (x6_kernel_v4): straight-line code, one MFMA per slot with its
fillers, every slot closed by sched_barrier(0).   usage: gen_body.py CB > x6_body_cbNN.inc
   slots 0..47      : the MFMAs of tile t (k16 step c // 24, accumulator (c % 24) // 6, piece product c % 6)
   slots 0..11      : + one fragment read of the tile's second k16 step
   slots 0..CB-1    : + the 48 split micro-steps of tile t+1 (16 element pairs x 3), image writes and the global load of tile t+2 after
                      every finished quad
   after slot CB-1  : barrier
   slots CB..47     : + the first-k16-step fragments of tile t+1"""
import sys
CB = int(sys.argv[1])
QA = [1, 0, 2, 0, 1, 0]
QB = [1, 2, 0, 1, 0, 0]
out = []
def frag(dst_s2, f, buf):
    q, i2 = f % 3, (f // 3) & 1
    if f < 6:
        return f"if (!(ABL & 8)) a[{dst_s2}][{i2}][{q}] = x6_frag<KA>({buf} + {q} * X6_PLANE, fa, {32 * i2}, {dst_s2});"
    return f"if (!(ABL & 8)) b[{dst_s2}][{i2}][{q}] = x6_frag<KB>({buf} + {q} * X6_PLANE, fb, {32 * i2}, {dst_s2});"
def micro(m):
    pr, ms = m // 3, m % 3
    qd, hh = pr >> 1, pr & 1
    x0, x1 = f"raw[{qd}][{2 * hh}]", f"raw[{qd}][{2 * hh + 1}]"
    L = []
    if ms == 0:
        L += [f"if (ABL & 1) {{ pk0[{hh}] = __float_as_uint({x0}); pk1[{hh}] = __float_as_uint({x1}); pk2[{hh}] = pk0[{hh}]; }} else {{",
              f"pk0[{hh}] = x6_cvt_pk({x0}, {x1});",
              f"a1 = __uint_as_float(pk0[{hh}] & 0xffff0000u);",
              f"r0 = {x0} - __uint_as_float(pk0[{hh}] << 16); }}"]
    elif ms == 1:
        L += [f"if (!(ABL & 1)) {{ r1 = {x1} - a1;", f"pk1[{hh}] = x6_cvt_pk(r0, r1); }}"]
    else:
        L += [f"if (!(ABL & 1)) pk2[{hh}] = x6_cvt_pk(r0 - __uint_as_float(pk1[{hh}] << 16), r1 - __uint_as_float(pk1[{hh}] & 0xffff0000u));"]
        if hh == 1:
            d = f"nxt + woa + {qd} * WQA" if qd < 4 else f"nxt + wob + {qd - 4} * WQB"
            for q in range(3):
                L.append(f"if (!(ABL & 2)) *(u32x2_*)({d} + {q} * X6_PLANE) = (u32x2_){{pk{q}[0], pk{q}[1]}};")
            if True:
                L.append(f"if (ABL & 2) {{ junk ^= pk0[0] ^ pk0[1] ^ pk1[0] ^ pk1[1] ^ pk2[0] ^ pk2[1]; }}")
            src = f"pa + {qd} * qa_" if qd < 4 else f"pb + {qd - 4} * qb_"
            L.append(f"if (!(ABL & 4)) raw[{qd}] = *(const f32x4*)({src});")
    return L
nrd = 48 - CB
per = (12 + nrd - 1) // nrd
for c in range(48):
    s2, ij, p6 = c // 24, (c % 24) // 6, c % 6
    i, j = ij >> 1, ij & 1
    out.append(f"// slot {c}")
    out.append(f"acc[{i}][{j}] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[{s2}][{i}][{QA[p6]}], b[{s2}][{j}][{QB[p6]}], acc[{i}][{j}], 0, 0, 0);")
    if c < 12:
        out.append(frag(1, c, "cur"))
    if c < CB:
        for m in range(c * 48 // CB, (c + 1) * 48 // CB):
            out += micro(m)
    if c == CB - 1:
        out.append("__builtin_amdgcn_sched_barrier(0);")
        out.append("if (!(ABL & 16)) __syncthreads();")
    if c >= CB:
        for f in range((c - CB) * per, min(12, (c - CB + 1) * per)):
            out.append(frag(0, f, "nxt"))
    out.append("__builtin_amdgcn_sched_barrier(0);")
print("\n".join(out))

#!/usr/bin/env python3
"""Where does the ping-pong GEMM's time go?  The stack's GEMM shapes at c2, timed with HIP events (cold caches: 512 MB fill
between launches); run once per library build (VLHIP_LIBRARY=.../ab_nodma/libvlhip.so etc.: timing experiments with wrong
results -- no operand DMA in the K loop / no LDS fragment reads / neither).  Usage: python3 tools/gemm_exp.py"""
import sys

import torch

sys.path.insert(0, ".")
from clg_vqa_amd import ops  # noqa: E402
from clg_vqa_amd.ops import BF16, EPI_BF16, EPI_DGELU_BF16, EPI_F32, EPI_GELU_SPLIT, EPI_SPLIT  # noqa: E402

DEV = "cuda"
M, H, I = 14336, 768, 3072


def med(fn, n=10):
    filler = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
    ts = []
    for _ in range(n):
        filler.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    g = torch.Generator().manual_seed(0)
    # GEMM_EXP_DATA=const: every operand element the same value (the matrix pipe's power draw depends on how many operand bits
    # toggle: with constant operands the same instruction stream runs at a higher sustained clock); default: normal random
    import os
    if os.environ.get("GEMM_EXP_DATA", "random") == "const":
        mk = lambda r, c: torch.full((r, c), 0.0625, device=DEV, dtype=BF16)  # noqa: E731
    else:
        mk = lambda r, c: (torch.randn(r, c, generator=g) * 0.1).to(DEV).to(BF16)  # noqa: E731
    xh, xl, hh, hl = mk(M, H), mk(M, H), mk(M, I), mk(M, I)
    wq, wql, w1, w1l, w2, w2l = mk(3 * H, H), mk(3 * H, H), mk(I, H), mk(I, H), mk(H, I), mk(H, I)
    b3, bI, bH = torch.zeros(3 * H, device=DEV), torch.zeros(I, device=DEV), torch.zeros(H, device=DEV)
    o16a, o16b, o16c = (torch.empty(M, I, dtype=BF16, device=DEV) for _ in range(3))
    q16a, q16b = (torch.empty(M, 3 * H, dtype=BF16, device=DEV) for _ in range(2))
    o32 = torch.empty(M, H, device=DEV)
    wqt = mk(H, 3 * H)
    out = []
    for name, flop, fn in (
            ("QKV  3-pass 14336x2304x768 split", 2.0 * M * 3 * H * H, lambda: ops.gemm_nt(xh, xl, wq, wql, M, 3 * H, H, 3, EPI_SPLIT, bias=b3, out_hi=q16a, out_lo=q16b)),
            ("FFN1 3-pass 14336x3072x768 gelu", 2.0 * M * I * H, lambda: ops.gemm_nt(xh, xl, w1, w1l, M, I, H, 3, EPI_GELU_SPLIT, bias=bI, out_hi=o16a, out_lo=o16b, aux16=o16c)),
            ("FFN2 3-pass 14336x768x3072 f32", 2.0 * M * I * H, lambda: ops.gemm_nt(hh, hl, w2, w2l, M, H, I, 3, EPI_F32, bias=bH, out32=o32)),
            ("out  3-pass 14336x768x768 f32", 2.0 * M * H * H, lambda: ops.gemm_nt(xh, xl, wq[:H], wql[:H], M, H, H, 3, EPI_F32, bias=bH, out32=o32)),
            ("dU   1-pass 14336x3072x768 dgelu", 2.0 * M * I * H, lambda: ops.gemm_nt(xh, None, w1, None, M, I, H, 1, EPI_DGELU_BF16, out_hi=o16a, aux16=o16c)),
            ("dX1  1-pass 14336x768x3072 f32", 2.0 * M * I * H, lambda: ops.gemm_nt(hh, None, w2, None, M, H, I, 1, EPI_F32, resid=o32, out32=o32, tile=2)),
            ("dX   1-pass 14336x768x2304 f32", 2.0 * M * 3 * H * H, lambda: ops.gemm_nt(q16a, None, wqt, None, M, H, 3 * H, 1, EPI_F32, out32=o32, tile=2))):
        t = med(fn)
        out.append("%s %6.1f us %4.0f TF" % (name, t, flop / t / 1e6))
    print(" | ".join(out), flush=True)


if __name__ == "__main__":
    main()

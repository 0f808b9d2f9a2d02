#!/usr/bin/env python3
"""Micro-benchmark of the LayerNorm kernels at the c2 stack shape (M = 14336 rows, H = 768; dropout 0.1 like the training
step): forward (fp32 in, fp32 + (hi, lo) out) and backward (dy, z in; dz fp32 + bf16 out, column partials), cold caches
(a 512 MB fill between launches) and warm (back to back).  Bytes are the algorithmic ones.  Usage: python3 tools/ln_bench.py"""
import sys

import torch

sys.path.insert(0, ".")
from clg_vqa_amd import ops  # noqa: E402
from clg_vqa_amd.ops import BF16  # noqa: E402

DEV = "cuda"


def timeit(fn, cold, n=12):
    filler = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
    ts = []
    for _ in range(n):
        if cold:
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
    for (M, H) in ((14336, 768), (30720, 768)):
        g = torch.Generator().manual_seed(0)
        y = torch.randn(M, H, generator=g).to(DEV)
        resid = torch.randn(M, H, generator=g).to(DEV)
        gamma, beta = torch.ones(H, device=DEV), torch.zeros(H, device=DEV)
        out32 = torch.empty(M, H, device=DEV)
        hi, lo = torch.empty(M, H, dtype=BF16, device=DEV), torch.empty(M, H, dtype=BF16, device=DEV)
        mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
        z = y.clone()
        ops.ln_fwd(z, resid, None, gamma, beta, 1e-5, out32, hi, lo, mean, rstd, M, H, p_pre=0.1, seed=5)
        dy = torch.randn(M, H, generator=g).to(DEV)
        dz = torch.empty(M, H, device=DEV)
        d16 = torch.empty(M, H, dtype=BF16, device=DEV)
        ws = ops.ln_bwd_ws(M, H, DEV)
        fwd_bytes = M * H * (4 + 4 + 4 + 4 + 2 + 2)   # y, resid in; z (in place), out32, hi, lo out
        bwd_bytes = M * H * (4 + 4 + 4 + 2)           # dy, z in; dz32, dpre16 out
        for cold in (True, False):
            tf = timeit(lambda: ops.ln_fwd(z, resid, None, gamma, beta, 1e-5, out32, hi, lo, mean, rstd, M, H, p_pre=0.1, seed=5), cold)
            tb = timeit(lambda: ops.ln_bwd(dy, z, mean, rstd, gamma, dz, d16, None, None, None, None, ws, M, H, p_pre=0.1, seed=5), cold)
            # the form the stack runs: residual recomputed from the previous LayerNorm's (z, mean, rstd), no fp32 output
            z2, m2, r2 = y.clone(), torch.empty_like(mean), torch.empty_like(rstd)
            tr = timeit(lambda: ops.ln_fwd(z2, None, None, gamma, beta, 1e-5, None, hi, lo, m2, r2, M, H, p_pre=0.1, seed=5,
                                           resid_ln=(z, mean, rstd, gamma, beta, None)), cold)
            rr_bytes = M * H * (4 + 4 + 4 + 2 + 2)
            print("M %5d H %d %s: ln_fwd, recomputed residual / no fp32 output %6.1f us = %.2f TB/s" % (
                M, H, "cold" if cold else "warm", tr, rr_bytes / tr / 1e6), flush=True)
            print("M %5d H %d %s: ln_fwd %6.1f us = %.2f TB/s | ln_bwd %6.1f us = %.2f TB/s" % (
                M, H, "cold" if cold else "warm", tf, fwd_bytes / tf / 1e6, tb, bwd_bytes / tb / 1e6), flush=True)


if __name__ == "__main__":
    main()

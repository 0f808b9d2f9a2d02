#!/usr/bin/env python3
"""Fixed cost per tile of the ping-pong GEMM: the QKV-shaped product (M 14336, N 2304: 504 tiles of 256 x 256 = two rounds on 256
CUs) at K = 256 ... 3072, cold caches, HIP events.  The time is linear in K; the intercept is what a tile costs before its first
and after its last K-tile (launch, DMA prologue, pipeline fill, epilogue), the slope the K loop.  Usage: python3 tools/gemm_kscan.py"""
import sys

import torch

sys.path.insert(0, ".")
from clg_vqa_amd import ops  # noqa: E402
from clg_vqa_amd.ops import BF16, EPI_BF16, EPI_SPLIT  # noqa: E402
from tools.gemm_exp import med  # noqa: E402

DEV = "cuda"
M, N = 14336, 2304


def main():
    g = torch.Generator().manual_seed(0)
    mk = lambda r, c: (torch.randn(r, c, generator=g) * 0.1).to(DEV).to(BF16)  # noqa: E731
    bias = torch.zeros(N, device=DEV)
    oh, ol = torch.empty(M, N, dtype=BF16, device=DEV), torch.empty(M, N, dtype=BF16, device=DEV)
    for passes, epi, name in ((3, EPI_SPLIT, "3-pass, split epilogue"), (1, EPI_BF16, "1-pass, bf16 epilogue")):
        pts = []
        for K in (256, 512, 768, 1536, 2304, 3072):
            xh, xl, wh, wl = mk(M, K), mk(M, K), mk(N, K), mk(N, K)
            for persist in (0, 256):
                t = med(lambda: ops.gemm_nt(xh, xl if passes == 3 else None, wh, wl if passes == 3 else None, M, N, K, passes, epi,
                                            bias=bias, out_hi=oh, out_lo=ol if passes == 3 else None, persist=persist))
                pts.append((K, persist, t))
        for persist in (0, 256):
            xs = [(k, t) for k, p, t in pts if p == persist]
            (k0, t0), (k1, t1) = xs[2], xs[-1]  # slope from K = 768 -> 3072
            slope = (t1 - t0) / (k1 - k0)
            print("%s, %s: " % (name, "persistent" if persist else "one tile per workgroup") +
                  "  ".join("K=%d %.1f" % (k, t) for k, t in xs) +
                  "  | us per 64 of K (two rounds) %.2f, intercept %.1f us = %.1f us per round" % (slope * 64, t0 - slope * k0, (t0 - slope * k0) / 2),
                  flush=True)


if __name__ == "__main__":
    main()

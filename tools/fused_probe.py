#!/usr/bin/env python3
"""what would a single-kernel "QKV projection + attention core" cost against the two-launch form?

A fused kernel has to own, per workgroup, ALL of Q|K|V of the (sample, head) pairs whose attention it runs: with one head
(3 x 64 = 192 projection columns) and 4 samples (4 x 56 = 224 rows) per workgroup that is a 224 x 192 output tile, 768 of
them at c2 -- instead of the 504 tiles of 256 x 256 the projection runs on today -- and the attention core runs in the
epilogue of that tile, on a CU whose matrix pipe then sits idle (one 8-wave workgroup per CU: nothing else is resident).
This probe measures the pieces on the real kernels (HIP events, cold caches: a 512 MB fill between launches):
  a. the projection alone with the tile the fusion forces (256 x 192, the closest existing configuration: same width, 14 %
     taller) against today's automatic choice;
  b. the attention core alone, cold (operands from HBM, what the two-launch form pays) and warm (operands cache-resident:
     an upper bound for what a fused epilogue could save -- its arithmetic and LDS traffic remain);
  c. the two-launch op as the stack runs it (projection, then attention right behind it: Q|K|V still in the Infinity Cache).
The fused form costs at least a.(forced tile) + [b.warm - its memory time]; it saves at most b.cold - b.warm.
Usage (GPU box): python3 tools/fused_probe.py"""
import sys

import torch

sys.path.insert(0, ".")
from clg_vqa_amd import ops  # noqa: E402
from clg_vqa_amd.ops import BF16, EPI_SPLIT  # noqa: E402

DEV = "cuda"
B, S, NH, H = 256, 56, 12, 768
M = B * S


def med(fn, cold=True, n=12):
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
    g = torch.Generator().manual_seed(0)
    mk = lambda r, c, s=1.0: (torch.randn(r, c, generator=g) * s).to(DEV)  # noqa: E731
    x, w = mk(M, H), mk(3 * H, H, 0.03)
    xh, xl = x.to(BF16), (x - x.to(BF16).float()).to(BF16)
    wh, wl = w.to(BF16), (w - w.to(BF16).float()).to(BF16)
    bias = torch.zeros(3 * H, device=DEV)
    qh, ql = torch.empty(M, 3 * H, dtype=BF16, device=DEV), torch.empty(M, 3 * H, dtype=BF16, device=DEV)
    ch, cl = torch.empty(M, H, dtype=BF16, device=DEV), torch.empty(M, H, dtype=BF16, device=DEV)
    lse = torch.empty(B * NH * S, device=DEV)
    am = torch.zeros(M, device=DEV)

    def proj(tile, persist=0):
        ops.gemm_nt(xh, xl, wh, wl, M, 3 * H, H, 3, EPI_SPLIT, bias=bias, out_hi=qh, out_lo=ql, tile=tile, persist=persist)

    def attn():
        ops.attn2_fwd(qh, ql, am, ch, cl, lse, B, S, NH, 64, 0.1, 7)

    flop_p, flop_a = 2.0 * M * 3 * H * H, 4.0 * B * S * S * H
    res = {}
    for name, tile, per in (("automatic (256x256, 504 tiles)", 0, 0), ("automatic, persistent", 0, 256),
                            ("256x192 (672 tiles; the fusion's tile is 224x192, 768 tiles)", 3, 0), ("256x192, persistent", 3, 256)):
        t = med(lambda: proj(tile, per))
        res[name] = t
        print("projection, tile %-62s %6.1f us = %4.0f TFLOP/s" % (name, t, flop_p / t / 1e6), flush=True)
    proj(0)
    tc, tw = med(attn, cold=True), med(attn, cold=False)
    print("attention core alone: cold %.1f us | warm %.1f us  (171 MB algorithmic: %.2f / %.2f TB/s)" % (tc, tw, 171e6 / tc / 1e6, 171e6 / tw / 1e6))
    t2 = med(lambda: (proj(0, 256), attn()))
    print("two launches back to back (the op as the stack runs it): %.1f us = %.0f TFLOP/s = %.1f %% of the bf16 peak" % (
        t2, (flop_p + flop_a) / t2 / 1e6, (flop_p + flop_a) / t2 / 1e6 / 25.0))
    t_auto = res["automatic, persistent"]
    t_forced = min(res["256x192 (672 tiles; the fusion's tile is 224x192, 768 tiles)"], res["256x192, persistent"])
    print("fused form, bounds: projection on the forced tile %.1f us (+%.1f vs automatic) + attention arithmetic in its epilogue; "
          "it can save at most the attention kernel's memory time: %.1f us (cold) ... %.1f us (in situ: two launches %.1f - "
          "projection %.1f - warm core %.1f)" % (t_forced, t_forced - t_auto, tc - tw, max(0.0, t2 - t_auto - tw), t2, t_auto, tw))


if __name__ == "__main__":
    main()

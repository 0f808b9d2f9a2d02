#!/usr/bin/env python3
"""Attention core (vl_attn2_fwd / vl_attn2_bwd) on the GPU box, HIP-event timed with cold caches, against its HBM floor."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clg_vqa_amd import ops  # noqa: E402
from clg_vqa_amd.ops import BF16  # noqa: E402


def timeit(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    for B, S in ((256, 56), (256, 76), (256, 120), (128, 120), (256, 140), (128, 156)):
        nh, H = 12, 768
        M = B * S
        qkv = torch.randn(M, 3 * H, device="cuda")
        hi, lo = torch.empty(M, 3 * H, dtype=BF16, device="cuda"), torch.empty(M, 3 * H, dtype=BF16, device="cuda")
        ops.split_f32(qkv, hi, lo)
        am = torch.zeros(M, device="cuda")
        ch, cl = torch.empty(M, H, dtype=BF16, device="cuda"), torch.empty(M, H, dtype=BF16, device="cuda")
        lse = torch.empty(B * nh * S, device="cuda")
        d16 = torch.randn(M, H, device="cuda").to(BF16)
        dq = torch.empty(M, 3 * H, dtype=BF16, device="cuda")
        junk = torch.empty(64 << 20, device="cuda")  # evict L2 / MALL between calls (256 MB)

        def cold(fn):
            def g():
                junk.zero_()
                fn()
            return g
        t_z = timeit(lambda: junk.zero_())
        fwd_mb = (2 * M * 3 * H * 2 + 2 * M * H * 2) / 1e6     # q,k,v (hi, lo) in, ctx (hi, lo) out
        bwd_mb = (M * 3 * H * 2 + M * H * 2 + M * 3 * H * 2) / 1e6  # q,k,v hi + dctx in, dqkv out
        for p in (0.0, 0.1):
            tf = timeit(cold(lambda: ops.attn2_fwd(hi, lo, am, ch, cl, lse, B, S, nh, 64, p, 1))) - t_z
            tb = timeit(cold(lambda: ops.attn2_bwd(hi, am, d16, lse, dq, B, S, nh, 64, p, 1))) - t_z
            print("B=%d S=%d p=%.1f: fwd %.1f us (%.2f TB/s of %.0f MB)  bwd %.1f us (%.2f TB/s of %.0f MB)" % (
                B, S, p, tf, fwd_mb / tf, fwd_mb, tb, bwd_mb / tb, bwd_mb), flush=True)


if __name__ == "__main__":
    main()

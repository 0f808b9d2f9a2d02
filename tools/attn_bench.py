#!/usr/bin/env python3
"""Attention core A/B on the GPU box: fp32-pipe kernels (vl_attn_*) vs bf16-pipe kernels (vl_attn2_*), HIP-event timed."""
import sys
import os

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
    for B, S in ((256, 56), (256, 120), (128, 120), (256, 140)):
        nh, H = 12, 768
        M = B * S
        qkv = torch.randn(M, 3 * H, device="cuda")
        hi, lo = torch.empty(M, 3 * H, dtype=BF16, device="cuda"), torch.empty(M, 3 * H, dtype=BF16, device="cuda")
        ops.split_f32(qkv, hi, lo)
        am = torch.zeros(M, device="cuda")
        ch, cl = torch.empty(M, H, dtype=BF16, device="cuda"), torch.empty(M, H, dtype=BF16, device="cuda")
        lse = torch.empty(B * nh * S, device="cuda")
        d32 = torch.randn(M, H, device="cuda")
        d16 = d32.to(BF16)
        dq = torch.empty(M, 3 * H, dtype=BF16, device="cuda")
        junk = torch.empty(64 << 20, device="cuda")  # evict L2 / MALL between calls (256 MB)

        def cold(fn):
            def g():
                junk.zero_()
                fn()
            return g
        t_z = timeit(lambda: junk.zero_())
        r = {}
        for p in (0.0, 0.1):
            r["fwd32 p=%.1f" % p] = timeit(cold(lambda: ops.attn_fwd(qkv, am, ch, cl, lse, B, S, nh, 64, p, 1))) - t_z
            r["fwd16 p=%.1f" % p] = timeit(cold(lambda: ops.attn2_fwd(hi, lo, am, ch, cl, lse, B, S, nh, 64, p, 1))) - t_z
            r["bwd32 p=%.1f" % p] = timeit(cold(lambda: ops.attn_bwd(qkv, am, ch, cl, d32, lse, dq, B, S, nh, 64, p, 1))) - t_z
            r["bwd16 p=%.1f" % p] = timeit(cold(lambda: ops.attn2_bwd(hi, am, d16, lse, dq, B, S, nh, 64, p, 1))) - t_z
        print("B=%d S=%d: " % (B, S) + "  ".join("%s %.1f us" % kv for kv in r.items()), flush=True)


if __name__ == "__main__":
    main()

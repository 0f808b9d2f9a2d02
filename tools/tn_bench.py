#!/usr/bin/env python3
"""Stand-alone timing of one layer's weight-gradient products: grouped ping-pong launch vs the per-product split-K path."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clg_vqa_amd import ops  # noqa: E402
from clg_vqa_amd.engine import dw_gemm  # noqa: E402

BF16 = torch.bfloat16


def bench(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    K, H, I = 14336, 768, 3072
    dev = "cuda"
    pad = int(sys.argv[1]) if len(sys.argv) > 1 else 0   # extra elements in every leading dimension (channel-spread test)
    mk = lambda n: torch.randn(K, n + pad, device=dev).to(BF16)[:, :n]  # noqa: E731
    dqkv, dt1, du16, dt2 = [mk(n) for n in (3 * H, H, I, H)]
    x, ctx, x1, h = [mk(n) for n in (H, H, H, I)]
    print("leading-dimension pad: %d elements" % pad)
    outs = [torch.empty(3 * H, H, device=dev), torch.empty(H, H, device=dev), torch.empty(I, H, device=dev),
            torch.empty(H, I, device=dev)]
    probs = [(dqkv[:, i * H:(i + 1) * H], x, outs[0][i * H:(i + 1) * H], None) for i in range(3)]
    probs += [(dt1, ctx, outs[1], None), (du16, x1, outs[2], None), (dt2, h, outs[3], None)]
    flops = 2.0 * K * (3 * H * H + H * H + 2 * I * H)
    for splits in (1, 2):
        us = bench(lambda: ops.gemm_tn_grouped(probs, K, splits))
        print("grouped splits=%d: %.1f us  %.0f TF" % (splits, us, flops / us / 1e6))
    for i, p in enumerate(probs):
        f = 2.0 * K * p[0].shape[1] * p[1].shape[1]
        us = bench(lambda: ops.gemm_tn_grouped([p], K, 1))
        print("  problem %d alone (%d tiles): %.1f us %.0f TF" % (i, ((p[0].shape[1] + 255) // 256) * ((p[1].shape[1] + 255) // 256), us, f / us / 1e6))

    def old():
        o = [torch.empty(3 * H, H, device=dev), torch.empty(H, H, device=dev), torch.empty(I, H, device=dev), torch.empty(H, I, device=dev)]
        ops.gemm_tn_splitk(dqkv, x, 3 * H, H, K, o[0])
        ops.gemm_tn_splitk(dt1, ctx, H, H, K, o[1])
        ops.gemm_tn_splitk(du16, x1, I, H, K, o[2])
        ops.gemm_tn_splitk(dt2, h, H, I, K, o[3])
    us = bench(old)
    print("per-product split-K: %.1f us  %.0f TF" % (us, flops / us / 1e6))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Which part of the grouped weight-gradient launch is slow?  Variants of the problem list, K = 14336."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clg_vqa_amd import ops  # noqa: E402
from tools.scale_probe import bench  # noqa: E402

BF16 = torch.bfloat16


def main():
    K, H, I, dev = 14336, 768, 3072, "cuda"
    mk = lambda n: torch.randn(K, n, device=dev).to(BF16)  # noqa: E731
    dqkv, dt1, du16, dt2, x, ctx, x1, h = [mk(n) for n in (3 * H, H, I, H, H, H, H, I)]
    o = lambda m, n: torch.empty(m, n, device=dev)  # noqa: E731
    q3 = [(dqkv[:, i * H:(i + 1) * H], x, o(H, H), None) for i in range(3)]
    q3_indep = [(mk(H), mk(H), o(H, H), None) for i in range(3)]
    po, p1, p2 = (dt1, ctx, o(H, H), None), (du16, x1, o(I, H), None), (dt2, h, o(H, I), None)
    variants = {
        "w1+w2 (72 tiles)": [p1, p2],
        "q,k,v,o (36 tiles)": q3 + [po],
        "q,k,v,o,w1,w2 (108)": q3 + [po, p1, p2],
        "qkv-as-one,o,w1,w2 (108)": [(dqkv, x, o(3 * H, H), None), po, p1, p2],
        "independent q,k,v + o,w1,w2 (108)": q3_indep + [po, p1, p2],
        "w1,w2,q,k,v,o order (108)": [p1, p2] + q3 + [po],
    }
    for name, probs in variants.items():
        us = bench(lambda: ops.gemm_tn_grouped(probs, K, 1))
        fl = sum(2.0 * K * p[0].shape[1] * p[1].shape[1] for p in probs)
        print("%-36s %7.1f us  %5.0f TF" % (name, us, fl / us / 1e6), flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""How the per-K-tile time of the ping-pong kernels scales with the number of resident workgroups (NT vs TN operand
layout), K = 14336, one 256 x 256 output tile per workgroup."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clg_vqa_amd import _lib, ops  # noqa: E402
from clg_vqa_amd.ops import EPI_F32  # noqa: E402

BF16 = torch.bfloat16


def bench(fn, iters=6):
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
    K, dev = 14336, "cuda"
    _lib.lib().vl_debug_set(7, 2)  # 256-wide ping-pong tiles
    for tm, tn in [(1, 1), (3, 3), (6, 6), (9, 8), (12, 9), (12, 12), (16, 12), (16, 16)]:
        M, N = 256 * tm, 256 * tn
        a_t, b_t = torch.randn(K, M, device=dev).to(BF16), torch.randn(K, N, device=dev).to(BF16)
        out = torch.empty(M, N, device=dev)
        us_tn = bench(lambda: ops.gemm_tn_grouped([(a_t, b_t, out, None)], K, 1))
        a_n, b_n = a_t.t().contiguous(), b_t.t().contiguous()
        us_nt = bench(lambda: ops.gemm_nt(a_n, None, b_n, None, M, N, K, 1, EPI_F32, out32=out))
        print("%3d workgroups: TN %7.1f us (%.2f us / K-tile, %4.0f TF)   NT %7.1f us (%.2f us / K-tile, %4.0f TF)" % (
            tm * tn, us_tn, us_tn / 224, 2.0 * M * N * K / us_tn / 1e6, us_nt, us_nt / 224, 2.0 * M * N * K / us_nt / 1e6), flush=True)
    _lib.lib().vl_debug_set(7, 1)


if __name__ == "__main__":
    main()

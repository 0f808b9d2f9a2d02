#!/usr/bin/env python3
"""Embedding backward at c2 (B 256, T 20, H 768, vocab 250002; 9216 box rows): fixed-order forms against the float-atomic
kernels.  Usage: python3 tools/embed_bwd_bench.py"""
import sys

import torch

sys.path.insert(0, ".")
from clg_vqa_amd import ops  # noqa: E402
from clg_vqa_amd.synthetic import make_batch  # noqa: E402

DEV = "cuda"


def timeit(fn, n=10):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    B, T, H, V, L = 256, 20, 768, 36, 7
    batch = make_batch(B, seed=3)
    ids, seg = batch[3].to(DEV).contiguous(), batch[6].to(DEV).contiguous()
    dz = torch.randn(B * T, H, device=DEV)
    dword = torch.zeros(250002, H, device=DEV)
    dpos, dtyp = torch.zeros(514, H, device=DEV), torch.zeros(2, H, device=DEV)
    flags = torch.zeros(250002, dtype=torch.uint8, device=DEV)
    t_at = timeit(lambda: ops.embed_text_bwd(ids, seg, dz, dword, dpos, dtyp, B, T, H, 1, row_flags=flags))
    t_det = timeit(lambda: ops.embed_text_bwd_det(ids, seg, dz, dword, dpos, dtyp, B, T, H, 1, row_flags=flags))
    print("text rows (word + position + type tables): atomics %.1f us | fixed order (sort + run sums) %.1f us" % (t_at, t_det))
    R = B * V
    loc, dy = torch.randn(R, L, device=DEV), torch.randn(R, H, device=DEV)
    dw, db = torch.zeros(H, L, device=DEV), torch.zeros(H, device=DEV)
    l_at = timeit(lambda: ops.loc_linear_bwd(loc, dy, dw, db, R, L, H, deterministic=False))
    l_det = timeit(lambda: ops.loc_linear_bwd(loc, dy, dw, db, R, L, H, deterministic=True))
    print("box-location Linear backward: atomics %.1f us | fixed order %.1f us" % (l_at, l_det))
    x = torch.randn(281_600_000 // 8, device=DEV)
    out, ws = torch.zeros(1, device=DEV), torch.empty(2048, device=DEV)
    s_at = timeit(lambda: ops.sumsq(x, out))
    s_det = timeit(lambda: ops.sumsq(x, out, ws=ws))
    print("sum of squares (35 M floats): atomics %.1f us | fixed order %.1f us" % (s_at, s_det))


if __name__ == "__main__":
    main()

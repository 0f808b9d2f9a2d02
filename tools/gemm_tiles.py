#!/usr/bin/env python3
"""The stack's GEMM shapes at c2 on every kernel / tile choice of vl_gemm_nt_ex (VL_GX_TILE): 0 automatic, 2 / 3 / 5 ping-pong
256x256 / 256x192 / 224x256 (one 8-wave workgroup per CU), 6 single-barrier kernel, 128 / 192 / 256 its widths, 7 generic
128x128 kernel (4 waves, two workgroups per CU, register-staged).  HIP events, cold caches.  Usage: python3 tools/gemm_tiles.py"""
import sys

import torch

sys.path.insert(0, ".")
from clg_vqa_amd import ops  # noqa: E402
from clg_vqa_amd.ops import BF16, EPI_DGELU_BF16, EPI_F32, EPI_GELU_SPLIT, EPI_SPLIT  # noqa: E402

DEV = "cuda"
M, H, I = 14336, 768, 3072


def med(fn, n=8):
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
    mk = lambda r, c: (torch.randn(r, c, generator=g) * 0.1).to(DEV).to(BF16)  # noqa: E731
    xh, xl, hh, hl = mk(M, H), mk(M, H), mk(M, I), mk(M, I)
    wq, wql, w1, w1l, w2, w2l = mk(3 * H, H), mk(3 * H, H), mk(I, H), mk(I, H), mk(H, I), mk(H, I)
    b3, bI, bH = torch.zeros(3 * H, device=DEV), torch.zeros(I, device=DEV), torch.zeros(H, device=DEV)
    o16a, o16b, o16c = (torch.empty(M, I, dtype=BF16, device=DEV) for _ in range(3))
    q16a, q16b = (torch.empty(M, 3 * H, dtype=BF16, device=DEV) for _ in range(2))
    o32 = torch.empty(M, H, device=DEV)
    shapes = (
        ("QKV  3-pass x2304x768 split", lambda t: ops.gemm_nt(xh, xl, wq, wql, M, 3 * H, H, 3, EPI_SPLIT, bias=b3, out_hi=q16a, out_lo=q16b, tile=t)),
        ("FFN1 3-pass x3072x768 gelu", lambda t: ops.gemm_nt(xh, xl, w1, w1l, M, I, H, 3, EPI_GELU_SPLIT, bias=bI, out_hi=o16a, out_lo=o16b, aux16=o16c, tile=t)),
        ("FFN2 3-pass x768x3072 f32", lambda t: ops.gemm_nt(hh, hl, w2, w2l, M, H, I, 3, EPI_F32, bias=bH, out32=o32, tile=t)),
        ("out  3-pass x768x768 f32", lambda t: ops.gemm_nt(xh, xl, wq[:H], wql[:H], M, H, H, 3, EPI_F32, bias=bH, out32=o32, tile=t)),
        ("dU   1-pass x3072x768 dgelu", lambda t: ops.gemm_nt(xh, None, w1, None, M, I, H, 1, EPI_DGELU_BF16, out_hi=o16a, aux16=o16c, tile=t)),
        ("dX1  1-pass x768x3072 f32", lambda t: ops.gemm_nt(hh, None, w2, None, M, H, I, 1, EPI_F32, resid=o32, out32=o32, tile=t)))
    tiles = (0, 2, 3, 5, 6, 128, 192, 256, 7)
    print("%-30s" % "tile" + "".join("%8s" % t for t in tiles))
    for name, fn in shapes:
        row = []
        for t in tiles:
            try:
                row.append("%8.1f" % med(lambda: fn(t)))
            except RuntimeError:
                row.append("%8s" % "-")
        print("%-30s" % name + "".join(row), flush=True)


if __name__ == "__main__":
    main()

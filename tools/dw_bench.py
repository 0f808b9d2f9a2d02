#!/usr/bin/env python3
"""Micro-benchmark of the weight-gradient GEMM forms on one layer's six problems at c2 (rows = 14336): K-major (mode 0),
row-major 8-wave ping-pong (mode 3); `per`: one 9-tile problem per operand width.  Usage (GPU box): python3 tools/dw_bench.py"""
import os
import sys

import torch

sys.path.insert(0, ".")
from clg_vqa_amd import _lib, ops  # noqa: E402
from clg_vqa_amd.ops import BF16  # noqa: E402

DEV = "cuda"
K, H, I = 14336, 768, 3072


def main():
    g = torch.Generator().manual_seed(0)
    mk = lambda n: (torch.randn(K, n, generator=g) * 0.5).to(DEV).to(BF16)  # noqa: E731
    dqkv, dt1, du, dt2, x, ctx, x1, hh = (mk(n) for n in (3 * H, H, I, H, H, H, H, I))
    names = ("q", "k", "v", "o", "w1", "w2")
    shapes = dict(q=(H, H), k=(H, H), v=(H, H), o=(H, H), w1=(I, H), w2=(H, I))
    outs = {n: torch.empty(shapes[n], device=DEV) for n in names}
    cs = {n: torch.empty((shapes[n][1] + 255) // 256, shapes[n][0], device=DEV) for n in ("q", "k", "v", "w1")}
    rm = [(dqkv[:, :H], None, x, None, outs["q"], None, H, H, cs["q"]), (dqkv[:, H:2 * H], None, x, None, outs["k"], None, H, H, cs["k"]),
          (dqkv[:, 2 * H:], None, x, None, outs["v"], None, H, H, cs["v"]), (dt1, None, ctx, None, outs["o"], None, H, H, None),
          (du, None, x1, None, outs["w1"], None, I, H, cs["w1"]), (dt2, None, hh, None, outs["w2"], None, H, I, None)]
    if os.environ.get("NOCS"):
        rm = [t[:8] + (None,) for t in rm]
    L = _lib.lib()
    mats = [dqkv, dt1, du, dt2, x, ctx, x1, hh]
    imgs = [torch.empty(L.vl_blocked_elems(K, m.shape[1]), dtype=BF16, device=DEV) for m in mats]
    ops.transpose_blocked([(m, d, None) for m, d in zip(mats, imgs)], K)
    Tqkv, Tt1, Tu, Tt2, Tx, Tctx, Tx1, Th = imgs
    km = [(Tqkv, 3 * H, Tx, H, outs["q"], None, H, H, None), (Tqkv[64 * H:], 3 * H, Tx, H, outs["k"], None, H, H, None),
          (Tqkv[2 * 64 * H:], 3 * H, Tx, H, outs["v"], None, H, H, None), (Tt1, H, Tctx, H, outs["o"], None, H, H, None),
          (Tu, I, Tx1, H, outs["w1"], None, I, H, None), (Tt2, H, Th, I, outs["w2"], None, H, I, None)]
    filler = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
    for mode, probs in ((0, km), (3, rm)):
        ts = []
        for it in range(8):
            filler.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.dw_grouped_mixed(probs, K, mode)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        flop = 2.0 * K * (4 * H * H + 2 * H * I)
        print("mode %d: %.1f us  (%.0f TFLOP/s on %d tiles)" % (mode, ts[len(ts) // 2], flop / ts[len(ts) // 2] / 1e6, 108), flush=True)
    # stream-K form (row-major operands) on a fixed workgroup budget
    rm9 = [(a, None, b, None, o, m, M_, N_, c) for (a, _, b, _, o, m, M_, N_, c) in rm]
    for budget in [int(b) for b in os.environ.get("BUDGETS", "64,72,80,88,96,108,128,160,216,256").split(",")]:
        ws = ops.dw_streamk_ws(budget, DEV)
        ts = []
        for it in range(8):
            filler.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.dw_grouped_mixed(rm9, K, 3, budget=budget, ws=ws)
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts.sort()
        t = ts[len(ts) // 2]
        print("stream-K, %3d workgroups: %.1f us  (%.0f TFLOP/s = %.2f TFLOP/s per CU; %.2f us per K-tile and CU)" % (
            budget, t, flop / t / 1e6, flop / t / 1e6 / budget, t / (108 * 224 / budget)), flush=True)




def per_problem():
    """each problem alone (mode 3 and 7 against 0): which operand shapes are slow"""
    g = torch.Generator().manual_seed(0)
    mk = lambda n: (torch.randn(K, n, generator=g) * 0.5).to(DEV).to(BF16)  # noqa: E731
    L = _lib.lib()
    filler = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
    for (na, nb) in ((H, H), (3 * H, H), (I, H), (H, I), (I, I)):
        a, b = mk(na), mk(nb)
        Mo, No = min(na, 768), min(nb, 768)  # a 768 x 768 output block (9 tiles) out of operands of different widths
        out = torch.empty(Mo, No, device=DEV)
        ta, tb = (torch.empty(L.vl_blocked_elems(K, t.shape[1]), dtype=BF16, device=DEV) for t in (a, b))
        ops.transpose_blocked([(a, ta, None), (b, tb, None)], K)
        res = []
        for mode, pr in ((0, [(ta, na, tb, nb, out, None, Mo, No, None)]), (1, [(a, None, tb, nb, out, None, Mo, No, None)]),
                         (2, [(ta, na, b, None, out, None, Mo, No, None)]), (3, [(a, None, b, None, out, None, Mo, No, None)])):
            ts = []
            for it in range(6):
                filler.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ops.dw_grouped_mixed(pr, K, mode)
                e1.record()
                e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            ts.sort()
            res.append(ts[len(ts) // 2])
        print("dY width %4d, X width %4d (9 tiles): modes 0 / 1 / 2 / 3: %.0f / %.0f / %.0f / %.0f us" % (na, nb, *res), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "per":
        per_problem()
    else:
        main()

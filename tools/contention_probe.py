#!/usr/bin/env python3
"""What does a weight-gradient GEMM on a second stream cost the dX products of the main stream -- CU slots or memory
system?  The four dX products of one c2 layer (dU 14336 x 3072 x 768 with the GELU' epilogue, dX1 14336 x 768 x 3072,
dctx 14336 x 768 x 768, dX 14336 x 768 x 2304) are timed with HIP events
  (a) alone,
  (b) beside the stream-K weight-gradient GEMM of one layer on G workgroups (G CUs are held by it, 256 - G stay free:
      every dX product with <= 256 - G tiles still fits one round),
  (c) beside the one-workgroup-per-tile form (108 CUs).
Usage (GPU box): python3 tools/contention_probe.py"""
import sys

import torch

sys.path.insert(0, ".")
from clg_vqa_amd import ops  # noqa: E402
from clg_vqa_amd.ops import BF16, EPI_BF16, EPI_DGELU_BF16, EPI_F32  # noqa: E402

DEV = "cuda"
M, H, I = 14336, 768, 3072


def main():
    g = torch.Generator().manual_seed(0)
    mk = lambda r, c: (torch.randn(r, c, generator=g) * 0.5).to(DEV).to(BF16)  # noqa: E731
    dt2, du, dt1, dqkv = mk(M, H), mk(M, I), mk(M, H), mk(M, 3 * H)
    x, ctx, x1, hh = mk(M, H), mk(M, H), mk(M, H), mk(M, I)
    w2t, w1t, wot, wqkvt = mk(I, H), mk(H, I), mk(H, H), mk(H, 3 * H)
    u16 = mk(M, I)
    du_out = torch.empty(M, I, dtype=BF16, device=DEV)
    f32 = lambda: torch.empty(M, H, device=DEV)  # noqa: E731
    dx1, dxo, resid = f32(), f32(), f32()
    dctx = torch.empty(M, H, dtype=BF16, device=DEV)
    outs = {n: torch.empty(s, device=DEV) for n, s in dict(q=(H, H), k=(H, H), v=(H, H), o=(H, H), w1=(I, H), w2=(H, I)).items()}
    probs = [(dqkv[:, :H], None, x, None, outs["q"], None, H, H, None), (dqkv[:, H:2 * H], None, x, None, outs["k"], None, H, H, None),
             (dqkv[:, 2 * H:], None, x, None, outs["v"], None, H, H, None), (dt1, None, ctx, None, outs["o"], None, H, H, None),
             (du, None, x1, None, outs["w1"], None, I, H, None), (dt2, None, hh, None, outs["w2"], None, H, I, None)]
    side = torch.cuda.Stream()

    def dx_products(tile_wide):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        ev[0].record()
        ops.gemm_nt(dt2, None, w2t, None, M, I, H, 1, EPI_DGELU_BF16, out_hi=du_out, aux16=u16, tile=tile_wide)
        ev[1].record()
        ops.gemm_nt(du, None, w1t, None, M, H, I, 1, EPI_F32, resid=resid, out32=dx1, tile=2)
        ev[2].record()
        ops.gemm_nt(dt1, None, wot, None, M, H, H, 1, EPI_BF16, out_hi=dctx, tile=2)
        ev[3].record()
        ops.gemm_nt(dqkv, None, wqkvt, None, M, H, 3 * H, 1, EPI_F32, resid=resid, out32=dxo, tile=2)
        ev[4].record()
        return ev

    def run(label, budget, tile_wide):
        ws = ops.dw_streamk_ws(budget, DEV) if budget and budget > 0 else None
        res = []
        for it in range(6):
            torch.cuda.synchronize()
            if budget is not None:
                side.wait_stream(torch.cuda.current_stream())
                ops.set_stream(side.cuda_stream)
                for _ in range(2):  # two layers' worth: the dW stream is busy for the whole measurement
                    ops.dw_grouped_mixed(probs, M, 3, budget=budget, ws=ws)
                ops.set_stream(None)
                torch.cuda._sleep(200000)  # let the weight-gradient workgroups take their CUs first
            ev = dx_products(tile_wide)
            torch.cuda.synchronize()
            res.append([ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(4)])
        res.sort(key=sum)
        r = res[len(res) // 2]
        print("%-46s dU %6.1f  dX1 %6.1f  dctx %5.1f  dX %6.1f   sum %6.1f us" % (label, r[0], r[1], r[2], r[3], sum(r)), flush=True)

    run("alone, dU tile automatic (224 x 256, 768 tiles)", None, 0)
    run("alone, dU tile 256 x 256 (672 tiles)", None, 2)
    for b in (44, 64, 88):
        run("beside stream-K dW on %d workgroups, dU 256x256" % b, b, 2)
    run("beside stream-K dW on 88 workgroups, dU automatic", 88, 0)
    run("beside tile-per-workgroup dW (108), dU automatic", 0, 0)
    run("beside tile-per-workgroup dW (108), dU 256x256", 0, 2)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Micro-benchmark of the small-M GEMM path (tile 8 of vl_gemm_nt_ex) against the automatic big-tile choice, to place
the switch-over point.  Usage (GPU box): python3 tools/small_gemm_bench.py > gpurun_out/small_gemm.txt"""
import sys

import torch

sys.path.insert(0, ".")
from clg_vqa_amd import _lib, ops  # noqa: E402
from clg_vqa_amd.ops import BF16, EPI_F32  # noqa: E402

DEV = "cuda"


def run(M, N, K, passes, tile, ws):
    L = _lib.lib()
    x = torch.randn(M, K, device=DEV)
    w = torch.randn(N, K, device=DEV) * 0.05
    xh, xl = torch.empty(M, K, dtype=BF16, device=DEV), torch.empty(M, K, dtype=BF16, device=DEV)
    wh, wl = torch.empty(N, K, dtype=BF16, device=DEV), torch.empty(N, K, dtype=BF16, device=DEV)
    ops.split_f32(x, xh, xl)
    ops.split_f32(w, wh, wl)
    out = torch.empty(M, N, device=DEV)
    filler = torch.empty(256 << 20, dtype=torch.uint8, device=DEV)
    st = torch.cuda.current_stream().cuda_stream

    import ctypes
    extra = (ctypes.c_int64 * 8)()
    extra[0] = tile
    if ws is not None:
        extra[1], extra[2] = ws.data_ptr(), ws.numel()

    def call():
        _lib.check(L.vl_gemm_nt_ex(xh.data_ptr(), xl.data_ptr() if passes == 3 else None, K, wh.data_ptr(),
                                   wl.data_ptr() if passes == 3 else None, K, M, N, K, passes, EPI_F32, None, None,
                                   out.data_ptr(), N, None, None, None, 0, ctypes.cast(extra, ctypes.c_void_p), st), "gemm")
    for _ in range(3):
        call()
    ts = []
    for _ in range(10):
        filler.zero_()  # cold operands, as in the step (the weights were last touched a layer ago)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def main():
    ws = torch.empty(64 << 20, dtype=torch.float32, device=DEV)
    print("%6s %6s %6s %2s | %9s %9s %9s" % ("M", "N", "K", "p", "auto us", "small us", "generic"))
    for passes in (3, 1):
        for M in (128, 256, 512, 1024, 2048, 4096):
            for (N, K) in ((768, 768), (3072, 768), (768, 3072), (1842, 1536), (2304, 768)):
                t_auto = run(M, N, K, passes, 0, None)
                t_small = run(M, N, K, passes, 8, ws)
                t_gen = run(M, N, K, passes, 7, None)
                print("%6d %6d %6d %2d | %9.1f %9.1f %9.1f" % (M, N, K, passes, t_auto, t_small, t_gen), flush=True)


if __name__ == "__main__":
    main()

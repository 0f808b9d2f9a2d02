#!/usr/bin/env python3
"""Micro-benchmark of vl_gemm_nt on the step's GEMM shapes (run on the GPU box).  Interleaved A/B rounds in one
process (cdna guide rule 24); random operands (rule 25)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clg_vqa_amd import _lib, ops  # noqa: E402
from clg_vqa_amd.ops import BF16, EPI_F32  # noqa: E402

DEV = "cuda"


def bench(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3  # us


def main():
    L = _lib.lib()
    shapes = [("qkv   fwd", 14336, 2304, 768), ("oproj fwd", 14336, 768, 768), ("ffn1  fwd", 14336, 3072, 768),
              ("ffn2  fwd", 14336, 768, 3072), ("qkv   dX ", 14336, 768, 2304)]
    variants = [("old-256", dict(g=0, bn=256, pp=0)), ("old-192", dict(g=0, bn=192, pp=0)),
                ("pp-256", dict(g=0, bn=0, pp=2)), ("pp-192", dict(g=0, bn=0, pp=3))]
    for passes in (1, 3):
        print("== passes %d ==" % passes)
        for name, M, N, K in shapes:
            # rotate through enough distinct A operands to defeat the 256 MiB Infinity Cache (HBM-cold, like in the
            # training step where A was produced by an earlier kernel)
            nbuf = max(2, int(600e6 // (M * K * 2 * (2 if passes == 3 else 1))) + 1)
            a_list = [torch.randn(M, K, device=DEV).to(BF16) for _ in range(nbuf)]
            al_list = [torch.randn(M, K, device=DEV).to(BF16) for _ in range(nbuf if passes == 3 else 1)]
            it = [0]
            a = a_list[0]; al = al_list[0]
            b = torch.randn(N, K, device=DEV).to(BF16)
            bl = torch.randn(N, K, device=DEV).to(BF16)
            out = torch.empty(M, N, device=DEV)
            row = []
            for vname, v in variants:
                L.vl_debug_set(2, v["g"])
                L.vl_debug_set(1, v["bn"])
                L.vl_debug_set(7, v.get("pp", 0))
                def run():
                    it[0] += 1
                    ops.gemm_nt(a_list[it[0] % nbuf], al_list[it[0] % len(al_list)], b, bl, M, N, K, passes, EPI_F32,
                                out32=out)
                us = bench(run)
                row.append("%s %7.1f us %6.0f TF" % (vname, us, 2.0 * M * N * K / us / 1e6))
            if passes == 1:  # library (hipBLASLt via torch) on the same cold operands, bf16 out: the speed to beat
                outb = torch.empty(M, N, device=DEV, dtype=BF16)
                def run_lib():
                    it[0] += 1
                    torch.matmul(a_list[it[0] % nbuf], b.t(), out=outb)
                us = bench(run_lib)
                row.append("torch.matmul %7.1f us %6.0f TF" % (us, 2.0 * M * N * K / us / 1e6))
            print("%s M=%5d N=%4d K=%4d | %s" % (name, M, N, K, " | ".join(row)), flush=True)
    L.vl_debug_set(1, 0)
    L.vl_debug_set(2, 0)
    L.vl_debug_set(7, 1)


if __name__ == "__main__":
    main()

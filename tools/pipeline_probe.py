#!/usr/bin/env python3
"""Is 2-way micro-batch pipelining of the FORWARD worth building?  Two half-batch forwards on two HIP streams (the
HBM-bound LayerNorm / attention kernels of one half under the MFMA-bound GEMMs of the other) against one full-batch
forward on one stream.  Forward only, no_grad, same weights."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clg_vqa_amd.config import GQA_TASK_CFG, BertConfig, uc2_base_config  # noqa: E402
from clg_vqa_amd.encoders import BertForVLTasks  # noqa: E402
from clg_vqa_amd.synthetic import make_batch  # noqa: E402


def main():
    dev = torch.device("cuda")
    config = BertConfig.from_dict(uc2_base_config(vocab=30000))
    torch.manual_seed(0)
    model = BertForVLTasks(config, GQA_TASK_CFG, ["TASK15"]).to(dev).train()
    full = tuple(t.to(dev) for t in make_batch(256, vocab_size=30000, seed=1))
    halves = [tuple(t[i * 128:(i + 1) * 128].contiguous() if t.dim() > 0 and t.shape[0] == 256 else t for t in full) for i in range(2)]

    def fwd(b):
        with torch.no_grad():
            return model(b[3], b[0], b[1], "TASK15", b[6], b[5], b[2])[0]

    def timed(fn, n=20):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    t_full = timed(lambda: fwd(full))
    t_half_serial = timed(lambda: (fwd(halves[0]), fwd(halves[1])))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def two_streams():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            fwd(halves[0])
        with torch.cuda.stream(s2):
            fwd(halves[1])
        cur.wait_stream(s1); cur.wait_stream(s2)
    t_two = timed(two_streams)
    print("forward of 256 samples: one stream %.2f ms | two halves back to back %.2f ms | two halves on two streams %.2f ms"
          % (t_full, t_half_serial, t_two))


if __name__ == "__main__":
    main()

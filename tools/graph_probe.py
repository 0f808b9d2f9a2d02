#!/usr/bin/env python3
"""Feasibility probe: capture forward+backward of the training step in a HIP graph (torch.cuda.CUDAGraph) and time
replays against the eager step (optimizer stays eager)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import TASK_CFG  # noqa: E402
from bench import uc2_full_cfg  # noqa: E402
from clg_vqa_amd import task_utils  # noqa: E402
from clg_vqa_amd.config import BertConfig  # noqa: E402
from clg_vqa_amd.encoders import BertForVLTasks  # noqa: E402
from clg_vqa_amd.optim import FusedAdamW  # noqa: E402
from clg_vqa_amd.synthetic import make_batch  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    config = BertConfig.from_dict(uc2_full_cfg())
    torch.manual_seed(1)
    model = BertForVLTasks(config, TASK_CFG, ["TASK15"]).to(dev)
    model.train()
    opt = FusedAdamW(model, base_lr=4e-5, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-6, correct_bias=True,
                     max_grad_norm=1.0, warmup_steps=100, t_total=100000)
    batch = tuple(t.to(dev) for t in make_batch(B, seed=3))
    crit = torch.nn.CrossEntropyLoss()

    def fwd_bwd():
        loss, score = task_utils.ForwardModelsTrain(config, TASK_CFG, dev, "TASK15", batch, model, crit)
        loss.backward()
        return loss

    def timeit(fn, n=20):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def eager():
        fwd_bwd()
        opt.step()
    for _ in range(3):
        eager()
    print("eager step: %.2f ms" % timeit(eager), flush=True)

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2):
            fwd_bwd()
            opt.step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        static_loss = fwd_bwd()
    torch.cuda.synchronize()
    print("captured", flush=True)

    def graphed():
        g.replay()
        opt.step()
    for _ in range(3):
        graphed()
    print("graphed fwd+bwd + eager optimizer: %.2f ms   loss %.4f" % (timeit(graphed), float(static_loss)), flush=True)
    print("graph replay only: %.2f ms" % timeit(g.replay), flush=True)


if __name__ == "__main__":
    main()

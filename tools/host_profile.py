#!/usr/bin/env python3
"""Where does the HOST spend its enqueue time?  Wraps every function of clg_vqa_amd.ops with a wall-clock accumulator and runs
a few c2 training steps (the GPU is far behind the host, so a call that blocks shows up here).  Usage: python3 tools/host_profile.py"""
import collections
import sys
import time

import torch

sys.path.insert(0, ".")
from clg_vqa_amd import ops, task_utils  # noqa: E402
from clg_vqa_amd.config import GQA_TASK_CFG as TASK_CFG, BertConfig, uc2_base_config  # noqa: E402
from clg_vqa_amd.encoders import BertForVLTasks  # noqa: E402
from clg_vqa_amd.optim import FusedAdamW  # noqa: E402
from clg_vqa_amd.synthetic import make_batch  # noqa: E402

acc = collections.defaultdict(lambda: [0, 0.0])


def wrap(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            e = acc[name]
            e[0] += 1
            e[1] += time.perf_counter() - t0
    return w


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(1234)
    config = BertConfig.from_dict(uc2_base_config())
    model = BertForVLTasks(config, TASK_CFG, ["TASK15"]).to(dev).train()
    opt = FusedAdamW(model, base_lr=4e-5, weight_decay=1e-4, warmup_steps=100, t_total=100000)
    batch = tuple(t.to(dev) for t in make_batch(256, seed=1))
    crit = torch.nn.CrossEntropyLoss()

    def step():
        loss, _ = task_utils.ForwardModelsTrain(config, TASK_CFG, dev, "TASK15", batch, model, crit)
        loss.backward()
        opt.step()
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    for n in dir(ops):
        f = getattr(ops, n)
        if callable(f) and not n.startswith("_") and getattr(f, "__module__", "") == ops.__name__:
            setattr(ops, n, wrap(n, f))
    n_steps = 10
    t0 = time.perf_counter()
    for _ in range(n_steps):
        step()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("host wall per step %.2f ms (GPU step ~16 ms)" % (1e3 * host / n_steps))
    for name, (cnt, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:14]:
        print("  %-28s %5.1f calls/step  %7.3f ms/step  (%.1f us/call)" % (name, cnt / n_steps, 1e3 * t / n_steps, 1e6 * t / cnt))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Host-side cost of one training step (tiny batch => GPU time negligible): wall time + cProfile hot spots."""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import TASK_CFG, uc2_cfg_dict  # noqa: E402
from clg_vqa_amd import task_utils  # noqa: E402
from clg_vqa_amd.config import BertConfig  # noqa: E402
from clg_vqa_amd.encoders import BertForVLTasks  # noqa: E402
from clg_vqa_amd.optim import FusedAdamW  # noqa: E402
from clg_vqa_amd.synthetic import make_batch  # noqa: E402

config = BertConfig.from_dict(uc2_cfg_dict(vocab=5000))
model = BertForVLTasks(config, TASK_CFG, ["TASK15"]).cuda().train()
opt = FusedAdamW(model, warmup_steps=10, t_total=1000)
batch = tuple(t.cuda() for t in make_batch(8, vocab_size=5000))
crit = torch.nn.CrossEntropyLoss()


def step():
    loss, _ = task_utils.ForwardModelsTrain(config, TASK_CFG, "cuda", "TASK15", batch, model, crit)
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("host enqueue time per step: %.2f ms (B=8, 12 layers)" % ((t1 - t0) / 10 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)

# the trunk's backward runs on autograd's worker thread, which cProfile does not see: call the engine directly
from clg_vqa_amd import ops  # noqa: E402
feats, locs, imask, ids, _, tmask, seg = batch[:7]
eng = model.engine


def direct():
    ops.set_stream(torch.cuda.current_stream().cuda_stream)
    try:
        with torch.no_grad():
            out, sv = eng.forward(ids, feats, locs, seg, tmask, imask, True)
            eng.backward(sv, torch.ones_like(out))
    finally:
        ops.set_stream(None)
        eng.stack.join()


for _ in range(3):
    direct()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    direct()
t1 = time.perf_counter()
torch.cuda.synchronize()
print("engine forward+backward called directly: %.2f ms host time per pass" % ((t1 - t0) / 10 * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    direct()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)

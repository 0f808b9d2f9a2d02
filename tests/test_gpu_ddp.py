"""Two ranks (gloo, both on the one GPU of the box) through the real training step: layer-wise gradient exchange
during backward + sparse word-embedding exchange + remaining dense ranges must give every rank the gradient of the
concatenated batch.  Needs a real MI355X."""
import os
import socket
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

from helpers import TASK_CFG, uc2_cfg_dict  # noqa: E402


def _model_and_batch(seed, B, rank_seed):
    from clg_vqa_amd.config import BertConfig
    from clg_vqa_amd.encoders import BertForVLTasks
    from clg_vqa_amd.synthetic import make_batch
    config = BertConfig.from_dict(uc2_cfg_dict(vocab=999, n_layers=2))
    torch.manual_seed(seed)
    model = BertForVLTasks(config, TASK_CFG, ["TASK15"]).cuda()
    model.eval()  # no dropout: the two-rank run and the single-process run see the same function
    return config, model, make_batch(B, vocab_size=999, seed=rank_seed)


def _one_step(config, model, batch, **kw):
    from clg_vqa_amd import task_utils
    from clg_vqa_amd.optim import FusedAdamW
    opt = FusedAdamW(model, base_lr=1e-4, weight_decay=0.0, correct_bias=True, max_grad_norm=1e9, **kw)
    opt.keep_reduced_grad = True
    loss, _ = task_utils.ForwardModelsTrain(config, TASK_CFG, "cuda", "TASK15", batch, model, torch.nn.CrossEntropyLoss())
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    return opt


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    # replicas are built from DIFFERENT seeds: FusedAdamW's rank-0 broadcast (apex distributed.py:253) aligns them
    config, model, batch = _model_and_batch(3 + 10 * rank, 4, 100 + rank)
    opt = _one_step(config, model, batch)
    assert opt._layer_plan is not None and model.engine.stack.layer_done_hook is not None  # overlap is the default
    torch.save({"grad": (opt.last_reduced_grad * opt.last_post).cpu(), "param": opt.arena.param.cpu()},
               os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_reduce_to_the_gradient_of_the_concatenated_batch():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory() as d:
        ctx = mp.get_context("spawn")
        procs = [ctx.Process(target=_worker, args=(r, 2, port, d)) for r in range(2)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
        got = [torch.load(os.path.join(d, "rank%d.pt" % r), weights_only=True) for r in range(2)]
    # replicas stay bit-identical
    assert torch.equal(got[0]["grad"], got[1]["grad"])
    assert torch.equal(got[0]["param"], got[1]["param"])
    # and equal the single-process gradient of the concatenated batch (mean loss over 8 = average of the two means)
    from clg_vqa_amd.synthetic import make_batch
    config, model, _ = _model_and_batch(3, 4, 100)
    both = tuple(torch.cat([a, b]) for a, b in zip(make_batch(4, vocab_size=999, seed=100), make_batch(4, vocab_size=999, seed=101)))
    opt = _one_step(config, model, both, overlap_reduce=False)
    ref = (opt.last_reduced_grad * opt.last_post).cpu()
    scale = ref.abs().max().item()
    err = (got[0]["grad"] - ref).abs().max().item()
    assert err <= 2e-3 * scale, (err, scale)  # backward GEMMs round their operands to bf16 per batch composition


def test_device_prefetcher_stages_batches_in_order_and_unchanged():
    from clg_vqa_amd.data import DevicePrefetcher
    from clg_vqa_amd.synthetic import make_batch
    batches = [make_batch(3, vocab_size=999, seed=s) for s in range(7)]
    got = list(DevicePrefetcher(iter(batches), "cuda", depth=2))
    assert len(got) == len(batches)
    for ref, dev in zip(batches, got):
        assert len(ref) == len(dev)
        for a, b in zip(ref, dev):
            assert b.is_cuda and b.dtype == a.dtype and torch.equal(b.cpu(), a)
    with pytest.raises(RuntimeError, match="no CPU path"):
        DevicePrefetcher(iter(batches), "cpu")

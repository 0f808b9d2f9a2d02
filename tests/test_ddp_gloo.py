"""world_size-2 gloo tests (CPU) of the multi-GPU host logic: flat-arena gradient reducer, the reference's
parameter grouping and LR schedule.  The path is pure data parallelism: the only exchange is one gradient
all-reduce(SUM) per optimizer step followed by 1/world_size (apex DDP, distributed.py:425-475)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from clg_vqa_amd.optim import FlatArena, GradReducer, reference_param_groups, warmup_linear


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # DDP start: replicas seeded DIFFERENTLY end up bit-identical to rank 0 (apex distributed.py:253)
    torch.manual_seed(1000 + rank)
    l0 = torch.nn.Linear(9, 5)
    a0 = FlatArena(reference_param_groups([("w", l0.weight), ("b", l0.bias)], 1e-3, 0.0), torch.device("cpu"))
    before = a0.param.clone()
    a0.broadcast_from_rank0()
    q.put(("bcast", rank, before.tolist(), a0.param.tolist(), l0.weight.detach().reshape(-1).tolist()))
    torch.manual_seed(0)  # identical replicas
    lin = torch.nn.Linear(37, 11)
    ln = torch.nn.LayerNorm(11)
    named = [("bert.x.dense.weight", lin.weight), ("bert.x.dense.bias", lin.bias),
             ("bert.x.LayerNorm.weight", ln.weight), ("bert.x.LayerNorm.bias", ln.bias)]
    groups = reference_param_groups(named, 4e-5, 1e-4)
    arena = FlatArena(groups, torch.device("cpu"))
    assert lin.weight.data_ptr() == arena.param.data_ptr()  # parameters were re-homed into the arena
    x = torch.randn(5, 37, generator=torch.Generator().manual_seed(100 + rank))
    ln(lin(x)).pow(2).sum().backward()
    local = [p.grad.clone() for _, p, _, _ in groups]
    arena.gather_grads()
    assert all(p.grad is None for _, p, _, _ in groups)
    red = GradReducer(bucket_bytes=256)  # tiny buckets -> several collectives
    post = red.allreduce_(arena.grad)
    assert post == 1.0 / world
    # sparse exchange of embedding rows: duplicates inside a rank, overlaps across ranks, and a skipped dense range
    gen = torch.Generator().manual_seed(7 + rank)
    ids = torch.randint(0, 13, (10,), generator=gen)
    rows = torch.randn(10, 4, generator=gen)
    table = torch.zeros(13, 4)
    red.exchange_sparse_rows(ids, rows, table,
                             lambda i, r, t: t.index_add_(0, i[i >= 0], r[i >= 0]))
    dense = torch.arange(20, dtype=torch.float32) * (rank + 1)
    red.allreduce_(dense, skip=(4, 12))
    q.put(("sparse", rank, ids.tolist(), rows.tolist(), table.tolist(), dense.tolist()))
    gathered = [torch.zeros_like(arena.grad) for _ in range(world)]
    q.put((rank, [g.tolist() for g in local], arena.grad.tolist(), arena.offsets))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_arena_allreduce_world2():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(3 * world)]
    bc = sorted([g for g in got if g[0] == "bcast"])
    got = [g for g in got if g[0] != "bcast"]
    assert bc[0][2] != bc[1][2]                      # the replicas really started from different parameters
    assert bc[0][3] == bc[1][3] == bc[0][2]          # ... and both hold rank 0's afterwards, bit for bit
    assert bc[1][4] == bc[0][2][:45]                 # the module's parameter is a view of the arena
    sparse = sorted([g for g in got if g[0] == "sparse"])
    res = sorted([g for g in got if g[0] != "sparse"])
    ref = torch.zeros(13, 4)
    for _, _, ids, rows, _, _ in sparse:
        ref.index_add_(0, torch.tensor(ids), torch.tensor(rows))
    for _, rank, _, _, table, dense in sparse:
        torch.testing.assert_close(torch.tensor(table), ref, rtol=1e-6, atol=1e-6)
        d = torch.tensor(dense)
        base = torch.arange(20, dtype=torch.float32)
        assert torch.equal(d[4:12], base[4:12] * (rank + 1))          # skipped range untouched
        assert torch.equal(d[:4], base[:4] * 3) and torch.equal(d[12:], base[12:] * 3)
    assert sparse[0][4] == sparse[1][4]                                # bit-identical on both ranks
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, g0, flat0, offs), (_, g1, flat1, _) = res
    assert flat0 == flat1  # both ranks hold the same reduced gradient
    flat = torch.tensor(flat0)
    for i, off in enumerate(offs):
        a, b = torch.tensor(g0[i]).reshape(-1), torch.tensor(g1[i]).reshape(-1)
        torch.testing.assert_close(flat[off:off + a.numel()], a + b, rtol=1e-6, atol=1e-6)


def test_param_groups_and_schedule():
    w, b = torch.nn.Parameter(torch.zeros(3, 3)), torch.nn.Parameter(torch.zeros(3))
    named = [("bert.encoder.layer.0.attention_output.dense.weight", w),
             ("bert.encoder.layer.0.attention_output.dense.bias", b),
             ("bert.encoder.layer.0.attention_output.LayerNorm.weight", torch.nn.Parameter(torch.ones(3))),
             ("vil_prediction.weight", torch.nn.Parameter(torch.ones(3))),
             ("alias.of.w", w)]
    g = reference_param_groups(named, 4e-5, 1e-4)
    assert [(n, lr, wd) for n, _, lr, wd in g] == [
        ("bert.encoder.layer.0.attention_output.dense.weight", 4e-5, 1e-4),
        ("bert.encoder.layer.0.attention_output.dense.bias", 4e-5, 0.0),
        ("bert.encoder.layer.0.attention_output.LayerNorm.weight", 4e-5, 0.0),
        ("vil_prediction.weight", 1e-4, 1e-4)]
    assert warmup_linear(0, 10, 100) == 0.0 and warmup_linear(5, 10, 100) == 0.5
    assert warmup_linear(10, 10, 100) == 1.0 and abs(warmup_linear(55, 10, 100) - 0.5) < 1e-12
    assert warmup_linear(100, 10, 100) == 0.0

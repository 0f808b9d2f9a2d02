#!/usr/bin/env python3
"""bench.py -- VQA train samples/sec (UC2, 36 boxes, seq 56, bs 256 per GPU) on N MI355X of one node.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full training step of BASELINE.json configs[1] on one synthetic batch already resident in HBM:
forward (dropout on) + GQA loss with semantic prior + backward + gradient all-reduce (N > 1) + clip + AdamW +
zero-grad.  Weak scaling: 256 samples per GPU.  Rank 0 prints ONE JSON line.

Extra objects on the line:
  roofline      the dominant kernel (the 3-pass bf16 MFMA GEMM of the forward): algorithmic FLOP (2*M*N*K per launch,
                NOT x3 for the split passes) / HIP-event-measured duration over the timed region, against the dense
                bf16 MFMA peak (2.5 PFLOP/s); "mfma_issue_frac" = passes x that (what the matrix pipe really issues).
  cpu_baseline  the oracle (CPU restatement of the reference, kind "port") timed on the box's host cores on a
                bounded sample: UC2 full config, micro-batch 32, fwd+bwd, N = 1 / rank 0 only.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def uc2_full_cfg():
    from helpers import uc2_cfg_dict
    return uc2_cfg_dict()  # volta/config/uc2_base.json values: 12 layers, H 768, 12 heads, I 3072, vocab 250002


class GemmTimer(object):
    """HIP-event pairs around launches of the dominant kernel, on the stream they are launched on.  Every STRIDE-th
    launch of the timed region is bracketed (an event pair costs ~12 us of host time and a bubble on the stream; timing
    all 84 GEMMs of a step made the step 4 % slower than it is).  84 launches per step and STRIDE = 5 are co-prime, so
    over the timed steps every GEMM of the step is sampled equally often."""
    STRIDE = 5

    def __init__(self):
        self.records = []  # (passes, flops, ev0, ev1)
        self.enabled = False
        self.count = 0

    def wrap(self, ops_mod):
        inner = ops_mod.gemm_nt
        timer = self

        def gemm_nt(a_hi, a_lo, b_hi, b_lo, M, N, K, passes, epilogue, **kw):
            if not timer.enabled:
                return inner(a_hi, a_lo, b_hi, b_lo, M, N, K, passes, epilogue, **kw)
            timer.count += 1
            if timer.count % timer.STRIDE:
                return inner(a_hi, a_lo, b_hi, b_lo, M, N, K, passes, epilogue, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            inner(a_hi, a_lo, b_hi, b_lo, M, N, K, passes, epilogue, **kw)
            e1.record()
            opb = 2 * (2 if passes == 3 else 1)  # bytes per operand element: bf16 hi (+ lo)
            nbytes = opb * (M * K + N * K) + sum((4 if k in ("resid", "out32") else 2) * M * N
                                                 for k in ("resid", "out32", "out_hi", "out_lo", "aux16") if kw.get(k) is not None)
            timer.records.append((passes, 2.0 * M * N * K, e0, e1, nbytes))

        ops_mod.gemm_nt = gemm_nt

    def summary(self):
        out = {}
        for passes in (1, 3):
            recs = [r for r in self.records if r[0] == passes]
            if recs:
                ms = sum(r[2].elapsed_time(r[3]) for r in recs)
                out[passes] = dict(launches=len(recs), flops=sum(r[1] for r in recs), ms=ms,
                                   bytes_per_launch=sum(r[4] for r in recs) / len(recs))
        return out


def pmc_traffic_per_launch(prefix="gemm3_kernel<3,"):
    """HBM bytes per launch of the dominant kernel from the newest committed PMC summary (profiles/r*_summary.json,
    produced by tools/profile_summary.py from separate rocprofv3 --pmc passes; FETCH_SIZE doubled for gfx950)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")))
    if not files:
        return None
    try:
        pm = json.load(open(files[-1])).get("pmc", {})
        tot, n = 0.0, 0
        for k, v in pm.items():
            if k.startswith(prefix):
                tot += (v["hbm_read_MB_per_launch"] + v["hbm_write_MB_per_launch"]) * 1e6 * v["launches"]
                n += v["launches"]
        return round(tot / n) if n else None
    except Exception:
        return None


def cpu_baseline(seconds_budget=25.0):
    """Oracle fwd+bwd on the host cores: UC2 full config, micro-batch 32."""
    from helpers import TASK_CFG
    from oracle import uc2_oracle as O
    from clg_vqa_amd.config import BertConfig
    from clg_vqa_amd.synthetic import make_batch
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)  # a one-GPU box owns a 16-core share of the host; more threads only oversubscribe it
    torch.set_num_threads(cores)
    print("[bench] cpu_baseline: building the oracle (UC2 full config) on %d host threads" % cores, file=sys.stderr, flush=True)
    config = BertConfig.from_dict(uc2_full_cfg())
    torch.manual_seed(0)
    model = O.OracleUC2ForVLTasks(config, TASK_CFG, ["TASK15"])
    for m in model.modules():
        if isinstance(m, (torch.nn.Linear, torch.nn.Embedding)):
            m.weight.data.normal_(0.0, 0.02)
    model.train()
    mb = 32
    batch = make_batch(mb, seed=99)
    times = []
    t_start = time.time()
    for i in range(4):
        t0 = time.time()
        model.zero_grad()
        loss, _, _ = O.forward_train(model, batch)
        loss.backward()
        dt = time.time() - t0
        print("[bench] cpu_baseline: step %d took %.2f s" % (i, dt), file=sys.stderr, flush=True)
        if i > 0:
            times.append(dt)
        if time.time() - t_start > seconds_budget and len(times) >= 2:
            break
    times.sort()
    med = times[len(times) // 2]
    return dict(value=mb / med, unit="samples/s", cores=cores, kind="port",
                sample="oracle (CPU restatement of volta BertForVLTasks, fp32 eager torch) UC2 full config, "
                       "micro-batch 32, T=20 V=36, fwd+bwd with dropout, median of %d steps after 1 warm-up"
                       % len(times))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="samples per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sft", action="store_true", help="configs[2]: train under a Bernoulli(0.59) SFT mask")
    ap.add_argument("--workload", default="c2", choices=["c2", "c3", "c4", "c5"],
                    help="BASELINE.json configs[1..4]: c2 UC2 dense (the headline), c3 = c2 + SFT masks, c4 M3P with 100 "
                         "boxes, c5 UC2 with 100 boxes + SFT at bs 128")
    ap.add_argument("--no-extras", action="store_true", help="skip the measurements outside the timed region "
                    "(forward+backward only, PCIe-inclusive, reference batch semantics): used under the profiler")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "the multi-rank logic on a single GPU)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    local_rank = local_rank % max(1, torch.cuda.device_count())  # gloo rehearsal: several ranks on one GPU
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from helpers import TASK_CFG
    from clg_vqa_amd import _lib, ops, task_utils
    from clg_vqa_amd.config import BertConfig
    from clg_vqa_amd.encoders import BertForVLTasks
    from clg_vqa_amd.optim import FusedAdamW
    from clg_vqa_amd.synthetic import make_batch

    timer = GemmTimer()
    timer.wrap(ops)

    for kv in filter(None, os.environ.get("VL_DEBUG", "").split(",")):  # tuning knobs, e.g. VL_DEBUG=7:0,8:1
        k, v = kv.split(":")
        _lib.lib().vl_debug_set(int(k), int(v))
    if os.environ.get("VL_LN_BLOCKS"):
        _lib.lib().vl_ln_debug_blocks(int(os.environ["VL_LN_BLOCKS"]))
    num_boxes, num_locs, l2n = 36, 7, False
    if args.workload in ("c3", "c5"):
        args.sft = True
    if args.workload == "c5":
        num_boxes, args.batch = 100, (128 if args.batch == 256 else args.batch)
    torch.manual_seed(1234)  # identical replicas on every rank (apex DDP broadcasts from rank 0 instead)
    if args.workload == "c4":
        from clg_vqa_amd.config import M3PConfig
        from clg_vqa_amd.m3p import M3PForVLTasks
        num_boxes, num_locs, l2n = 100, 5, True
        config = M3PConfig.from_dict(dict(n_heads=12, emb_dim=768, n_layers=12, n_words=250002, vocab_size=250002,
                                          hidden_size=768, pooler_size=768, clf_hidden_size=1536, num_locs=5,
                                          image_embeddings="m3p", model="roberta", fusion_method="text",
                                          norm_embeddings=True))  # volta/config/m3p_base.json
        model = M3PForVLTasks(config, TASK_CFG, ["TASK15"]).to(dev)
    else:
        config = BertConfig.from_dict(uc2_full_cfg())
        model = BertForVLTasks(config, TASK_CFG, ["TASK15"]).to(dev)
    if args.sft:
        from torch.nn.utils import prune
        sys.path.insert(0, ROOT)
        from oracle.uc2_oracle import uc2_prunable_names
        gen = torch.Generator(device="cpu").manual_seed(4321)
        mods = dict(model.named_modules())
        for n in uc2_prunable_names():
            w = mods[n].weight
            mask = (torch.rand(w.shape, generator=gen) < 0.59).float().to(dev)
            w.data.mul_(mask)
            prune.CustomFromMask.apply(mods[n], "weight", mask=mask)
    if os.environ.get("BENCH_PAIR_REDUCE", "1") == "0":
        model.engine.stack.pair_reduce = False
    if os.environ.get("BENCH_EARLY_JOIN", "0") == "1":
        model.engine.stack.early_join = True
    if os.environ.get("BENCH_GROUP_DW", "0") == "1":  # A/B: one grouped weight-gradient launch per layer
        model.engine.stack.group_dw = True
    model.train()
    # reference hyper-parameters: experiments/zero_shot/uc2/xgqa/train.dtu.sh:20-28
    opt = FusedAdamW(model, base_lr=4e-5, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-6, correct_bias=True,
                     max_grad_norm=1.0, warmup_steps=100, t_total=100000,
                     overlap_reduce=False if os.environ.get("BENCH_LAYER_HOOK", "1") == "0" else None)
    opt.flag_sumsq = os.environ.get("BENCH_FLAG_SUMSQ", "1") != "0"
    batch = tuple(t.to(dev) for t in make_batch(args.batch, num_boxes=num_boxes, num_locs=num_locs, l2_normalize=l2n,
                                                seed=1234 + rank))
    crit = torch.nn.CrossEntropyLoss()

    host = [0.0, 0.0, 0.0]  # host-side enqueue seconds (forward, backward, optimizer) -- diagnostics only

    def step():
        c0 = time.perf_counter()
        loss, score = task_utils.ForwardModelsTrain(config, TASK_CFG, dev, "TASK15", batch, model, crit)
        c1 = time.perf_counter()
        loss.backward()
        c2 = time.perf_counter()
        opt.step()
        c3 = time.perf_counter()
        host[0] += c1 - c0
        host[1] += c2 - c1
        host[2] += c3 - c2
        return loss

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    timer.enabled = os.environ.get("BENCH_NO_GEMM_TIMER", "0") != "1"
    host[:] = [0.0, 0.0, 0.0]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    final_loss = float(loss.detach())

    # extras outside the contract's timed region: (i) forward+backward only, (ii) for N > 1 the reference's
    # batch semantics (task_utils.py:478-479 divides the YAML batch over the ranks: global 256, strong scaling)
    def timed(fn, n):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t = torch.tensor([time.perf_counter() - c0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def fwd_bwd():
        loss_, _ = task_utils.ForwardModelsTrain(config, TASK_CFG, dev, "TASK15", batch, model, crit)
        loss_.backward()
        opt.discard_grads()  # no optimizer step: drop the gradients instead of accumulating them over iterations
    n_extra = 0 if args.no_extras else max(3, args.steps // 2)
    fb_rate = None
    if n_extra:
        if world > 1:
            opt.set_overlap(False)  # forward+backward only: no gradient exchange either
        fwd_bwd()
        fb_rate = world * args.batch * n_extra / timed(fwd_bwd, n_extra)
        opt.zero_grad()
        opt.set_overlap(True)
    # (iii) PCIe-inclusive: every step's batch comes from pinned host memory through the copy-stream prefetcher
    h2d_rate = None
    if world == 1 and n_extra:
        from clg_vqa_amd.data import DevicePrefetcher
        pinned = [tuple(t.pin_memory() for t in make_batch(args.batch, num_boxes=num_boxes, num_locs=num_locs, l2_normalize=l2n,
                                                         seed=77 + i)) for i in range(3)]
        feed = DevicePrefetcher((pinned[i % 3] for i in range(n_extra + 2)), dev, depth=2)

        def fed_step():
            b_ = next(feed)
            loss_, _ = task_utils.ForwardModelsTrain(config, TASK_CFG, dev, "TASK15", b_, model, crit)
            loss_.backward()
            opt.step()
        fed_step()
        h2d_rate = args.batch * n_extra / timed(fed_step, n_extra)
    strong_rate = None
    if world > 1 and n_extra and args.workload in ("c2", "c3"):
        small = tuple(t.to(dev) for t in make_batch(max(1, 256 // world), num_boxes=num_boxes, num_locs=num_locs,
                                                    l2_normalize=l2n, seed=4321 + rank))

        def small_step():
            loss_, _ = task_utils.ForwardModelsTrain(config, TASK_CFG, dev, "TASK15", small, model, crit)
            loss_.backward()
            opt.step()
        small_step()
        strong_rate = world * max(1, 256 // world) * n_extra / timed(small_step, n_extra)
    if rank == 0:
        print("[bench] host enqueue per step: fwd %.2f ms, bwd %.2f ms, opt %.2f ms" % tuple(1e3 * h / args.steps for h in host),
              file=sys.stderr, flush=True)

    if rank == 0:
        value = world * args.batch * args.steps / elapsed
        gs = timer.summary()
        roof = None
        if 3 in gs:
            ach = gs[3]["flops"] / (gs[3]["ms"] * 1e-3) / 1e12
            roof = dict(bound="mfma", kernel="gemm3_kernel<3,*> (forward GEMMs, 3-pass split bf16 MFMA, 8-wave ping-pong)", achieved=round(ach, 2),
                        peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s", frac=round(ach / MFMA_BF16_PEAK_TFLOPS, 4),
                        traffic=pmc_traffic_per_launch(), algorithmic_bytes_per_launch=round(gs[3]["bytes_per_launch"]), mfma_passes=3, mfma_issue_frac=round(3 * ach / MFMA_BF16_PEAK_TFLOPS, 4),
                        launches=gs[3]["launches"], avg_launch_us=round(1e3 * gs[3]["ms"] / gs[3]["launches"], 2))
            if 1 in gs:
                a1 = gs[1]["flops"] / (gs[1]["ms"] * 1e-3) / 1e12
                roof["backward_gemm"] = dict(kernel="gemm3_kernel<1,*> (backward dX GEMMs, bf16 MFMA, 8-wave ping-pong)", achieved=round(a1, 2),
                                             frac=round(a1 / MFMA_BF16_PEAK_TFLOPS, 4), launches=gs[1]["launches"],
                                             avg_launch_us=round(1e3 * gs[1]["ms"] / gs[1]["launches"], 2))
        line = {
            "metric": "VQA train samples/sec (UC2, 36 boxes, seq56, bs256)" if args.workload in ("c2", "c3") else
                      "VQA train samples/sec (%s, %d boxes, seq%d, bs%d) [not the headline config]" % (
                          "M3P" if args.workload == "c4" else "UC2", num_boxes, 20 + num_boxes, args.batch),
            "value": round(value, 2), "unit": "samples/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "UC2 full config (12 layers, H768, 12 heads, I3072, vocab 250002), GQA 1842 labels, "
                                   "T=20 + V=36 (S=56), bs %d per GPU, %s fine-tune with_prior, dropout 0.1, "
                                   "full step = fwd+loss+bwd+allreduce+clip+AdamW+zero_grad" % (
                                       args.batch, "SFT-masked" if args.sft else "dense"),
                       "baseline_config": args.workload, "global_batch": world * args.batch, "seq_len": 20 + num_boxes, "parallelism": "dp%d" % world,
                       "precision": "forward GEMMs 3-pass split bf16 MFMA (fp32-grade, logits within 1e-3); attention "
                                    "core fp32 MFMA; backward GEMMs bf16 MFMA; fp32 residual stream / LN / optimizer",
                       "algorithmic_tflop_per_step": round(29.241e-3 * world * args.batch, 3), "final_loss": final_loss},
            "roofline": roof,
        }
        gflop_per_sample = {"c2": 29.241, "c3": 29.241, "c4": 63.72, "c5": 63.71}[args.workload]
        line["step_tflops"] = round(gflop_per_sample * 1e-3 * world * args.batch / (elapsed / args.steps), 2)
        line["extras"] = {"fwd_bwd_only_samples_per_s": None if fb_rate is None else round(fb_rate, 1),
                          "h2d_inclusive_samples_per_s": None if h2d_rate is None else round(h2d_rate, 1),
                          "reference_semantics_global256_samples_per_s": None if strong_rate is None else round(strong_rate, 1)}
        print("[bench] gpu part done: %.1f samples/s, %.2f ms/step" % (value, 1e3 * elapsed / args.steps), file=sys.stderr, flush=True)
        if world == 1 and not args.no_cpu_baseline:
            del model, opt
            torch.cuda.empty_cache()
            line["cpu_baseline"] = cpu_baseline()
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Fused optimizer step + gradient reducer over flat HBM arenas.

Reference semantics restated:
* parameter grouping -- one group per tensor, ``lr = 1e-4 if "vil_" in name else base_lr``, weight decay 0 for names
  containing "bias" / "LayerNorm.bias" / "LayerNorm.weight" (volta/train_task.py:249-260);
* ``pytorch_transformers.optimization.AdamW(lr, eps, betas, correct_bias)`` (call site train_task.py:264-268; the
  package is an un-vendored dependency of the reference -> "parity unpinned" by the reference; its arithmetic is
  restated in oracle/adamw_oracle.py and this kernel path is held to that restatement by
  tests/test_gpu_trajectory.py (8 optimizer steps) and tests/test_gpu_kernels.py::test_adamw_and_sumsq);
* ``WarmupLinearSchedule(warmup_steps, t_total)`` (train_task.py:271-274): lr multiplier step/warmup, then linear
  decay to 0 at t_total;
* ``clip_grad_norm_(model.parameters(), 1.0)`` (train_task.py:330);
* apex DDP with ``delay_allreduce=True``: ONE flat fp32 all-reduce(SUM) after backward, then * 1/world_size
  (volta/apex/apex/parallel/distributed.py:425-475, :491-510).

MI355X design: all parameters live in one flat fp32 arena (``p.data`` are views), with sibling arenas for the
gradient and the two Adam moments.  The native layer stack writes every layer gradient straight into its arena view
(``grad_sink``; only the embeddings / pooler / classifier gradients arrive from autograd and are copied by one
multi-tensor copy); per-layer RCCL all-reduces run on slices of that arena behind each layer's weight-gradient kernels
(no flatten / unflatten copies), the word-embedding rows are exchanged sparsely, then one sum-of-squares kernel and one
AdamW kernel that computes the clip coefficient itself (from device memory -- no host sync) and zeroes the grads.
"""
import torch
import torch.distributed as dist

from . import ops

NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")


def warmup_linear(step, warmup_steps, t_total):
    """pytorch_transformers.WarmupLinearSchedule.lr_lambda."""
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    return max(0.0, float(t_total - step) / float(max(1.0, t_total - warmup_steps)))


def reference_param_groups(named_params, base_lr, weight_decay):
    """[(name, param, lr, wd)] in ``named_parameters`` order (train_task.py:249-260)."""
    out, seen = [], set()
    for name, p in named_params:
        if not p.requires_grad or id(p) in seen:
            continue
        seen.add(id(p))
        lr = 1e-4 if "vil_" in name else base_lr
        wd = 0.0 if any(nd in name for nd in NO_DECAY) else weight_decay
        out.append((name, p, lr, wd))
    return out


class FlatArena(object):
    """Re-homes parameters into one contiguous fp32 buffer (16-byte aligned segments)."""

    def __init__(self, groups, device):
        self.groups = groups
        offs, total = [], 0
        for _, p, _, _ in groups:
            offs.append(total)
            total += (p.numel() + 3) // 4 * 4
        self.offsets, self.total = offs, total
        self.param = torch.zeros(total, dtype=torch.float32, device=device)
        self.grad = torch.zeros(total, dtype=torch.float32, device=device)
        for (name, p, _, _), off in zip(groups, offs):
            view = self.param[off:off + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
        self.grad_views = [self.grad[off:off + p.numel()].view_as(p) for (_, p, _, _), off in zip(groups, offs)]

    def broadcast_from_rank0(self, group=None):
        """Every rank takes rank 0's parameters (apex DistributedDataParallel.__init__,
        volta/apex/apex/parallel/distributed.py:253: ``flat_dist_call([p.data for p in module.parameters()],
        dist.broadcast, (0,))``): replicas that were seeded differently, or loaded a different / late checkpoint, cannot
        silently diverge.  One broadcast of the flat arena (the parameters are views of it)."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.broadcast(self.param, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)

    def gather_grads(self, pre=()):
        """Copy the autograd-produced gradients into the flat gradient arena (params without a grad -- e.g.
        M3P's never-used modules -- contribute zeros, like apex skipping ``grad is None``).  `pre`: indices whose
        gradient already sits in the arena (written there during backward)."""
        dst, src, has = [], [], []
        for i, ((_, p, _, _), gv) in enumerate(zip(self.groups, self.grad_views)):
            has.append(p.grad is not None or i in pre)
            if p.grad is not None:
                if i in pre:
                    raise RuntimeError("clg_vqa_amd.FusedAdamW: parameter %s has an autograd gradient although its "
                                       "gradient was already reduced during backward" % self.groups[i][0])
                dst.append(gv)
                src.append(p.grad)
        if dst:
            torch._foreach_copy_(dst, src)
        for _, p, _, _ in self.groups:
            p.grad = None
        return has


class GradReducer(object):
    """Gradient exchange over RCCL (backend "nccl") or gloo, on the flat gradient arena.

    * dense part: bucketed all-reduce(SUM) on slices of the arena (no flatten / unflatten copies), launched
      asynchronously back-to-back so that RCCL pipelines them over all xGMI links;
    * word-embedding table (68 % of the gradient bytes, but only <= B*T rows touched per rank): exchanged SPARSELY --
      each rank combines duplicate token ids locally, all ranks all-gather (ids, rows) (<= 15.7 MB per rank instead of
      a 768 MB all-reduce) and apply the chunks in rank order (ids are unique inside a chunk, chunks are applied one
      after the other -> every rank computes bit-identical sums, replicas do not drift);
    * the 1/world_size factor is folded into the optimizer's grad scale.
    The reference exchanges everything as ONE dense fp32 bucket (apex distributed.py:491-510)."""

    def __init__(self, bucket_bytes=64 << 20, group=None):
        self.bucket_elems = max(1, bucket_bytes // 4)
        self.group = group

    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def allreduce_(self, flat, skip=None, ranges=None):
        """All-reduce `flat` in buckets; `skip` = (start, end) element range to leave out (sparse segment), or
        `ranges` = explicit list of (start, end) element ranges to reduce."""
        ws = self.world_size()
        if ws == 1:
            return 1.0
        if ranges is None:
            ranges = [(0, flat.numel())] if skip is None else [(0, skip[0]), (skip[1], flat.numel())]
        works = self.allreduce_async(flat, ranges)
        for w in works:
            w.wait()
        return 1.0 / ws

    def allreduce_async(self, flat, ranges):
        """Enqueue the bucketed all-reduce of the given element ranges; returns the work handles (wait() makes the
        current stream wait for the collective)."""
        works = []
        for lo, hi in ranges:
            for s in range(lo, hi, self.bucket_elems):
                works.append(dist.all_reduce(flat[s:min(s + self.bucket_elems, hi)], op=dist.ReduceOp.SUM,
                                             group=self.group, async_op=True))
        return works

    def exchange_sparse_rows(self, ids, rows, table_grad, scatter_fn):
        """ids [R] int64, rows [R,H] fp32 (this rank's touched rows) -> table_grad += sum over ranks."""
        ws = self.world_size()
        R, H = rows.shape
        # combine duplicate ids locally with fixed-size ops only (torch.unique would synchronise with the device to
        # learn its output size): sort, number the runs, add every row into its run's slot
        sid, order = torch.sort(ids)
        first = torch.ones(R, dtype=torch.bool, device=ids.device)
        first[1:] = sid[1:] != sid[:-1]
        slot = torch.cumsum(first, 0) - 1                       # run index of every sorted position
        comb = torch.zeros(R, H, dtype=rows.dtype, device=rows.device)
        comb.index_add_(0, slot, rows.index_select(0, order))
        uid = torch.full((R,), -1, dtype=torch.int64, device=ids.device)
        uid.scatter_(0, slot, sid)                              # duplicates write the same value
        if ws > 1:
            all_ids = torch.empty(ws * R, dtype=torch.int64, device=ids.device)
            all_rows = torch.empty(ws * R, H, dtype=rows.dtype, device=rows.device)
            dist.all_gather_into_tensor(all_ids, uid, group=self.group)
            dist.all_gather_into_tensor(all_rows, comb, group=self.group)
        else:
            all_ids, all_rows = uid, comb
        for r in range(ws):  # rank order; ids unique inside a chunk (id -1 = padding of the fixed-size buffer)
            scatter_fn(all_ids[r * R:(r + 1) * R], all_rows[r * R:(r + 1) * R], table_grad)


class FusedAdamW(object):
    """AdamW(correct_bias) + clip + LR schedule + zero-grad as two kernels over the flat arena."""

    def __init__(self, model, base_lr=4e-5, weight_decay=1e-4, betas=(0.9, 0.999), eps=1e-6, correct_bias=True,
                 max_grad_norm=1.0, warmup_steps=0, t_total=None, reducer=None, overlap_reduce=None, pipeline_update=False):
        params = list(model.named_parameters())
        device = params[0][1].device
        if device.type != "cuda":
            raise RuntimeError("clg_vqa_amd.FusedAdamW: parameters must be on the MI355X (no CPU path)")
        self.model = model
        self.groups = reference_param_groups(params, base_lr, weight_decay)
        self.arena = FlatArena(self.groups, device)
        self.reducer = reducer or GradReducer()
        self.arena.broadcast_from_rank0(self.reducer.group)  # DDP start: rank 0's parameters everywhere
        n = self.arena.total
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=device)
        ends = [off + (p.numel() + 3) // 4 * 4 for (_, p, _, _), off in zip(self.groups, self.arena.offsets)]
        self.seg_end = torch.tensor(ends, dtype=torch.int64, device=device)
        self.seg_lr = torch.tensor([g[2] for g in self.groups], dtype=torch.float32, device=device)
        self.seg_wd = torch.tensor([g[3] for g in self.groups], dtype=torch.float32, device=device)
        self.betas, self.eps, self.correct_bias = betas, eps, correct_bias
        self.max_grad_norm = max_grad_norm
        self.warmup_steps, self.t_total = warmup_steps, t_total
        self.sched_step = 0  # scheduler.step() count (train_task.py:335)
        self.opt_step = 0
        # pytorch_transformers.AdamW keeps state['step'] PER PARAMETER and advances it only on steps where the parameter has
        # a gradient: a segment that first receives one at step k > 1, or misses steps, gets its own bias correction.  The
        # counts live on the host; the device copy is only made (and passed to the kernel) once they stop being uniform
        self.seg_steps = [0] * len(self.groups)
        self._seg_step_dev = None
        self._sumsq = torch.zeros(2, dtype=torch.float32, device=device)  # two accumulators alternate: AdamW zeroes the next
        self._sumsq_ws = torch.empty(2048, dtype=torch.float32, device=device)  # fixed-order reduction of the block sums
        self.return_norm = False  # step() returns the gradient norm tensor (one extra tiny launch) only on request
        if hasattr(model, "mark_weights_dirty"):
            model.mark_weights_dirty()
        self._active, self._sink_index, self.row_flags = None, -1, None
        self._planned, self._loose = frozenset(), None
        eng = getattr(model, "engine", None)
        if eng is not None:  # scatter the word-embedding gradient straight into the (zeroed) arena
            emb = model.bert.embeddings.word_embeddings if hasattr(model.bert, "embeddings") else model.bert.encoder.embeddings
            for i, ((_, p, _, _), gv) in enumerate(zip(self.groups, self.arena.grad_views)):
                if p is emb.weight:
                    eng.word_grad_sink = gv
                    self._sink_index = i
                    # rows of the table that ever received a gradient; all others keep m = v = 0 and only decay
                    self.row_flags = torch.zeros(p.shape[0], dtype=torch.uint8, device=device)
                    eng.word_row_flags = self.row_flags
            eng.defer_word_grad = self.reducer.world_size() > 1
        # multi-GPU: all-reduce each transformer layer's gradients as soon as that layer's backward is enqueued
        # (needs one optimizer step per backward: pass overlap_reduce=False when accumulating gradients)
        self._pre, self._works, self._layer_plan = set(), [], None
        # the update (AdamW + weight preparation) of chunk c = embeddings | layer l | heads on the engine's update stream, the
        # NEXT forward waiting chunk by chunk: 0.6 + 0.2 ms of HBM-bound work per step run under the MFMA-bound forward GEMMs
        # instead of in front of them (the clip norm needs every gradient, so nothing can start before backward has ended).
        # OPT-IN: with it the parameters are final on the caller's stream only after the next forward of the model or after
        # wait_update() -- a training loop that owns its schedule (train_task.py, bench.py) switches it on and calls
        # wait_update() before it reads parameters itself (checkpoints, evaluation outside the model's forward, logging)
        self.pipeline_update = bool(pipeline_update)
        self._chunks = None
        self._ev_start = None
        self.keep_reduced_grad = False
        self.flag_sumsq = True
        if overlap_reduce is None:
            # also with one GPU: the layer gradients then reach the arena by one multi-tensor copy per layer on the
            # weight-gradient stream instead of ~90 autograd copies (views of the packed Q/K/V gradient) + one big
            # gather on the main stream (-0.4 ms per step); further backward passes before step() accumulate
            overlap_reduce = True
        if overlap_reduce and eng is not None and hasattr(eng, "stack"):
            index_of = {id(g[1]): i for i, g in enumerate(self.groups)}
            plan = []
            for sp in eng.stack.specs:
                idx = [index_of.get(id(p)) for p in sp.params()]
                if any(i is None for i in idx):  # frozen / foreign parameter in this layer: leave it to step()
                    plan.append(None)
                    continue
                order = sorted(set(idx))
                ranges, lo, hi, prev = [], None, None, None
                for i in order:  # merge neighbours in the arena into one range
                    b = self.arena.offsets[i]
                    e = b + (self.groups[i][1].numel() + 3) // 4 * 4
                    if prev is not None and i == prev + 1:
                        hi = e
                    else:
                        if lo is not None:
                            ranges.append((lo, hi))
                        lo, hi = b, e
                    prev = i
                ranges.append((lo, hi))
                plan.append((idx, [self.arena.grad_views[i] for i in idx], ranges))
            self._layer_plan = plan
            # steady state of a training step: every planned layer gradient arrives through the sinks; only the `loose`
            # parameters (embeddings, pooler, classifier) still come from autograd
            self._planned = frozenset(i for pl in plan if pl is not None for i in pl[0])
            self._loose = [i for i in range(len(self.groups)) if i not in self._planned]
            eng.stack.layer_done_hook = self._on_layer_grads
            eng.stack.grad_sink = self._sink_for_layer

    def observe_exchange(self, enabled):
        """Multi-GPU observability: bracket the wait for the gradient exchange in step() with HIP events and count the bytes
        (read with exchange_stats(); the events cost two records per step)."""
        self._obs = dict(pairs=[], dense=0, sparse=0, colls=0, steps=0) if enabled else None

    def exchange_stats(self):
        o = getattr(self, "_obs", None)
        if not o or not o["steps"]:
            return dict(wait_ms_per_step=None, dense_bytes_per_step=0, sparse_bytes_per_step=0, collectives_per_step=0)
        torch.cuda.synchronize()
        wait = sum(a.elapsed_time(b) for a, b in o["pairs"]) / o["steps"]
        return dict(wait_ms_per_step=round(wait, 4), dense_bytes_per_step=o["dense"] // o["steps"],
                    sparse_bytes_per_step=o["sparse"] // o["steps"], collectives_per_step=round(o["colls"] / o["steps"], 1))

    def _update_chunks(self, eng):
        """[(elem lo, elem hi, seg lo, seg hi, rebased seg_end on the device)] per chunk, or False when the arena order does
        not allow it (a chunk's parameters are not one contiguous run of segments, in chunk order)."""
        if self._chunks is not None:
            return self._chunks
        self._chunks = False
        if not getattr(eng, "supports_update_pipeline", False):
            return False
        chunk_of = {}
        for p_ in eng.param_list()[:len(eng.param_list()) - 16 * len(eng.stack.specs)]:
            chunk_of[id(p_)] = 0
        for l, sp in enumerate(eng.stack.specs):
            for p_ in sp.params():
                chunk_of[id(p_)] = 1 + l
        last = eng.n_chunks() - 1
        ids = [chunk_of.get(id(g[1]), last) for g in self.groups]
        if any(b < a_ for a_, b in zip(ids, ids[1:])):  # chunk ids must be non-decreasing along the arena
            return False
        ends = [off + (g[1].numel() + 3) // 4 * 4 for g, off in zip(self.groups, self.arena.offsets)]
        chunks, i = [], 0
        for c in range(last + 1):
            j = i
            while j < len(ids) and ids[j] == c:
                j += 1
            if j == i:
                chunks.append(None)  # nothing trainable in this chunk (frozen)
                continue
            lo, hi = self.arena.offsets[i], ends[j - 1]
            seg_end = torch.tensor([e - lo for e in ends[i:j]], dtype=torch.int64, device=self.arena.param.device)
            chunks.append((lo, hi, i, j, seg_end))
            i = j
        self._chunks = chunks
        return chunks

    def set_overlap(self, enabled):
        """Switch the during-backward gradient exchange on / off (off: everything is reduced inside step())."""
        eng = getattr(self.model, "engine", None)
        if eng is not None and hasattr(eng, "stack"):
            eng.stack.layer_done_hook = self._on_layer_grads if (enabled and self._layer_plan is not None) else None

    def _sink_for_layer(self, layer):
        """Destination views (LayerSpec.params order) for the layer's weight-gradient GEMMs + whether they must add to
        what an earlier backward of this step left there."""
        plan = self._layer_plan[layer]
        if plan is None:
            return None
        idx, views, _ = plan
        return views, bool(self._pre.intersection(idx))

    def _on_layer_grads(self, layer, grads, stream):
        plan = self._layer_plan[layer]
        if plan is None:
            return False
        idx, views, ranges = plan
        again = bool(self._pre.intersection(idx))
        if again and self.reducer.world_size() > 1:
            raise RuntimeError("clg_vqa_amd.FusedAdamW: a second backward arrived before step(); construct the optimizer "
                               "with overlap_reduce=False when accumulating gradients over micro-batches")
        with torch.cuda.stream(stream):  # behind this layer's weight-gradient kernels
            # (None = the GEMM already wrote that gradient into its arena view: the six weight matrices, 99.9 % of
            # the layer's gradient bytes; what is copied here are the ten bias / LayerNorm vectors)
            # (None = the native stack already wrote that gradient into its arena view -- all 16 of a layer when the
            # sink is installed; tensors only arrive here from a custom caller)
            dst = [v for g, v in zip(grads, views) if g is not None]
            src = [g.view_as(v) for g, v in zip(grads, views) if g is not None]
            if dst and again:  # one GPU, gradient accumulation: add to what the earlier backward left in the arena
                torch._foreach_add_(dst, src)
            elif dst:
                torch._foreach_copy_(dst, src)
            if self.reducer.world_size() > 1:
                works = self.reducer.allreduce_async(self.arena.grad, ranges)
                self._works += works
                o = getattr(self, "_obs", None)
                if o is not None:
                    o["dense"] += 4 * sum(hi - lo for lo, hi in ranges)
                    o["colls"] += len(works)
        self._pre.update(idx)
        return True

    def wait_update(self):
        """Make the current stream wait for a pipelined update still running on the engine's update stream."""
        eng = getattr(self.model, "engine", None)
        if eng is not None and eng.chunk_events is not None:
            torch.cuda.current_stream().wait_event(eng.chunk_events[-1])  # (chunks are recorded in order on one stream)

    def _step_pipelined(self, a, eng, chunks, seg_step, nxt, kw):
        """AdamW + weight preparation chunk by chunk on the engine's update stream; the next forward waits per chunk."""
        dev = a.param.device
        main = torch.cuda.current_stream()
        upd = eng.update_stream(dev)
        if eng.chunk_events is None:
            eng.chunk_events = [torch.cuda.Event() for _ in range(eng.n_chunks())]
            for e in eng.chunk_events:
                e.record()  # materialises the hipEvent_t handles the native stack waits on
        if self._ev_start is None:
            self._ev_start = torch.cuda.Event()
        self._ev_start.record(main)  # gradients complete, clip norm accumulated
        upd.wait_event(self._ev_start)
        lr_mult = self.lr_mult()
        flags = self._flag_args()
        ops.set_stream(upd.cuda_stream)
        try:
            with torch.cuda.stream(upd):
                first = True
                for c, ch in enumerate(chunks):
                    if ch is not None:
                        lo, hi, i, j, seg_end = ch
                        fl = {}
                        if flags and lo <= flags["flag_begin"] < hi:
                            fl = dict(flags, flag_begin=flags["flag_begin"] - lo)
                        ops.adamw(a.param[lo:hi], a.grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi], seg_end,
                                  self.seg_lr[i:j], self.seg_wd[i:j], self.betas[0], self.betas[1], self.eps, self.opt_step,
                                  self.correct_bias, lr_mult, sumsq_next=nxt if first else None,
                                  seg_step=None if seg_step is None else seg_step[i:j], **kw, **fl)
                        first = False
                    eng.prepare_chunk(c)
                    eng.chunk_events[c].record(upd)
        finally:
            ops.set_stream(None)
        if hasattr(self.model, "mark_weights_dirty"):
            self.model.mark_weights_dirty()  # (module-level Linears outside the engine re-prepare themselves)
        eng.finish_prepare()

    def state_dict(self):
        self.wait_update()
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "opt_step": self.opt_step,
                "seg_steps": list(self.seg_steps), "sched_step": self.sched_step, "names": [g[0] for g in self.groups],
                "row_flags": self.row_flags}

    def load_state_dict(self, sd):
        assert sd["names"] == [g[0] for g in self.groups], "optimizer state belongs to a different parameter list"
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.opt_step, self.sched_step = int(sd["opt_step"]), int(sd["sched_step"])
        self.seg_steps = list(sd["seg_steps"]) if sd.get("seg_steps") is not None else [self.opt_step] * len(self.groups)
        self._sumsq.zero_()  # (the accumulator in use alternates with the parity of opt_step)
        if self.row_flags is not None:
            if sd.get("row_flags") is not None:
                self.row_flags.copy_(sd["row_flags"])
            else:
                self.row_flags.fill_(1)  # unknown history: treat every row as touched (dense update)

    def zero_grad(self):
        self.arena.grad.zero_()
        for _, p, _, _ in self.groups:
            p.grad = None
        for w in self._works:
            w.wait()
        self._works, self._pre = [], set()
        eng = getattr(self.model, "engine", None)
        if eng is not None:
            eng.pending_word_grad = None

    def discard_grads(self):
        """Forget the gradients of the last backward without zero-filling the arena: the next backward overwrites the
        layer gradients (benchmarking forward+backward alone; everything else should use zero_grad())."""
        for w in self._works:
            w.wait()
        self._works, self._pre = [], set()
        for _, p, _, _ in self.groups:
            p.grad = None
        eng = getattr(self.model, "engine", None)
        if eng is not None:
            eng.pending_word_grad = None

    def _flag_args(self):
        if self.row_flags is None:
            return {}
        gv = self.arena.grad_views[self._sink_index]
        return dict(row_flags=self.row_flags, flag_begin=self.arena.offsets[self._sink_index], flag_rows=gv.shape[0],
                    flag_row_len=gv.shape[1])

    def lr_mult(self):
        if self.t_total is None:
            return 1.0
        return warmup_linear(self.sched_step, self.warmup_steps, self.t_total)

    def step(self):
        """reduce -> clip -> AdamW -> scheduler step -> zero_grad  (train_task.py:326-338)."""
        a = self.arena
        pre = self._pre
        if self._loose is not None and len(pre) == len(self._planned) and pre == self._planned:
            # fast path (no walk over the 215 groups): the planned layer gradients sit in the arena already
            dst, src, has_loose = [], [], []
            for i in self._loose:
                g = self.groups[i][1].grad
                has_loose.append(g is not None)
                if g is not None:
                    dst.append(a.grad_views[i])
                    src.append(g)
                    self.groups[i][1].grad = None
            if dst:
                torch._foreach_copy_(dst, src)
            akey = ("fast", tuple(has_loose))
            if akey != self._active:
                hl = dict(zip(self._loose, has_loose))
                has = [True if i in self._planned else hl[i] for i in range(len(self.groups))]
            else:
                has = None
        else:
            has = a.gather_grads(pre)
            akey = ("slow", tuple(has))
        # parameters that received no gradient are skipped entirely, like `if p.grad is None: continue` in
        # pytorch_transformers.AdamW (no moment decay, no weight decay): M3P's 93 M never-used parameters
        if akey != self._active:
            self._active = akey
            self._active_list = active = tuple(h or (i == self._sink_index) for i, h in enumerate(has))
            lr = [g[2] if act else -1.0 for g, act in zip(self.groups, active)]  # negative = skipped by the kernel
            self.seg_lr.copy_(torch.tensor(lr, dtype=torch.float32))
        active = self._active_list
        eng = getattr(self.model, "engine", None)
        skip = None
        if eng is not None and eng.pending_word_grad is not None:  # multi-GPU sparse path
            ids, rows = eng.pending_word_grad
            eng.pending_word_grad = None
            gv = a.grad_views[self._sink_index]
            H = gv.shape[1]
            self.reducer.exchange_sparse_rows(
                ids, rows, gv, lambda i, r, t: ops.embed_scatter_add(i.contiguous(), r.contiguous(), t, i.numel(), H, -1,
                                                                     row_flags=self.row_flags))
            off = a.offsets[self._sink_index]
            skip = (off, off + (gv.numel() + 3) // 4 * 4)
        obs = getattr(self, "_obs", None)
        if obs is not None:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            if skip is not None:
                obs["sparse"] += self.reducer.world_size() * ids.numel() * (8 + 4 * H)
                obs["colls"] += 2
        for w in self._works:  # layer exchanges launched during backward
            w.wait()
        self._works = []
        if self.reducer.world_size() > 1:
            # whatever was not exchanged during backward: merged ranges of the segments that have a gradient
            # (skipping e.g. M3P's 93 M never-used parameters and the sparsely exchanged word-embedding table)
            ranges, lo, hi = [], None, None
            for i, act in enumerate(active):
                if not act or i in pre or i == self._sink_index and skip is not None:
                    continue
                b = a.offsets[i]
                e = b + (self.groups[i][1].numel() + 3) // 4 * 4
                if hi is not None and b == hi:
                    hi = e
                else:
                    if lo is not None:
                        ranges.append((lo, hi))
                    lo, hi = b, e
            if lo is not None:
                ranges.append((lo, hi))
            post = self.reducer.allreduce_(a.grad, ranges=ranges)
            if obs is not None:
                obs["dense"] += 4 * sum(hi - lo for lo, hi in ranges)
                obs["colls"] += sum((hi - lo + self.reducer.bucket_elems - 1) // self.reducer.bucket_elems for lo, hi in ranges)
        else:
            post = 1.0
        if obs is not None:
            ev1.record()  # the main stream has now waited for every collective of this step
            obs["pairs"].append((ev0, ev1))
            obs["steps"] += 1
        self._pre = set()
        if self.keep_reduced_grad:  # tests: the summed gradient and the 1/world factor the kernels apply to it
            self.last_reduced_grad, self.last_post = a.grad.clone(), post
        # sum(g^2) into this step's accumulator (zeroed by the previous step's AdamW launch; table rows that never
        # received a gradient are exact zeros: not read); the clip coefficient min(1, max_norm / (||g|| + 1e-6)) with
        # ||g|| of the averaged gradient is computed by the AdamW kernel itself -- no scalar glue kernels in between
        cur, nxt = self._sumsq[self.opt_step % 2:self.opt_step % 2 + 1], self._sumsq[(self.opt_step + 1) % 2:(self.opt_step + 1) % 2 + 1]
        ops.sumsq(a.grad, cur, ws=self._sumsq_ws, **(self._flag_args() if self.flag_sumsq else {}))
        norm = (cur.sqrt() * post) if self.return_norm else None
        self.opt_step += 1
        uniform = True
        for i, act in enumerate(active):
            if act:
                self.seg_steps[i] += 1
                uniform = uniform and self.seg_steps[i] == self.opt_step
        seg_step = None
        if not uniform:  # (rare: a second task head, a parameter that joins late)
            self._seg_step_dev = torch.tensor(self.seg_steps, dtype=torch.int64).to(a.param.device)
            seg_step = self._seg_step_dev
        max_norm = self.max_grad_norm if self.max_grad_norm is not None else float("inf")
        kw = dict(sumsq=cur, max_norm=min(max_norm, 3.0e38), post=post, zero_grad=True)
        chunks = self._update_chunks(eng) if (self.pipeline_update and eng is not None) else False
        if chunks and eng.chunk_tables() is not None:
            self._step_pipelined(a, eng, chunks, seg_step, nxt, kw)
            self.sched_step += 1
            return norm
        ops.adamw(a.param, a.grad, self.exp_avg, self.exp_avg_sq, self.seg_end, self.seg_lr, self.seg_wd,
                  self.betas[0], self.betas[1], self.eps, self.opt_step, self.correct_bias, self.lr_mult(),
                  sumsq_next=nxt, seg_step=seg_step, **kw, **self._flag_args())
        self.sched_step += 1
        if hasattr(self.model, "mark_weights_dirty"):
            self.model.mark_weights_dirty()
        return norm

"""Host-side orchestration of the native UC2 trunk (embeddings + 12 x {attention, feed-forward}).

One ``torch.autograd.Function`` covers the whole trunk: its forward enqueues the HIP kernels of
``include/vlhip.h`` on the current stream and keeps the activations the backward needs; its backward enqueues
the backward kernels and returns the parameter gradients.  torch is used for device memory, the stream and the
autograd hand-off at the trunk's output only.

Reference call stack restated here: BertModel.forward (volta/volta/encoders.py:958-1021) ->
UC2Embeddings.forward (volta/volta/embeddings.py:636-669) -> BertEncoder.forward (encoders.py:848-892) ->
24 x {BertGatedAttention :434 | BertGatedFeedForward :576}, in the single-stream form of SURVEY.md Appendix A.

Data layout in HBM (B = batch, T text tokens, V boxes, S = T + V, M = B*S rows, H hidden, I intermediate):
  stream  x32 [M,H] fp32 (residual stream) + (x_hi, x_lo) bf16 split feeding the 3-pass forward GEMMs
  qkv32   [M,3H] fp32, columns [Q|K|V]             ctx (hi,lo) [M,H] bf16
  u16/h   [M,I] bf16 pre-activation / GELU output  weights: (hi,lo) [N,K] + transposed hi [K,N], rebuilt per step
"""
import torch

from . import ops
from .ops import BF16, EPI_BF16, EPI_DGELU_BF16, EPI_F32, EPI_GELU_SPLIT


def _ceil8(n):
    return (n + 7) // 8 * 8


def linear_params(lin):
    """(differentiable weight, mask or None) of a leaf Linear, honouring torch.nn.utils.prune's
    reparametrisation (weight_orig / weight_mask) that the reference's SFT driver installs by module name
    (volta/train_task_sft.py:122-132)."""
    if "weight_orig" in lin._parameters:
        return lin._parameters["weight_orig"], lin._buffers["weight_mask"]
    return lin.weight, None


class PreparedWeight(object):
    """bf16 operand forms of one (possibly packed) Linear weight: (hi, lo) [N,K] and transposed hi [K,Np]."""

    def __init__(self, linears, device, need_t=True):
        self.linears = linears
        self.N = sum(l.out_features for l in linears)
        self.K = linears[0].in_features
        self.Np = _ceil8(self.N)
        self.hi = torch.empty(self.N, self.K, dtype=BF16, device=device)
        self.lo = torch.empty(self.N, self.K, dtype=BF16, device=device)
        self.t_hi = torch.zeros(self.K, self.Np, dtype=BF16, device=device) if need_t else None
        self.bias = None
        self.key = None

    def _key(self):
        k = []
        for l in self.linears:
            w, m = linear_params(l)
            k.append((w.data_ptr(), w._version, None if m is None else (m.data_ptr(), m._version),
                      l.bias.data_ptr(), l.bias._version))
        return tuple(k)

    def descriptors(self):
        """Rows of the vl_weight_prep_multi table (one per source Linear)."""
        rows, r = [], 0
        for l in self.linears:
            w, m = linear_params(l)
            n = l.out_features
            hi, lo = self.hi[r:r + n], self.lo[r:r + n]
            t = None if self.t_hi is None else self.t_hi[:, r:r + n]
            rows.append([w.data_ptr(), 0 if m is None else m.data_ptr(), hi.data_ptr(), lo.data_ptr(),
                         0 if t is None else t.data_ptr(), n, self.K, self.K, 0 if t is None else self.Np])
            r += n
        return rows

    def refresh_bias(self, fingerprint=True):
        self.bias = (self.linears[0].bias.detach() if len(self.linears) == 1
                     else torch.cat([l.bias.detach() for l in self.linears]))
        # after an explicit mark_dirty() the version fingerprint is left empty: the next call that is not preceded by
        # an optimizer step sees a mismatch, prepares once more and records the real fingerprint
        self.key = self._key() if fingerprint else None

    def refresh(self, force=False):
        key = self._key()
        if not force and key == self.key:
            return
        r = 0
        for l in self.linears:
            w, m = linear_params(l)
            n = l.out_features
            ops.weight_prep(w.detach(), m, self.hi[r:r + n], self.lo[r:r + n],
                            None if self.t_hi is None else self.t_hi[:, r:r + n])
            r += n
        self.bias = (self.linears[0].bias.detach() if len(self.linears) == 1
                     else torch.cat([l.bias.detach() for l in self.linears]))
        self.key = key


def _src_ptrs(lin):
    w, m = linear_params(lin)
    return (w.data_ptr(), 0 if m is None else m.data_ptr())


def dw_gemm(dy16, x16, M, N, K):
    """dW[N,K] = dY[M,N]^T . X[M,K] (bf16 operands, fp32 out) as an NT GEMM over transposed copies."""
    dev = dy16.device
    dW = torch.empty(N, K, dtype=torch.float32, device=dev)
    if ops.gemm_tn_splitk(dy16, x16, N, K, M, dW):
        return dW
    Mp = _ceil8(M)
    alloc = torch.zeros if Mp != M else torch.empty
    dyT = ops._tmp(alloc(N, Mp, dtype=BF16, device=dev))
    xT = ops._tmp(alloc(K, Mp, dtype=BF16, device=dev))
    ops.transpose_bf16(dy16, dyT, M, N)
    ops.transpose_bf16(x16, xT, M, K)
    ops.gemm_nt_splitk(dyT, xT, N, K, Mp, dW)
    return dW


class LayerSpec(object):
    """Leaf modules of one post-LN transformer layer (UC2: attention sub-layer 2l + feed-forward sub-layer 2l+1;
    M3P: attentions[l] / layer_norm1[l] / ffns[l] / layer_norm2[l])."""

    def __init__(self, q, k, v, o, ln1, w1, w2, ln2):
        self.q, self.k, self.v, self.o, self.ln1, self.w1, self.w2, self.ln2 = q, k, v, o, ln1, w1, w2, ln2

    def params(self):
        ps = []
        for lin in (self.q, self.k, self.v, self.o):
            ps += [linear_params(lin)[0], lin.bias]
        ps += [self.ln1.weight, self.ln1.bias]
        for lin in (self.w1, self.w2):
            ps += [linear_params(lin)[0], lin.bias]
        ps += [self.ln2.weight, self.ln2.bias]
        return ps


def _masked(dw, lin):
    m = linear_params(lin)[1]
    if m is not None:  # SFT: grad(weight_orig) = grad(weight) (*) mask  (train_task_sft.py:128-132 autograd)
        ops.mask_mul(dw, m, dw)
    return dw


class LayerStack(object):
    """N x { QKV GEMM -> fused attention -> out-proj GEMM -> dropout+residual+LN -> FFN1 GEMM (+GELU) -> FFN2 GEMM ->
    dropout+residual+LN [-> * row mask] } on the single [B*S, H] stream, forward and backward."""

    def __init__(self, specs, H, nh, I, eps):
        self.specs, self.H, self.nh, self.I, self.eps = specs, H, nh, I, eps
        # weight-gradient GEMMs (dW = dY^T X, bias column sums) are off the backward critical path: they run on a
        # second HIP stream so that their workgroups fill the CUs the dX / LayerNorm / attention kernels leave idle
        self.overlap_dw = True
        self._pending = None
        self.side_reduce = False  # LayerNorm-backward column sums on the weight-gradient stream: measured +0.1 ms (that stream is the longer one)
        self.pair_reduce = True  # the two LayerNorm-backward column-sum reductions of a layer share one launch
        self.early_join = False  # A/B knob: join the streams at the end of the layer stack instead of the trunk
        self.layer_done_hook = None  # callable(layer, grads in LayerSpec.params order, stream) -> consumed?
        # callable(layer) -> (16 destination views in LayerSpec.params order, accumulate) or None: when the optimizer
        # provides it, the weight-gradient GEMMs write straight into its flat gradient arena (no gradient copies)
        self.grad_sink = None
        self._fork = None
        self.group_dw = False  # one grouped launch per layer (ops.gemm_tn_grouped): measured equal in situ, see DESIGN.md
        self._side = None

    def make_prepared(self, device):
        return [dict(qkv=PreparedWeight([sp.q, sp.k, sp.v], device), o=PreparedWeight([sp.o], device),
                     w1=PreparedWeight([sp.w1], device), w2=PreparedWeight([sp.w2], device)) for sp in self.specs]

    def forward(self, pw_layers, x32, x_hi, x_lo, am, B, S, p_hid, p_att, seed, row_post=None):
        H, I, nh, M, dev = self.H, self.I, self.nh, B * S, x32.device
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        b16 = lambda *s: torch.empty(*s, dtype=BF16, device=dev)  # noqa: E731
        saved = []
        for l, (sp, lw) in enumerate(zip(self.specs, pw_layers)):
            ls = dict(x_hi=x_hi)
            qkv32 = f32(M, 3 * H)
            ops.gemm_nt(x_hi, x_lo, lw["qkv"].hi, lw["qkv"].lo, M, 3 * H, H, 3, EPI_F32, bias=lw["qkv"].bias,
                        out32=qkv32)
            ctx_hi, ctx_lo, lse = b16(M, H), b16(M, H), f32(B * nh * S)
            ops.attn_fwd(qkv32, am, ctx_hi, ctx_lo, lse, B, S, nh, 64, p_att, seed(16 * l + 3))
            z1 = f32(M, H)
            ops.gemm_nt(ctx_hi, ctx_lo, lw["o"].hi, lw["o"].lo, M, H, H, 3, EPI_F32, bias=lw["o"].bias, out32=z1)
            x1_32, x1_hi, x1_lo, mean1, rstd1 = f32(M, H), b16(M, H), b16(M, H), f32(M), f32(M)
            ops.ln_fwd(z1, x32, None, sp.ln1.weight.detach(), sp.ln1.bias.detach(), self.eps, x1_32, x1_hi, x1_lo,
                       mean1, rstd1, M, H, p_pre=p_hid, seed=seed(16 * l + 4))
            u16, h_hi, h_lo = b16(M, I), b16(M, I), b16(M, I)
            ops.gemm_nt(x1_hi, x1_lo, lw["w1"].hi, lw["w1"].lo, M, I, H, 3, EPI_GELU_SPLIT, bias=lw["w1"].bias,
                        out_hi=h_hi, out_lo=h_lo, aux16=u16)
            z2 = f32(M, H)
            ops.gemm_nt(h_hi, h_lo, lw["w2"].hi, lw["w2"].lo, M, H, I, 3, EPI_F32, bias=lw["w2"].bias, out32=z2)
            x2_32, x2_hi, x2_lo, mean2, rstd2 = f32(M, H), b16(M, H), b16(M, H), f32(M), f32(M)
            ops.ln_fwd(z2, x1_32, None, sp.ln2.weight.detach(), sp.ln2.bias.detach(), self.eps, x2_32, x2_hi, x2_lo,
                       mean2, rstd2, M, H, p_pre=p_hid, seed=seed(16 * l + 5), row_post=row_post)
            ls.update(qkv32=qkv32, ctx_hi=ctx_hi, ctx_lo=ctx_lo, lse=lse, z1=z1, mean1=mean1, rstd1=rstd1,
                      x1_hi=x1_hi, u16=u16, h_hi=h_hi, z2=z2, mean2=mean2, rstd2=rstd2)
            saved.append(ls)
            x32, x_hi, x_lo = x2_32, x2_hi, x2_lo
        return x32, x_hi, x_lo, saved

    def backward(self, pw_layers, saved, dy, am, B, S, p_hid, p_att, seed, ws, row_post=None):
        """dy [M,H] fp32 = dL/d(stack output).  Returns (dL/d(stack input), per-layer grads in LayerSpec.params order)."""
        H, I, nh, M, dev = self.H, self.I, self.nh, B * S, dy.device
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        b16 = lambda *s: torch.empty(*s, dtype=BF16, device=dev)  # noqa: E731
        layer_grads = [None] * len(self.specs)
        main = torch.cuda.current_stream()
        side = None
        if self.overlap_dw:
            if self._side is None or self._side.device != dev:
                self._side = torch.cuda.Stream(device=dev)
            side = self._side
        keep = []  # operands of side-stream kernels stay referenced until the streams are joined
        # one grouped weight-gradient launch per layer needs B*S % 64 == 0 and 8-aligned feature sizes
        grouped = self.group_dw and M % 64 == 0 and H % 8 == 0 and I % 8 == 0
        defer_red = side is not None and self.side_reduce
        pair_red = self.pair_reduce and not defer_red
        ws_b = ops.ln_bwd_ws(M, H, dev) if pair_red else None  # second workspace: both sets of partials are live

        main_ptr = main.cuda_stream
        side_ptr = side.cuda_stream if side is not None else None
        if self._fork is None:
            self._fork = torch.cuda.Event()  # re-recorded for every fork: a wait captures the record enqueued before it

        def on_side(fn, *tensors):
            """Run fn (weight-gradient work) on the side stream after everything enqueued so far on the main one.
            torch's current stream stays the main one (no stream context switch: ~30 us each): the launches are routed
            by handle, results are allocated from the main stream's pool and only consumed after the join, operands
            and wrapper temporaries stay referenced in `keep` until then."""
            if side is None:
                return fn()
            self._fork.record(main)
            side.wait_event(self._fork)
            keep.extend(tensors)
            ops.set_stream(side_ptr, hold=keep)
            try:
                return fn()
            finally:
                ops.set_stream(main_ptr)

        def dw_to(dy16, x16, n_out, k_in, views, accumulate, lins):
            """Weight gradient(s) of the Linear(s) `lins` (packed along the output rows): into the optimizer's arena
            views when there are any and the shape is on the TN fast path (-> [None, ...]), else as tensors."""
            n = len(lins)
            if views is not None and ops.gemm_tn_splitk_to(dy16, x16, n_out, k_in, M, views, accumulate=accumulate):
                for v, lin in zip(views, lins):
                    _masked(v, lin)
                return [None] * n
            dW = dw_gemm(dy16, x16, M, n_out, k_in)
            rows = n_out // n
            return [_masked(dW[i * rows:(i + 1) * rows] if n > 1 else dW, lin) for i, lin in enumerate(lins)]

        for l in reversed(range(len(self.specs))):
            sp, lw, ls = self.specs[l], pw_layers[l], saved[l]
            sink = self.grad_sink(l) if (self.grad_sink is not None and self.layer_done_hook is not None) else None
            sv_, sacc = sink if sink is not None else (None, False)
            pick = (lambda *idx: [sv_[i] for i in idx]) if sv_ is not None else (lambda *idx: None)  # noqa: E731
            dz2, dt2 = f32(M, H), b16(M, H)
            dg2, db2, dbias2 = f32(H), f32(H), f32(H)
            # the column sums (dgamma, dbeta, the dense layer's bias gradient) are only needed by the optimizer: the
            # main stream leaves per-workgroup partials (own workspace per call) and the 15-us reduce launch goes to
            # the weight-gradient stream instead of sitting between the kernels of the critical path
            ws2 = ops.ln_bwd_ws(M, H, dev) if defer_red else ws
            if defer_red:
                ops.ln_bwd(dy, ls["z2"], ls["mean2"], ls["rstd2"], sp.ln2.weight.detach(), dz2, dt2, None, None, None,
                           None, ws2, M, H, p_pre=p_hid, seed=seed(16 * l + 5), row_post=row_post)
                on_side(lambda: ops.ln_bwd_reduce(ws2, M, H, dg2, db2, dbias2), ws2)
            elif pair_red:  # partials only; summed together with the attention sub-layer's LayerNorm below
                ops.ln_bwd(dy, ls["z2"], ls["mean2"], ls["rstd2"], sp.ln2.weight.detach(), dz2, dt2, None, None, None,
                           None, ws_b, M, H, p_pre=p_hid, seed=seed(16 * l + 5), row_post=row_post)
            else:
                ops.ln_bwd(dy, ls["z2"], ls["mean2"], ls["rstd2"], sp.ln2.weight.detach(), dz2, dt2, None, dg2, db2,
                           dbias2, ws, M, H, p_pre=p_hid, seed=seed(16 * l + 5), row_post=row_post)
            du16 = b16(M, I)
            ops.gemm_nt(dt2, None, lw["w2"].t_hi, None, M, I, H, 1, EPI_DGELU_BF16, out_hi=du16, aux16=ls["u16"])
            if not grouped:
                dW2, dW1, dbias1 = on_side(lambda: (dw_to(dt2, ls["h_hi"], H, I, pick(12), sacc, [sp.w2])[0],
                                                    dw_to(du16, ls["x1_hi"], I, H, pick(10), sacc, [sp.w1])[0],
                                                    ops.colsum_bf16(du16, M, I, f32(I))),
                                           dt2, du16, ls["h_hi"], ls["x1_hi"])
            dx1 = f32(M, H)
            ops.gemm_nt(du16, None, lw["w1"].t_hi, None, M, H, I, 1, EPI_F32, resid=dz2, out32=dx1)
            dz1, dt1 = f32(M, H), b16(M, H)
            dg1, db1, dbias_o = f32(H), f32(H), f32(H)
            ws1 = ops.ln_bwd_ws(M, H, dev) if defer_red else ws
            if defer_red:
                ops.ln_bwd(dx1, ls["z1"], ls["mean1"], ls["rstd1"], sp.ln1.weight.detach(), dz1, dt1, None, None, None,
                           None, ws1, M, H, p_pre=p_hid, seed=seed(16 * l + 4))
                on_side(lambda: ops.ln_bwd_reduce(ws1, M, H, dg1, db1, dbias_o), ws1)
            elif pair_red:  # one launch sums the partials of both LayerNorms of the layer
                ops.ln_bwd(dx1, ls["z1"], ls["mean1"], ls["rstd1"], sp.ln1.weight.detach(), dz1, dt1, None, None, None,
                           None, ws, M, H, p_pre=p_hid, seed=seed(16 * l + 4))
                ops.ln_bwd_reduce2(ws_b, M, (dg2, db2, dbias2), ws, M, (dg1, db1, dbias_o), H)
            else:
                ops.ln_bwd(dx1, ls["z1"], ls["mean1"], ls["rstd1"], sp.ln1.weight.detach(), dz1, dt1, None, dg1, db1,
                           dbias_o, ws, M, H, p_pre=p_hid, seed=seed(16 * l + 4))
            if not grouped:
                dWo = on_side(lambda: dw_to(dt1, ls["ctx_hi"], H, H, pick(6), sacc, [sp.o])[0], dt1, ls["ctx_hi"])
            dctx = f32(M, H)
            ops.gemm_nt(dt1, None, lw["o"].t_hi, None, M, H, H, 1, EPI_F32, out32=dctx)
            dqkv = b16(M, 3 * H)
            ops.attn_bwd(ls["qkv32"], am, ls["ctx_hi"], ls["ctx_lo"], dctx, ls["lse"], dqkv, B, S, nh, 64, p_att,
                         seed(16 * l + 3))

            def qkv_grads():
                return (dw_to(dqkv, ls["x_hi"], 3 * H, H, pick(0, 2, 4), sacc, [sp.q, sp.k, sp.v]),
                        ops.colsum_bf16(dqkv, M, 3 * H, f32(3 * H)))

            def layer_grads_grouped():
                # the layer's six weight gradients in ONE launch (no split-K slabs); SFT masks ride in the epilogue
                dWqkv, dWo, dW1, dW2 = f32(3 * H, H), f32(H, H), f32(I, H), f32(H, I)
                mk = lambda lin: linear_params(lin)[1]  # noqa: E731
                probs = [(dqkv[:, i * H:(i + 1) * H], ls["x_hi"], dWqkv[i * H:(i + 1) * H], mk(lin))
                         for i, lin in enumerate((sp.q, sp.k, sp.v))]
                probs += [(dt1, ls["ctx_hi"], dWo, mk(sp.o)), (du16, ls["x1_hi"], dW1, mk(sp.w1)),
                          (dt2, ls["h_hi"], dW2, mk(sp.w2))]
                ops.gemm_tn_grouped(probs, M)
                return (dWqkv, ops.colsum_bf16(dqkv, M, 3 * H, f32(3 * H)), dWo, dW1,
                        ops.colsum_bf16(du16, M, I, f32(I)), dW2)

            if grouped:
                dWqkv, dbqkv, dWo, dW1, dbias1, dW2 = on_side(layer_grads_grouped, dqkv, dt1, du16, dt2, ls["x_hi"],
                                                              ls["ctx_hi"], ls["x1_hi"], ls["h_hi"])
                dWq, dWk, dWv = dWqkv[0:H], dWqkv[H:2 * H], dWqkv[2 * H:]
            else:
                (dWq, dWk, dWv), dbqkv = on_side(qkv_grads, dqkv, ls["x_hi"])
            dx0 = f32(M, H)
            ops.gemm_nt(dqkv, None, lw["qkv"].t_hi, None, M, H, 3 * H, 1, EPI_F32, resid=dz1, out32=dx0)
            layer_grads[l] = [dWq, dbqkv[0:H], dWk, dbqkv[H:2 * H], dWv, dbqkv[2 * H:],
                              dWo, dbias_o, dg1, db1, dW1, dbias1, dW2, dbias2, dg2, db2]
            if self.layer_done_hook is not None:
                # multi-GPU: the optimizer takes this layer's gradients now (copy into its flat arena + asynchronous
                # all-reduce behind the weight-gradient kernels), overlapping the exchange with the rest of backward
                # (the gradient tensors go into `keep`: the hook's copy kernels read them on the other stream after
                # this frame has dropped its references)
                if on_side(lambda: self.layer_done_hook(l, layer_grads[l], side if side is not None else main),
                           *[g for g in layer_grads[l] if g is not None]):
                    layer_grads[l] = [None] * len(layer_grads[l])
            dy = dx0
            saved[l] = None  # release this layer's activations (side-stream operands stay alive through `keep`)
        # the join with the weight-gradient stream is left to the caller (TrunkFunction.backward, after the embedding
        # backward has been enqueued too): the last layer's dW kernels then run under the embedding kernels instead of
        # stalling the main stream (~0.9 ms of tail per step at c2); `keep` lives until then
        self._pending = (side, keep) if side is not None else None
        if self.early_join:
            self.join()
        return dy, layer_grads

    def join(self):
        """Make the current stream wait for the weight-gradient stream; releases the operands kept alive for it."""
        if self._pending is not None:
            side, _ = self._pending
            torch.cuda.current_stream().wait_stream(side)
            self._pending = None


class EngineBase(object):
    """Weight preparation (one launch per step), dropout seeds, dirty tracking shared by the UC2 and M3P engines."""

    def _init_common(self, H, nh):
        if H % nh != 0 or H // nh != 64:
            raise ValueError("clg_vqa_amd: the native attention kernel needs head dim 64 (hidden %d / heads %d)" % (H, nh))
        if H % 256 != 0:
            raise ValueError("clg_vqa_amd: hidden size must be a multiple of 256 for the native LayerNorm")
        self._prepared = None
        self._dirty = True
        # optional: a zero-initialised, optimizer-owned fp32 buffer the word-embedding gradient is scatter-added
        # into directly (FusedAdamW installs a view of its flat gradient arena; the arena is re-zeroed by the
        # optimizer kernel).  Saves the 768 MB zero-fill and the 768 MB copy per step of the dense path.
        self.word_grad_sink = None
        self.word_row_flags = None  # uint8 [vocab]: rows that ever received a gradient (owned by the optimizer)
        # multi-GPU: instead of scattering, hand (token ids, gradient rows) to the reducer, which exchanges the
        # <= B*T touched rows sparsely (all-gather) instead of all-reducing the dense 768 MB table gradient
        self.defer_word_grad = False
        self.pending_word_grad = None
        # counter-based dropout RNG: (base_seed, calls) identify every mask of the run.  base_seed follows torch's seed
        # (--seed -> torch.manual_seed before the model is built), the pair is stored in checkpoints and restored by
        # train_utils.resume, so a resumed run continues the mask sequence instead of replaying it
        self.base_seed = (0x5EED ^ torch.initial_seed()) & 0xFFFFFFFF
        self.calls = 0

    def mark_dirty(self):
        """Call after updating parameters through raw pointers (the fused optimizer does)."""
        self._dirty = True

    def _push_word_grad(self, ids, rows, pad_id):
        rows = rows * (ids != pad_id).to(rows.dtype).unsqueeze(1)  # the pad row receives no gradient
        if self.pending_word_grad is None:
            self.pending_word_grad = (ids, rows)
        else:  # gradient accumulation over micro-batches
            self.pending_word_grad = (torch.cat([self.pending_word_grad[0], ids]),
                                      torch.cat([self.pending_word_grad[1], rows]))

    def next_seed(self):
        self.calls += 1
        seed0 = (self.base_seed * 0x9E3779B1 + self.calls * 0x10001) & 0x7FFFFFFFFFFF
        return lambda site: (seed0 * 4096 + site) & 0xFFFFFFFFFFFFFFFF

    def prepared(self, device):
        if self._prepared is None or self._prepared["device"] != device:
            self._prepared = dict(device=device, img=PreparedWeight([self.image_linear()], device, need_t=False),
                                  layers=self.stack.make_prepared(device))
            self._dirty = True
        pw = self._prepared
        all_pw = [pw["img"]] + [lw[k] for lw in pw["layers"] for k in ("qkv", "o", "w1", "w2")]
        explicit = self._dirty  # set by the optimizer every step: no need to fingerprint versions to find that out
        if explicit or any(p._key() != p.key for p in all_pw):
            # the device table only depends on where the source weights / masks live: compare those pointers first
            ident = tuple(q for p in all_pw for l in p.linears for q in _src_ptrs(l))
            if pw.get("table_ident") != ident:
                rows, tile0 = [], 0
                for r in (r for p in all_pw for r in p.descriptors()):
                    rows.append(r + [tile0])
                    tile0 += ((r[5] + 63) // 64) * ((r[6] + 63) // 64)
                pw["table"] = torch.tensor(rows, dtype=torch.int64).to(device)
                pw["table_ident"], pw["table_tiles"] = ident, tile0
            ops.weight_prep_multi(pw["table"], pw["table"].shape[0], pw["table_tiles"])  # one launch per step
            for p in all_pw:
                p.refresh_bias(fingerprint=not explicit)
        self._dirty = False
        return pw


class UC2Engine(EngineBase):
    """Binds a ``BertForVLTasks`` module tree (reference parameter names) to the native kernels."""

    def __init__(self, model):
        self.model = model
        cfg = model.config
        self.H, self.nh, self.I, self.eps = cfg.hidden_size, cfg.num_attention_heads, cfg.intermediate_size, cfg.layer_norm_eps
        self.n_layers = len(model.bert.encoder.layer) // 2
        self._init_common(self.H, self.nh)
        specs = []
        for l in range(self.n_layers):
            at = model.bert.encoder.layer[2 * l]
            ff = model.bert.encoder.layer[2 * l + 1]
            sa, so = at.attention_self, at.attention_output
            specs.append(LayerSpec(sa.query, sa.key, sa.value, so.dense, so.LayerNorm, ff.intermediate.dense,
                                   ff.output.dense, ff.output.LayerNorm))
        self.stack = LayerStack(specs, self.H, self.nh, self.I, self.eps)

    def image_linear(self):
        return self.model.bert.embeddings.image_embeddings

    # ---- parameters ------------------------------------------------------------------------------------------
    def param_list(self):
        """Differentiable tensors of the trunk in a fixed order (backward returns grads in this order)."""
        e = self.model.bert.embeddings
        ps = [e.word_embeddings.weight, e.position_embeddings.weight, e.new_token_type_embeddings.weight,
              e.LayerNorm.weight, e.LayerNorm.bias, linear_params(e.image_embeddings)[0], e.image_embeddings.bias,
              e.image_location_embeddings.weight, e.image_location_embeddings.bias,
              e.image_layer_norm.weight, e.image_layer_norm.bias,
              e.image_location_layer_norm.weight, e.image_location_layer_norm.bias,
              e.v_LayerNorm.weight, e.v_LayerNorm.bias]
        for sp in self.stack.specs:
            ps += sp.params()
        return ps

    # ---- forward -------------------------------------------------------------------------------------------------
    def forward(self, ids, feats, locs, seg, tmask, imask, training):
        cfg = self.model.config
        emb = self.model.bert.embeddings
        dev = feats.device
        if not feats.is_cuda:
            raise RuntimeError("clg_vqa_amd: BertForVLTasks runs on the MI355X only (no CPU path); move the model "
                               "and the batch to a cuda/HIP device")
        B, T = ids.shape
        V, F = feats.shape[1], feats.shape[2]
        L = locs.shape[2]
        S, H = T + V, self.H
        M, BT, BV = B * S, B * T, B * V
        p_hid = float(cfg.hidden_dropout_prob) if training else 0.0
        p_att = float(cfg.attention_probs_dropout_prob) if training else 0.0
        seed = self.next_seed()
        pw = self.prepared(dev)
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        b16 = lambda *s: torch.empty(*s, dtype=BF16, device=dev)  # noqa: E731
        ids = ids.contiguous(); seg = seg.contiguous()
        tmask = tmask.contiguous().to(torch.int64); imask = imask.contiguous().to(torch.int64)
        feats2 = feats.contiguous().view(BV, F)
        locs2 = locs.contiguous().view(BV, L)
        sv = dict(B=B, T=T, V=V, F=F, L=L, S=S, p_hid=p_hid, p_att=p_att, seed=seed, ids=ids, seg=seg,
                  locs=locs2, pw=pw)

        am = f32(M)
        ops.addmask(tmask, imask, am, B, T, V)
        sv["am"] = am

        x32, x_hi, x_lo = f32(M, H), b16(M, H), b16(M, H)
        # text rows: word + position + type -> LN -> dropout   (embeddings.py:648-655)
        z_t, mean_t, rstd_t = f32(BT, H), f32(BT), f32(BT)
        type_w = emb.new_token_type_embeddings.weight.detach()
        ops.embed_text_fwd(ids, seg, emb.word_embeddings.weight.detach(), emb.position_embeddings.weight.detach(),
                           type_w, z_t, B, T, H, int(cfg.pad_token_id))
        ops.ln_fwd(z_t, None, None, emb.LayerNorm.weight.detach(), emb.LayerNorm.bias.detach(), self.eps, x32, x_hi,
                   x_lo, mean_t, rstd_t, BT, H, group=T, out_stride=S, out_off=0, p_post=p_hid, seed=seed(1))
        # box rows: LN(feat W^T + b) + LN(loc W^T + b) + type[1] -> LN -> dropout   (embeddings.py:660-667)
        f_hi, f_lo = b16(BV, F), b16(BV, F)
        ops.split_f32(feats2, f_hi, f_lo)
        z_i, mean_i, rstd_i, a32 = f32(BV, H), f32(BV), f32(BV), f32(BV, H)
        ops.gemm_nt(f_hi, f_lo, pw["img"].hi, pw["img"].lo, BV, H, F, 3, EPI_F32, bias=pw["img"].bias, out32=z_i)
        ops.ln_fwd(z_i, None, None, emb.image_layer_norm.weight.detach(), emb.image_layer_norm.bias.detach(),
                   self.eps, a32, None, None, mean_i, rstd_i, BV, H)
        z_l, mean_l, rstd_l, b32 = f32(BV, H), f32(BV), f32(BV), f32(BV, H)
        ops.loc_linear_fwd(locs2, emb.image_location_embeddings.weight.detach(),
                           emb.image_location_embeddings.bias.detach(), z_l, BV, L, H)
        ops.ln_fwd(z_l, None, None, emb.image_location_layer_norm.weight.detach(),
                   emb.image_location_layer_norm.bias.detach(), self.eps, b32, None, None, mean_l, rstd_l, BV, H)
        mean_v, rstd_v = f32(BV), f32(BV)
        ops.ln_fwd(a32, b32, type_w[1], emb.v_LayerNorm.weight.detach(), emb.v_LayerNorm.bias.detach(), self.eps,
                   x32, x_hi, x_lo, mean_v, rstd_v, BV, H, group=V, out_stride=S, out_off=T, p_post=p_hid,
                   seed=seed(2))
        sv.update(z_t=z_t, mean_t=mean_t, rstd_t=rstd_t, f_hi=f_hi, z_i=z_i, mean_i=mean_i, rstd_i=rstd_i,
                  z_l=z_l, mean_l=mean_l, rstd_l=rstd_l, z_v=a32, mean_v=mean_v, rstd_v=rstd_v)
        x32, x_hi, x_lo, sv["layers"] = self.stack.forward(pw["layers"], x32, x_hi, x_lo, am, B, S, p_hid, p_att, seed)
        return x32.view(B, S, H), sv

    # ---- backward ------------------------------------------------------------------------------------------------
    def backward(self, sv, dx):
        """dx: [B,S,H] fp32 gradient of the trunk output.  Returns grads in ``param_list`` order."""
        cfg = self.model.config
        emb = self.model.bert.embeddings
        B, T, V, F, L, S = sv["B"], sv["T"], sv["V"], sv["F"], sv["L"], sv["S"]
        H = self.H
        M, BT, BV = B * S, B * T, B * V
        dev = dx.device
        p_hid, p_att, seed, pw, am = sv["p_hid"], sv["p_att"], sv["seed"], sv["pw"], sv["am"]
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        b16 = lambda *s: torch.empty(*s, dtype=BF16, device=dev)  # noqa: E731
        ws = ops.ln_bwd_ws(M, H, dev)
        dy, layer_grads = self.stack.backward(pw["layers"], sv["layers"], dx.contiguous().view(M, H), am, B, S, p_hid,
                                              p_att, seed, ws)

        # --- embeddings backward (dy = dL/dX0 [M,H]) ---
        type_w = emb.new_token_type_embeddings.weight
        sink = self.word_grad_sink
        use_sink = sink is not None and sink.shape == emb.word_embeddings.weight.shape and sink.device == dev
        defer = use_sink and self.defer_word_grad
        dword = None if defer else (sink if use_sink else torch.zeros_like(emb.word_embeddings.weight))
        dpos = torch.zeros_like(emb.position_embeddings.weight)
        dtype_ = torch.zeros_like(type_w)
        # box rows
        dz_v, dg_v, db_v, dtype1 = f32(BV, H), f32(H), f32(H), f32(H)
        ops.ln_bwd(dy, sv["z_v"], sv["mean_v"], sv["rstd_v"], emb.v_LayerNorm.weight.detach(), dz_v, None, None,
                   dg_v, db_v, dtype1, ws, BV, H, group=V, out_stride=S, out_off=T, p_post=p_hid, seed=seed(2))
        dimg16, dg_i, db_i, dbias_img = b16(BV, H), f32(H), f32(H), f32(H)
        ops.ln_bwd(dz_v, sv["z_i"], sv["mean_i"], sv["rstd_i"], emb.image_layer_norm.weight.detach(), None, dimg16,
                   None, dg_i, db_i, dbias_img, ws, BV, H)
        dWimg = _masked(dw_gemm(dimg16, sv["f_hi"], BV, H, F), emb.image_embeddings)
        dloc32, dg_l, db_l = f32(BV, H), f32(H), f32(H)
        ops.ln_bwd(dz_v, sv["z_l"], sv["mean_l"], sv["rstd_l"], emb.image_location_layer_norm.weight.detach(), None,
                   None, dloc32, dg_l, db_l, None, ws, BV, H)
        dWl = torch.zeros_like(emb.image_location_embeddings.weight)
        dbl = torch.zeros_like(emb.image_location_embeddings.bias)
        ops.loc_linear_bwd(sv["locs"], dloc32, dWl, dbl, BV, L, H)
        # text rows
        dz_t, dg_e, db_e = f32(BT, H), f32(H), f32(H)
        ops.ln_bwd(dy, sv["z_t"], sv["mean_t"], sv["rstd_t"], emb.LayerNorm.weight.detach(), dz_t, None, None, dg_e,
                   db_e, None, ws, BT, H, group=T, out_stride=S, out_off=0, p_post=p_hid, seed=seed(1))
        ops.embed_text_bwd(sv["ids"], sv["seg"], dz_t, dword, dpos, dtype_, B, T, H, int(cfg.pad_token_id),
                           row_flags=self.word_row_flags if use_sink else None)
        if defer:
            self._push_word_grad(sv["ids"].view(-1), dz_t, int(cfg.pad_token_id))
        dtype_[1] += dtype1  # image_token_type_embeddings is new_token_type_embeddings (embeddings.py:628)
        grads = [None if use_sink else dword, dpos, dtype_, dg_e, db_e, dWimg, dbias_img, dWl, dbl, dg_i, db_i, dg_l,
                 db_l, dg_v, db_v]
        for lg in layer_grads:
            grads += lg
        return grads


class TrunkFunction(torch.autograd.Function):
    """autograd boundary of a native trunk: (batch, *params) -> X_final [B,S,H]."""

    @staticmethod
    def forward(ctx, engine, training, ids, feats, locs, seg, tmask, imask, *params):
        ops.set_stream(torch.cuda.current_stream().cuda_stream)  # one stream lookup for all launches of the pass
        try:
            out, sv = engine.forward(ids, feats, locs, seg, tmask, imask, training)
        finally:
            ops.set_stream(None)
        ctx.engine = engine
        ctx.sv = sv
        return out

    @staticmethod
    def backward(ctx, dx):
        ops.set_stream(torch.cuda.current_stream().cuda_stream)
        try:
            grads = ctx.engine.backward(ctx.sv, dx)
        finally:
            ops.set_stream(None)
            ctx.engine.stack.join()
        ctx.sv = None
        needs = ctx.needs_input_grad[8:]
        grads = [g if need else None for g, need in zip(grads, needs)]
        return (None,) * 8 + tuple(grads)


UC2TrunkFunction = TrunkFunction

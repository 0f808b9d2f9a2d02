"""The task head of the VL-classifier(-GQA) path on the native kernels: pooler -> dropout -> SimpleClassifier.

Reference: ``BertForVLTasks.forward`` (volta/volta/encoders.py:1231-1259: ``t_pooler`` -> ``dropout`` ->
``clfs_dict[task_id]``), ``BertTextPooler`` (:597-608), ``SimpleClassifier`` (:788-815: Linear -> GeLU -> LayerNorm ->
Linear); ``M3PForVLTasks`` (:1262-1353) with the tanh ``BertPooler`` (m3p/m3p_transformer.py:548-560).

The reference runs the head as ~30 eager kernels forward and ~45 backward on [batch, 768..1842] tensors: pure launch
latency (0.6 ms of an 18 ms step with nothing else on the chip).  Here it is ONE autograd node whose forward is 10
launches and whose backward is 15: the three products go through the small-M GEMM path (csrc/gemm.hip), activation +
dropout + operand split are one kernel per stage (csrc/head.hip), the last LayerNorm of the trunk hands over the (hi, lo)
operand form of the pooled rows, and all three weight gradients + bias gradients are one grouped GEMM launch straight
from the row-major operands (csrc/dw.hip; the bias column sums ride in it) and one column-reduction launch.  Head
weights are prepared by the engine's single per-step weight-preparation launch.

No CPU fallback: module-by-module execution (``encoders.VLLinear`` etc., still native kernels) is only kept for feature
sizes the fused kernels do not cover (not multiples of 64).
"""
import torch

from . import _lib, ops
from .engine import arena_busy, arena_claim, linear_params
from .ops import BF16, EPI_F32, ACT_GELU, ACT_NONE, ACT_RELU, ACT_TANH

HEAD_SEED_SITE = 4090  # dropout site of the pooled vector (the stack uses 16 l + 3 .. 16 l + 5, the embeddings 1 and 2)


def _ceil64(n):
    return (n + 63) // 64 * 64


class HeadArena(object):
    """Buffers of the head for one batch size, allocated once and re-used every step."""

    def __init__(self, B, H, P, C, NL, device):
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)  # noqa: E731
        b16 = lambda *s: torch.empty(*s, dtype=BF16, device=device)  # noqa: E731
        lib = _lib.lib()
        self.B = B
        self.in_flight = False
        self.ticket = None
        self.x_hi, self.x_lo = b16(B, H), b16(B, H)          # only when the trunk did not hand the split over
        self.z1, self.y1_hi, self.y1_lo = f32(B, P), b16(B, P), b16(B, P)
        self.z2, self.g32 = f32(B, C), f32(B, C)
        self.n_hi, self.n_lo, self.mean, self.rstd = b16(B, C), b16(B, C), f32(B), f32(B)
        NLp = _ceil64(NL)
        self.dl16, self.dn, self.dg = b16(B, NLp), f32(B, C), f32(B, C)
        self.dz2_16, self.dy1, self.dz1_16 = b16(B, C), f32(B, P), b16(B, P)
        self.lnws = f32(lib.vl_ln_bwd_ws_floats(B, C))
        img = lambda N: b16(lib.vl_blocked_elems(B, N))  # noqa: E731
        self.t_dl, self.t_n, self.t_dz2, self.t_y1, self.t_dz1, self.t_x = img(NLp), img(C), img(C), img(P), img(P), img(H)
        mb = (B + 63) // 64
        self.cs_dl, self.cs_dz2, self.cs_dz1 = f32(mb, NLp), f32(mb, C), f32(mb, P)


class TaskHead(object):
    """pooler Linear (+ activation) -> dropout -> Linear -> GeLU -> LayerNorm -> Linear of one task."""

    def __init__(self, engine, pooler_dense, act, dropout, clf):
        self.engine = engine
        self.pooler, self.act, self.dropout = pooler_dense, act, dropout
        self.fc1, self.ln, self.fc2 = clf.logit_fc[0], clf.logit_fc[2], clf.logit_fc[3]
        self.H, self.P = pooler_dense.in_features, pooler_dense.out_features
        self.C, self.NL = self.fc1.out_features, self.fc2.out_features
        dims_ok = all(d % 64 == 0 for d in (self.H, self.P, self.C)) and (self.C == 128 or self.C % 256 == 0)
        self.supported = dims_ok and self.fc1.in_features == self.P and self.fc2.in_features == self.C
        self._arenas = {}

    def arena(self, B, device):
        lst = self._arenas.setdefault((B, str(device)), [])
        for a in lst:
            if not arena_busy(a):
                return a
        a = HeadArena(B, self.H, self.P, self.C, self.NL, device)
        lst.append(a)
        if len(lst) > 4:
            raise RuntimeError("clg_vqa_amd: more than 4 training forwards without a backward on one head")
        return a

    def params(self):
        return [linear_params(self.pooler)[0], self.pooler.bias, linear_params(self.fc1)[0], self.fc1.bias,
                self.ln.weight, self.ln.bias, linear_params(self.fc2)[0], self.fc2.bias]

    def __call__(self, x, training):
        return HeadFunction.apply(self, training, x, *self.params())


class HeadFunction(torch.autograd.Function):
    """(trunk output [B, 1 or S, H], head parameters) -> logits [B, num_labels]."""

    @staticmethod
    def forward(ctx, head, training, x, *params):
        if not x.is_cuda:
            raise RuntimeError("clg_vqa_amd: the task head runs on the MI355X only (no CPU path)")
        eng = head.engine
        B, H, P, C, NL = x.shape[0], head.H, head.P, head.C, head.NL
        dev = x.device
        need_grad = eng.grad_mode and any(ctx.needs_input_grad)
        ar = head.arena(B, dev)
        if eng.chunks_pending:  # the optimizer update of the head's parameters may still be running (engine.wait_chunk)
            eng.wait_chunk(eng.n_chunks() - 1)
            eng.chunks_pending = False  # the forward has now waited for every chunk
        pwp, pw1, pw2 = (lin._vl_prepared(dev) for lin in (head.pooler, head.fc1, head.fc2))
        p = float(head.dropout.p) if training else 0.0
        seed = eng.last_seed(HEAD_SEED_SITE) if p > 0.0 else 0
        act = ACT_RELU if head.act == "relu" else ACT_TANH
        ops.set_stream(torch.cuda.current_stream().cuda_stream)
        try:
            split = eng.pooled_split(x)  # (hi, lo) [B, H] written by the trunk's last LayerNorm, when x is its output
            if split is None:
                ops.split_f32(x[:, 0].contiguous(), ar.x_hi, ar.x_lo)
                split = (ar.x_hi, ar.x_lo)
            x_hi, x_lo = split
            ops.gemm_nt(x_hi, x_lo, pwp.hi, pwp.lo, B, P, H, 3, EPI_F32, bias=pwp.bias, out32=ar.z1)
            ops.act_fwd(ar.z1, B, P, act, p, seed, out_hi=ar.y1_hi, out_lo=ar.y1_lo)
            ops.gemm_nt(ar.y1_hi, ar.y1_lo, pw1.hi, pw1.lo, B, C, P, 3, EPI_F32, bias=pw1.bias, out32=ar.z2)
            ops.act_fwd(ar.z2, B, C, ACT_GELU, 0.0, 0, out32=ar.g32)
            ops.ln_fwd(ar.g32, None, None, head.ln.weight.detach(), head.ln.bias.detach(), head.ln.variance_epsilon, None,
                       ar.n_hi, ar.n_lo, ar.mean, ar.rstd, B, C)
            logits = torch.empty(B, NL, dtype=torch.float32, device=dev)
            ops.gemm_nt(ar.n_hi, ar.n_lo, pw2.hi, pw2.lo, B, NL, C, 3, EPI_F32, bias=pw2.bias, out32=logits)
        finally:
            ops.set_stream(None)
        if need_grad:
            ar.in_flight = True
            arena_claim(ctx, ar)
            ctx.head, ctx.ar, ctx.x_hi, ctx.pw = head, ar, x_hi, (pwp, pw1, pw2)
            ctx.p, ctx.seed, ctx.act, ctx.x_shape = p, seed, act, x.shape
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        head, ar, x_hi = ctx.head, ctx.ar, ctx.x_hi
        pwp, pw1, pw2 = ctx.pw
        B, H, P, C, NL = ar.B, head.H, head.P, head.C, head.NL
        dev = dlogits.device
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        NLp = _ceil64(NL)
        ops.set_stream(torch.cuda.current_stream().cuda_stream)
        try:
            # critical path: d(logits) -> d(pooled rows)
            ops.act_bwd(dlogits.contiguous(), None, B, NL, ACT_NONE, 0.0, 0, dz16=ar.dl16)
            ops.gemm_nt(ar.dl16, None, pw2.t_hi, None, B, C, pw2.Np, 1, EPI_F32, out32=ar.dn)
            dg_ln, db_ln = f32(C), f32(C)
            ops.ln_bwd(ar.dn, ar.g32, ar.mean, ar.rstd, head.ln.weight.detach(), ar.dg, None, None, dg_ln, db_ln, None,
                       ar.lnws, B, C)
            ops.act_bwd(ar.dg, ar.z2, B, C, ACT_GELU, 0.0, 0, dz16=ar.dz2_16)
            ops.gemm_nt(ar.dz2_16, None, pw1.t_hi, None, B, P, pw1.Np, 1, EPI_F32, out32=ar.dy1)
            ops.act_bwd(ar.dy1, ar.z1, B, P, ctx.act, ctx.p, ctx.seed, dz16=ar.dz1_16)
            dx0 = f32(B, H)
            ops.gemm_nt(ar.dz1_16, None, pwp.t_hi, None, B, H, pwp.Np, 1, EPI_F32, out32=dx0)
            # weight / bias gradients of the three Linears
            m2, m1, mp = (linear_params(l_)[1] for l_ in (head.fc2, head.fc1, head.pooler))
            db2p, db1, dbp = f32(NLp), f32(C), f32(P)
            dW1, dWp = f32(C, P), f32(P, H)
            if B % 64 == 0 and m2 is None:
                # one grouped GEMM straight from the row-major operands, bias column sums in the same pass, one column
                # reduction (d(logits) is zero-padded to NLp columns: the pad rows of dW2 / db2 are dropped)
                dW2p = f32(NLp, C)
                cs2, cs1, csp = f32((C + 255) // 256, NLp), f32((P + 255) // 256, C), f32((H + 255) // 256, P)
                ops.dw_grouped_rowmajor([(ar.dl16, ar.n_hi, dW2p, None, NLp, C, cs2), (ar.dz2_16, ar.y1_hi, dW1, m1, C, P, cs1),
                                         (ar.dz1_16, x_hi, dWp, mp, P, H, csp)], B)
                ops.colreduce_multi([(cs2, NLp, (db2p,)), (cs1, C, (db1,)), (csp, P, (dbp,))])
                dW2 = dW2p[:NL]
            else:  # ragged batch: one re-layout, one column reduction, one grouped GEMM on the K-major images
                ops.transpose_blocked([(ar.dl16, ar.t_dl, ar.cs_dl), (ar.n_hi, ar.t_n, None), (ar.dz2_16, ar.t_dz2, ar.cs_dz2),
                                       (ar.y1_hi, ar.t_y1, None), (ar.dz1_16, ar.t_dz1, ar.cs_dz1), (x_hi, ar.t_x, None)], B)
                ops.colreduce_multi([(ar.cs_dl, NLp, (db2p,)), (ar.cs_dz2, C, (db1,)), (ar.cs_dz1, P, (dbp,))])
                dW2 = f32(NL, C)
                ops.dw_grouped([(ar.t_dl, 0, NLp, ar.t_n, C, dW2, m2, NL, C), (ar.t_dz2, 0, C, ar.t_y1, P, dW1, m1, C, P),
                                (ar.t_dz1, 0, P, ar.t_x, H, dWp, mp, P, H)], B)
        finally:
            ops.set_stream(None)
        ar.in_flight = False
        if len(ctx.x_shape) == 3 and ctx.x_shape[1] != 1:  # dense trunk output: only row 0 of a sample has a gradient
            dx = torch.zeros(ctx.x_shape, dtype=torch.float32, device=dev)
            dx[:, 0] = dx0
        else:
            dx = dx0.view(ctx.x_shape)
        grads = [dWp, dbp, dW1, db1, dg_ln, db_ln, dW2, db2p[:NL]]
        needs = ctx.needs_input_grad[3:]
        return (None, None, dx if ctx.needs_input_grad[2] else None) + tuple(g if n else None for g, n in zip(grads, needs))

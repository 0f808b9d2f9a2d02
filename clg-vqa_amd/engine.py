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
  qkv     (hi, lo) [M,3H] bf16 x 2, columns [Q|K|V], written by the projection's epilogue and read in place by attention
          (no fp32 copy);  ctx (hi, lo) [M,H] bf16
  u16     [M,I] bf16 = GELU'(pre-activation) for the backward epilogue;  h (hi, lo) [M,I] bf16 = GELU output
  weights (hi, lo) [N,K] + transposed hi [K,N], Q|K|V packed, rebuilt once per step by one launch (SFT mask folded in)
All of it lives in a StackArena allocated once per batch shape; the 12-layer stack is ONE native call per direction.
"""
import weakref

import numpy as np
import torch

from . import _lib, ops
from ._lib import header_constants
from .ops import BF16, EPI_BF16, EPI_DGELU_BF16, EPI_F32, EPI_GELU_SPLIT

VL = header_constants()  # VL_ST_* / VL_LY_* descriptor field indices, parsed from include/vlhip.h


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
        # the bias the GEMM epilogue reads: the parameter itself for one Linear, a persistent packed buffer for Q|K|V
        # (stable pointers: the native stack descriptor holds them)
        self.bias = (linears[0].bias.detach() if len(linears) == 1
                     else torch.empty(self.N, dtype=torch.float32, device=device))
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

    def bias_pairs(self):
        """(destination view, source bias) pairs of a packed bias buffer (empty for a single Linear, whose bias IS the
        parameter): the engine batches the pairs of all layers into one multi-tensor copy per step."""
        if len(self.linears) == 1:
            b = self.linears[0].bias.detach()
            if b.data_ptr() != self.bias.data_ptr():
                self.bias = b
            return []
        pairs, r = [], 0
        for l in self.linears:
            pairs.append((self.bias[r:r + l.out_features], l.bias.detach()))
            r += l.out_features
        return pairs

    def _pack_bias(self):
        pairs = self.bias_pairs()
        if pairs:
            torch._foreach_copy_([d for d, _ in pairs], [s for _, s in pairs])

    def refresh_bias(self, fingerprint=True, pack=True):
        if pack:
            self._pack_bias()
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
        self._pack_bias()
        self.key = key


def _src_ptrs(lin):
    w, m = linear_params(lin)
    return (w.data_ptr(), 0 if m is None else m.data_ptr())


def dw_gemm(dy16, x16, M, N, K, mask=None):
    """dW[N,K] = dY[M,N]^T . X[M,K] (bf16 operands, fp32 out; optional SFT mask in the epilogue).  Whole 64-row blocks
    with feature sizes that are multiples of 8 go straight from the row-major operands through the grouped ping-pong
    kernel (csrc/dw.hip); feature sizes that are multiples of 64 through its K-major form (one re-layout launch);
    anything else through transposed copies + the split-K NT kernel."""
    dev = dy16.device
    dW = torch.empty(N, K, dtype=torch.float32, device=dev)

    def parts_for(tiles):
        # few output tiles x many rows (the image-embedding Linear: 24 tiles x 9216 rows): split the rows over up to 8
        # problems of one launch (fp32 slabs, summed in a fixed order) so that the grid covers the chip
        nblk = M // 64
        return max([s_ for s_ in range(1, 9) if M % 64 == 0 and nblk % s_ == 0 and tiles * s_ <= 256 and nblk // s_ >= 8]
                   or [1])
    tiles = ((N + 255) // 256) * ((K + 255) // 256)
    if M % 64 == 0 and N % 8 == 0 and K % 8 == 0 and dy16.stride(0) % 8 == 0 and x16.stride(0) % 8 == 0:
        # straight from the row-major operands (transposing LDS reads, csrc/dw.hip): no re-layout pass
        parts = parts_for(tiles)
        if parts == 1:
            ops.dw_grouped_rowmajor([(dy16, x16, dW, mask, N, K, None)], M)
            return dW
        slabs = ops._tmp(torch.empty(parts, N * K, dtype=torch.float32, device=dev))
        rows = M // parts
        ops.dw_grouped_rowmajor([(dy16[i * rows:(i + 1) * rows], x16[i * rows:(i + 1) * rows], slabs[i].view(N, K), mask, N, K,
                                  None) for i in range(parts)], rows)
        ops.colreduce_multi([(slabs, N * K, (dW,))])
        return dW
    if N % 64 == 0 and K % 64 == 0 and dy16.stride(0) % 8 == 0 and x16.stride(0) % 8 == 0:
        lib = _lib.lib()
        tA = ops._tmp(torch.empty(lib.vl_blocked_elems(M, N), dtype=BF16, device=dev))
        tB = ops._tmp(torch.empty(lib.vl_blocked_elems(M, K), dtype=BF16, device=dev))
        ops.transpose_blocked([(dy16, tA, None), (x16, tB, None)], M)
        ops.dw_grouped([(tA, 0, N, tB, K, dW, mask, N, K)], M)
        return dW
    if ops.gemm_tn_splitk(dy16, x16, N, K, M, dW):
        return dW if mask is None else ops.mask_mul(dW, mask, dW)
    Mp = _ceil8(M)
    # (torch.zeros would fill on torch's current stream while the launches below may be routed to another one:
    # the pad columns are zeroed by a native memset on the launch stream instead)
    dyT = ops._tmp(torch.empty(N, Mp, dtype=BF16, device=dev))
    xT = ops._tmp(torch.empty(K, Mp, dtype=BF16, device=dev))
    if Mp != M:
        ops.memset_zero(dyT)
        ops.memset_zero(xT)
    ops.transpose_bf16(dy16, dyT, M, N)
    ops.transpose_bf16(x16, xT, M, K)
    ops.gemm_nt_splitk(dyT, xT, N, K, Mp, dW)
    return dW if mask is None else ops.mask_mul(dW, mask, dW)


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


def _f32_bits(x):
    return int(np.float32(x).view(np.uint32))


class ArenaTicket(object):
    """Held by the autograd node of a training forward (``ctx.ticket``).  An arena is busy while its ticket is alive and
    its backward has not run: a forward whose graph is dropped (an exception in the loss, a skipped step, a logging call
    outside no_grad) releases its arena when the node dies instead of pinning it forever."""
    __slots__ = ("__weakref__",)


def arena_busy(a):
    if not a.in_flight:
        return False
    t = a.ticket
    if t is not None and t() is None:  # the autograd node that owned the saved activations is gone
        a.in_flight, a.ticket = False, None
        return False
    return True


def arena_claim(ctx, a):
    """Tie a training forward's arena to the life of its autograd node."""
    if a.in_flight:
        ctx.ticket = ArenaTicket()
        a.ticket = weakref.ref(ctx.ticket)


class StackArena(object):
    """Device buffers of the layer stack for one (B, S) shape, allocated ONCE and re-used every step (no allocator
    traffic on the hot path; sized for 288 GB of HBM: ~0.55 GB per layer at c2):

    * the stream between layers (fp32 + its (hi, lo) split) -- ``x_hi`` per layer, it is the X operand of the layer's
      weight gradients; fp32 and lo ping-pong;
    * the activations backward reads, per layer when gradients are needed (one shared set for inference);
    * backward buffers: those the weight-gradient stream reads (dt2, du16, dt1, dqkv, LayerNorm partials) are private
      to a layer, the rest of the critical path shares one set;
    * scratch of the weight-gradient stream: the K-major images of the eight GEMM operands, column-sum partials."""

    def __init__(self, L, B, S, H, I, nh, device, need_grad, images=True):
        self.key = (L, B, S, H, I, nh, str(device), need_grad)
        self.L, self.B, self.S, self.need_grad = L, B, S, need_grad
        M = B * S
        n = L if need_grad else 1
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)  # noqa: E731
        b16 = lambda *s: torch.empty(*s, dtype=BF16, device=device)  # noqa: E731
        self.x32, self.x_lo = f32(2, M, H), b16(2, M, H)
        self.x_hi = b16(n + 1, M, H)
        self.qkv_hi, self.qkv_lo = b16(n, M, 3 * H), b16(n, M, 3 * H)
        self.ctx_hi, self.ctx_lo, self.lse = b16(n, M, H), b16(n, M, H), f32(n, B * nh * S)
        self.z1, self.mean1, self.rstd1 = f32(n, M, H), f32(n, M), f32(n, M)
        self.x1_32, self.x1_hi, self.x1_lo = f32(M, H), b16(n, M, H), b16(M, H)
        self.u16, self.h_hi, self.h_lo = b16(n, M, I), b16(n, M, I), b16(M, I)
        self.z2, self.mean2, self.rstd2 = f32(n, M, H), f32(n, M), f32(n, M)
        self.addmask, self.row_post = f32(M), f32(M)
        self.rows0 = torch.arange(B, dtype=torch.int64, device=device) * S  # the pooled row of every sample
        self.in_flight = False  # a training forward whose backward has not run yet owns the saved activations
        self.ticket = None      # weakref to the ArenaTicket of that forward's autograd node (arena_busy)
        # workspace of the small-M GEMM path: the B-row products of the pooled last layer (and every product of a batch
        # with few rows); launches of one stream share it
        lib = _lib.lib()
        shapes = [(B, H, H), (B, I, H), (B, H, I)]
        if M <= 8192:
            shapes += [(M, 3 * H, H), (M, H, H), (M, I, H), (M, H, I), (M, H, 3 * H)]
        self.small_ws = f32(max(lib.vl_gemm_small_ws_floats(*s_) for s_ in shapes))
        if need_grad:
            self.dbuf = f32(2, M, H)
            self.dz2, self.dx1, self.dz1, self.dctx16 = f32(M, H), f32(M, H), f32(M, H), b16(M, H)
            self.dt2, self.du16, self.dt1, self.dqkv = b16(L, M, H), b16(L, M, I), b16(L, M, H), b16(L, M, 3 * H)
            nws = lib.vl_ln_bwd_ws_floats(M, H)
            self.lnws1, self.lnws2 = f32(L, nws), f32(L, nws)
            # K-major images of the eight GEMM operands: only when the weight-gradient GEMM does not read the row-major
            # activations directly (ragged row counts, A/B knobs)
            self.images = images
            if images:
                img = lambda N: b16(lib.vl_blocked_elems(M, N))  # noqa: E731
                self.t_dqkv, self.t_dt1, self.t_du, self.t_dt2 = img(3 * H), img(H), img(I), img(H)
                nx, ni = lib.vl_blocked_elems(M, H), lib.vl_blocked_elems(M, I)
                self.t_x, self.t_ctx, self.t_x1, self.t_h = b16(L, nx), b16(L, nx), b16(L, nx), b16(L, ni)
                # du's image + its column-sum partials per layer (VL_ST_FUSE_IMAGES): written by the GELU' epilogue on
                # the main stream while the side stream may still be reading the layer above's
                self.t_du_l, self.cs_du_l = b16(L, ni), f32(L, 8 * ((M + 255) // 256), I)
            # column-sum partials: one row per 64-row block (re-layout pass) or ceil(H / 256) rows per problem
            # (row-major dW GEMM: q | k | v at csq + {0, 1, 2} * tq * H, stack.hip) -- whichever is larger
            mb, tq = (M + 63) // 64, (H + 255) // 256
            self.cs_qkv, self.cs_u = f32(max(mb, 3 * tq), 3 * H), f32(max(mb, tq), I)
            self.fork = torch.cuda.Event()
            self.fork.record()  # materialises the hipEvent_t behind the handle
            self.dw_sk_ws = None  # workspace of the stream-K weight-gradient GEMM (LayerStack.dw_budget), made on demand

    def lay(self, t, l):
        return t[l if self.need_grad else 0]


class LayerStack(object):
    """N x { QKV GEMM -> attention -> out-proj GEMM -> dropout+residual+LN -> FFN1 GEMM (+GELU) -> FFN2 GEMM ->
    dropout+residual+LN [-> * row mask] } on the single [B*S, H] stream.  The sequencing itself is native
    (csrc/stack.hip: ONE call per direction); this class owns the buffers and the descriptor the native side walks."""

    def __init__(self, specs, H, nh, I, eps):
        self.specs, self.H, self.nh, self.I, self.eps = specs, H, nh, I, eps
        self.overlap_dw = True   # weight-gradient work on a second HIP stream (A/B knob)
        self.side_priority = None
        # operand layouts of the weight-gradient GEMM whenever B*S is a multiple of 64 (bit 0: dY row-major, bit 1: X
        # row-major -- transposing LDS reads instead of a K-major image written by a re-layout pass); 0 = both sides
        # through the re-layout pass (also the fallback for ragged row counts)
        self.dw_rowmajor = 3
        # tile of the narrow (N <= 1024) single-pass products of backward: 2 = 256 x 256 (168 workgroups at c2, one round on
        # the CUs the concurrent weight-gradient GEMM leaves free; -0.18 ms / step against the automatic 256 x 192); 0 = automatic
        self.dx_tile = 2
        # CU budget of the weight-gradient GEMMs: > 0 = stream-K form on that many workgroups (csrc/dw.hip), the dX products
        # of the main stream keep the remaining CUs (88 + 168 one-round dX products at c2); 0 = one workgroup per 256 x 256
        # tile (108 per layer at c2)
        self.dw_budget = 0
        # ... of the last weight-gradient launch of backward alone (layer 0: the main stream has nothing left to run beside
        # it, so it may take the whole chip: 216 workgroups = every tile cut in two)
        self.dw_tail_budget = 0
        # persistent form of the ping-pong GEMM (VL_GX_PERSIST): workgroup counts for (forward 3-pass, backward single-pass)
        # products; 0 = one workgroup per tile.  Measured at c2 (same box, profiles/r03_ab_log.txt): forward 16.09 -> 16.05
        # ms / step (bit-identical results), backward neutral to negative beside the weight-gradient stream
        self.gemm_persist = (256, 0)
        # the heads of this path read hidden_states[:, 0] only (BertTextPooler encoders.py:597-608, M3P BertPooler): the
        # last layer then runs on the B live rows after its K/V projection and the stack returns [B, 1, H] (exact: the
        # live rows are bit-identical to the dense run, the dead ones are never computed)
        self.pooled_only = True
        # K-major images written by GEMM epilogues instead of the re-layout pass: bit 0 = h (FFN1 forward), bit 1 = du +
        # its column sums (FFN1 backward).  Measured at c2 (same box): 0 -> 17.40, 1 -> 17.56, 2 -> 17.58, 3 -> 17.61 ms:
        # the image is 88 MB more HBM writes in an epilogue all CUs reach together (+17 us per FFN1 forward, +46 us per
        # backward), on the critical path, while the re-layout pass it saves runs beside the GEMMs on the side stream -> off
        self.fuse_images = 0
        self.tr_blocks = (0, 0)  # workgroup caps of the K-major re-layout launches (forward, backward); 0 = default
        # K-major X images: the bottom `tr_bwd_layers` layers' are written in backward on the side stream, the others at
        # the end of forward, in the window of the task head.  None = all but the top layer: since the head became one
        # native node its window (~0.3 ms) no longer hides 12 layers of re-layout (0.78 ms); measured at c2, same box:
        # 0 -> 17.28, 6 -> 17.11, 8 -> 17.04, 11 -> 16.98, 12 -> 17.02 ms / step
        self.tr_bwd_layers = None
        self.layer_ready_events = None  # per-layer torch.cuda.Event the forward waits for (pipelined optimizer update)
        self.layer_done_hook = None  # callable(layer, grads in LayerSpec.params order, stream) -> consumed?
        # callable(layer) -> (16 destination views in LayerSpec.params order, accumulate) or None: when the optimizer
        # provides it, gradients are written straight into its flat arena
        self.grad_sink = None
        self._side = None
        self._pending = None
        self._arenas = {}
        self._desc = {}
        self.prof = None  # numpy int64 VlProf block (bench.py): GEMM launch timing by caller-owned events

    def _new_side_stream(self, dev):
        # optimizer-only work must not starve the critical path of CUs: the side stream gets the LOWEST queue priority
        # (side_priority: None = lowest the runtime offers; an int = that priority)
        pr = self.side_priority
        if pr is None:
            pr = max(torch.cuda.Stream.priority_range())
        return torch.cuda.Stream(device=dev, priority=pr)

    def side_stream(self, dev):
        if self._side is None or self._side.device != dev:
            self._side = self._new_side_stream(dev)
        return self._side

    def make_prepared(self, device):
        return [dict(qkv=PreparedWeight([sp.q, sp.k, sp.v], device), o=PreparedWeight([sp.o], device),
                     w1=PreparedWeight([sp.w1], device), w2=PreparedWeight([sp.w2], device)) for sp in self.specs]

    # ---- buffers + descriptor -------------------------------------------------------------------------------------
    def arena(self, B, S, device, need_grad):
        # K-major operand images are only needed when the weight-gradient GEMM cannot read the row-major activations
        rowmajor = int(self.dw_rowmajor) == 3 and (B * S) % 64 == 0 and (not self.pooled_only or B % 64 == 0)
        key = (B, S, str(device), need_grad, rowmajor)
        lst = self._arenas.setdefault(key, [])
        for a in lst:
            if not arena_busy(a):
                return a
        a = StackArena(len(self.specs), B, S, self.H, self.I, self.nh, device, need_grad, images=not rowmajor)
        lst.append(a)
        if len(lst) > 4:
            raise RuntimeError("clg_vqa_amd: more than 4 training forwards without a backward on one model")
        return a

    def _descriptor(self, ar, pw_layers):
        """numpy int64 descriptor of (arena, prepared weights, parameters, masks): rebuilt only when a pointer moved."""
        ptrs = []
        for sp, lw in zip(self.specs, pw_layers):
            for k in ("qkv", "o", "w1", "w2"):
                p = lw[k]
                ptrs += [p.hi.data_ptr(), p.lo.data_ptr(), p.t_hi.data_ptr(), p.bias.data_ptr()]
            ptrs += [sp.ln1.weight.data_ptr(), sp.ln1.bias.data_ptr(), sp.ln2.weight.data_ptr(), sp.ln2.bias.data_ptr()]
            for lin in (sp.q, sp.k, sp.v, sp.o, sp.w1, sp.w2):
                m = linear_params(lin)[1]
                ptrs.append(0 if m is None else m.data_ptr())
        fp = (id(ar), tuple(ptrs))
        hit = self._desc.get(id(ar))
        if hit is not None and hit[0] == fp:
            return hit[1]
        L = len(self.specs)
        F, LF = VL["VL_ST_FIELDS"], VL["VL_LY_FIELDS"]
        d = np.zeros(F + L * LF, dtype=np.int64)
        d[VL["VL_ST_MAGIC"]] = VL["VL_ST_MAGIC_VALUE"]
        for k, v in (("B", ar.B), ("S", ar.S), ("H", self.H), ("I", self.I), ("NH", self.nh), ("NLAYERS", L)):
            d[VL["VL_ST_" + k]] = v
        d[VL["VL_ST_EPS"]] = _f32_bits(self.eps)
        d[VL["VL_ST_ADDMASK"]] = ar.addmask.data_ptr()
        d[VL["VL_ST_ROWS0"]] = ar.rows0.data_ptr()
        if ops.SMALL_GEMM:
            d[VL["VL_ST_SMALL_WS"]], d[VL["VL_ST_SMALL_WS_FLOATS"]] = ar.small_ws.data_ptr(), ar.small_ws.numel()
        if ar.need_grad:
            d[VL["VL_ST_EV_FORK"]] = ar.fork.cuda_event
            for k in ("t_dqkv", "t_dt1", "t_du", "t_dt2", "cs_qkv", "cs_u"):
                if ar.images or k.startswith("cs_"):
                    d[VL["VL_ST_" + k.upper()]] = getattr(ar, k).data_ptr()
        it = iter(ptrs)
        for l in range(L):
            y = F + l * LF

            def put(name, t):
                d[y + VL["VL_LY_" + name]] = t if isinstance(t, int) else t.data_ptr()
            put("X32", ar.x32[l % 2]); put("X_LO", ar.x_lo[l % 2])
            put("X_HI", ar.x_hi[l] if ar.need_grad else ar.x_hi[l % 2])  # per layer when backward needs it, else ping-pong
            put("OUT32", ar.x32[(l + 1) % 2]); put("OUT_LO", ar.x_lo[(l + 1) % 2])
            put("OUT_HI", ar.x_hi[l + 1] if ar.need_grad else ar.x_hi[(l + 1) % 2])
            for k in ("QKV", "O", "W1", "W2"):
                for suffix in ("_HI", "_LO", "_T"):
                    put("W" + k + suffix if k in ("QKV", "O") else k + suffix, next(it))
                put("B" + (k if k in ("QKV", "O") else k[1:]), next(it))
            for name in ("LN1_G", "LN1_B", "LN2_G", "LN2_B"):
                put(name, next(it))
            for i in range(6):
                d[y + VL["VL_LY_MASK0"] + i] = next(it)
            for name in ("qkv_hi", "qkv_lo", "ctx_hi", "ctx_lo", "lse", "z1", "mean1", "rstd1", "x1_hi", "u16", "h_hi", "z2",
                         "mean2", "rstd2"):
                put(name.upper(), ar.lay(getattr(ar, name), l))
            put("X1_32", ar.x1_32); put("X1_LO", ar.x1_lo); put("H_LO", ar.h_lo)
            if ar.need_grad:
                put("DX", ar.dbuf[l % 2])
                put("DY", ar.dbuf[(l + 1) % 2])  # (the top layer's DY is patched per backward)
                put("DZ2", ar.dz2); put("DX1", ar.dx1); put("DZ1", ar.dz1); put("DCTX16", ar.dctx16)
                for name in ("dt2", "du16", "dt1", "dqkv", "lnws1", "lnws2"):
                    put(name.upper(), getattr(ar, name)[l])
                if ar.images:
                    for name in ("t_x", "t_ctx", "t_x1", "t_h"):
                        put(name.upper(), getattr(ar, name)[l])
                    put("T_DU", ar.t_du_l[l]); put("CS_DU", ar.cs_du_l[l])
        self._desc[id(ar)] = (fp, d)
        return d

    # ---- forward / backward ----------------------------------------------------------------------------------------
    def input_buffers(self, ar):
        """Where the embeddings write the stack's input: (x32, x_hi, x_lo) of layer 0."""
        return ar.x32[0], ar.x_hi[0], ar.x_lo[0]

    def forward(self, ar, pw_layers, p_hid, p_att, seed0, row_post=None):
        d = self._descriptor(ar, pw_layers)
        d[VL["VL_ST_P_HID"]], d[VL["VL_ST_P_ATT"]] = _f32_bits(p_hid), _f32_bits(p_att)
        d[VL["VL_ST_SEED0"]] = seed0
        d[VL["VL_ST_ROW_POST"]] = 0 if row_post is None else ar.row_post.data_ptr()
        d[VL["VL_ST_PROF"]] = 0 if self.prof is None else self.prof.ctypes.data
        d[VL["VL_ST_POOLED_ONLY"]] = 1 if self.pooled_only else 0
        d[VL["VL_ST_TR_BLOCKS_FWD"]], d[VL["VL_ST_TR_BLOCKS_BWD"]] = self.tr_blocks
        d[VL["VL_ST_TR_BWD_LAYERS"]] = len(self.specs) - 1 if self.tr_bwd_layers is None else self.tr_bwd_layers
        d[VL["VL_ST_FUSE_IMAGES"]] = self.fuse_images
        d[VL["VL_ST_DW_ROWMAJOR"]] = int(self.dw_rowmajor)
        d[VL["VL_ST_DX_TILE"]] = int(self.dx_tile)
        d[VL["VL_ST_GEMM_PERSIST"]] = int(self.gemm_persist[0]) | (int(self.gemm_persist[1]) << 16)
        evs = self.layer_ready_events  # set by the engine when an optimizer update is running under this forward
        F_, LF_ = VL["VL_ST_FIELDS"], VL["VL_LY_FIELDS"]
        for l in range(len(self.specs)):
            d[F_ + l * LF_ + VL["VL_LY_EV_READY"]] = 0 if evs is None else evs[l].cuda_event
        sk = max(int(self.dw_budget), int(self.dw_tail_budget))
        if ar.need_grad and sk > 0:
            if ar.dw_sk_ws is None or ar.dw_sk_budget != sk:
                ar.dw_sk_ws, ar.dw_sk_budget = ops.dw_streamk_ws(sk, ar.x32.device), sk
            d[VL["VL_ST_DW_BUDGET"]] = int(self.dw_budget)
            d[VL["VL_ST_DW_TAIL_BUDGET"]] = int(self.dw_tail_budget)
            d[VL["VL_ST_DW_SK_WS"]] = ar.dw_sk_ws.data_ptr()
            d[VL["VL_ST_DW_SK_WS_BYTES"]] = ar.dw_sk_ws.numel()
        else:
            d[VL["VL_ST_DW_BUDGET"]] = d[VL["VL_ST_DW_TAIL_BUDGET"]] = 0
        side_ptr = None
        if ar.need_grad and self.overlap_dw:
            dev = ar.x32.device
            if self._side is None or self._side.device != dev:
                self._side = self._new_side_stream(dev)
            # (a forward whose backward never ran may still have its re-layout in flight on the side stream)
            torch.cuda.current_stream().wait_stream(self._side)
            side_ptr = self._side.cuda_stream
        ops.stack_fwd(d, 0, len(self.specs), side_ptr)
        ar.desc = d  # backward re-uses it (same pointers: nothing can move between a forward and its backward)
        L = len(self.specs)
        if ar.need_grad:
            ar.in_flight = True
        ar.pooled = self.pooled_only
        out = ar.x32[L % 2]
        H = self.H
        return out[:ar.B].view(ar.B, 1, H) if self.pooled_only else out.view(ar.B, ar.S, H)

    def backward(self, ar, pw_layers, dy, p_hid, p_att, seed0, row_post=None):
        """dy [M,H] fp32 = dL/d(stack output).  Returns (dL/d(stack input), per-layer grads in LayerSpec.params order:
        None where the gradient went straight into the optimizer's arena)."""
        L, dev = len(self.specs), dy.device
        d = ar.desc
        d[VL["VL_ST_P_HID"]], d[VL["VL_ST_P_ATT"]] = _f32_bits(p_hid), _f32_bits(p_att)
        d[VL["VL_ST_SEED0"]] = seed0
        d[VL["VL_ST_ROW_POST"]] = 0 if row_post is None else ar.row_post.data_ptr()
        d[VL["VL_ST_PROF"]] = 0 if self.prof is None else self.prof.ctypes.data
        d[VL["VL_ST_POOLED_ONLY"]] = 1 if ar.pooled else 0
        assert dy.numel() == (ar.B if ar.pooled else ar.B * ar.S) * self.H
        F, LF = VL["VL_ST_FIELDS"], VL["VL_LY_FIELDS"]
        d[F + (L - 1) * LF + VL["VL_LY_DY"]] = dy.data_ptr()
        layer_grads, accumulate = [], None
        use_sink = self.grad_sink is not None and self.layer_done_hook is not None
        for l, sp in enumerate(self.specs):
            sink = self.grad_sink(l) if use_sink else None
            if sink is not None:
                views, acc = sink
                layer_grads.append([None] * 16)
            else:
                acc = False
                sizes = [p.numel() for p in sp.params()]
                flat = torch.empty(sum((n + 3) // 4 * 4 for n in sizes), dtype=torch.float32, device=dev)
                views, off = [], 0
                for p_, n in zip(sp.params(), sizes):
                    views.append(flat[off:off + n].view_as(p_))
                    off += (n + 3) // 4 * 4
                layer_grads.append(list(views))
            if accumulate is None:
                accumulate = acc
            elif accumulate != acc:
                raise RuntimeError("clg_vqa_amd: layers disagree on gradient accumulation")
            g0 = F + l * LF + VL["VL_LY_GRAD0"]
            for i, v in enumerate(views):
                d[g0 + i] = v.data_ptr()
        d[VL["VL_ST_ACCUMULATE"]] = 1 if accumulate else 0
        main = torch.cuda.current_stream()
        side = None
        if self.overlap_dw:
            if self._side is None or self._side.device != dev:
                self._side = self._new_side_stream(dev)
            side = self._side
        side_ptr = side.cuda_stream if side is not None else None
        if self.layer_done_hook is None or not use_sink:
            ops.stack_bwd(d, L, 0, main.cuda_stream, side_ptr)
        else:
            # multi-GPU: after each layer's weight-gradient work is enqueued, the optimizer launches that layer's
            # all-reduce behind it on the same stream, overlapping the exchange with the rest of backward
            for l in reversed(range(L)):
                ops.stack_bwd(d, l + 1, l, main.cuda_stream, side_ptr)
                self.layer_done_hook(l, layer_grads[l], side if side is not None else main)
        self._pending = side
        ar.in_flight = False
        self._keep = layer_grads  # engine-owned gradient buffers are written by the side stream until the join
        return ar.dbuf[0], layer_grads

    def join(self):
        """Make the current stream wait for the weight-gradient stream."""
        if self._pending is not None:
            torch.cuda.current_stream().wait_stream(self._pending)
            self._pending = None
        self._keep = None


class EngineBase(object):
    """Weight preparation (one launch per step), dropout seeds, dirty tracking shared by the UC2 and M3P engines."""

    def _init_common(self, H, nh):
        if H % nh != 0 or H // nh not in (32, 64):
            raise ValueError("clg_vqa_amd: the native attention kernel needs head dim 64 or 32 (hidden %d / heads %d)" % (H, nh))
        if H != 128 and H % 256 != 0:
            raise ValueError("clg_vqa_amd: hidden size must be 128 or a multiple of 256 for the native LayerNorm")
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
        self.grad_mode = True  # torch.is_grad_enabled() at the model's call site (set by the module's forward)
        self._seed0 = 0
        self._last_arena = None
        # optimizer update under the next forward (FusedAdamW.pipeline_update): the update + weight preparation of chunk c
        # (0 = embeddings, 1 + l = layer l, last = task heads) runs on `update_stream`; the forward waits for chunk_events[c]
        self.supports_update_pipeline = False
        self._update_stream = None
        self.chunk_events = None
        self.chunks_pending = False
        self._skip_prep_once = False

    def mark_dirty(self):
        """Call after updating parameters through raw pointers (the fused optimizer does)."""
        self._dirty = True

    # ---- optimizer update under the next forward ---------------------------------------------------------------------
    def update_stream(self, dev):
        if self._update_stream is None or self._update_stream.device != dev:
            self._update_stream = torch.cuda.Stream(device=dev)
        return self._update_stream

    def n_chunks(self):
        return len(self.stack.specs) + 2

    def chunk_tables(self):
        """Per-chunk weight-preparation tables (built with the one-launch table by prepared()), or None before that."""
        return None if self._prepared is None else self._prepared.get("chunks")

    def prepare_chunk(self, c):
        """Weight preparation + packed-bias copies of chunk c on the CURRENT native / torch stream."""
        ch = self._prepared["chunks"][c]
        if ch["nrows"]:
            ops.weight_prep_multi(ch["table"], ch["nrows"], ch["tiles"])
        if ch["bias_dst"]:
            torch._foreach_copy_(ch["bias_dst"], ch["bias_src"])

    def finish_prepare(self):
        """All chunks were prepared by the optimizer: the next forward only waits for their events."""
        pw = self._prepared
        for p_ in pw["all_pw"]:
            p_.refresh_bias(fingerprint=False, pack=False)
        self._dirty = False
        self._skip_prep_once = True
        self.chunks_pending = True

    def wait_chunk(self, c):
        if self.chunks_pending and self.chunk_events is not None:
            torch.cuda.current_stream().wait_event(self.chunk_events[c])

    def _push_word_grad(self, ids, rows, pad_id):
        rows = rows * (ids != pad_id).to(rows.dtype).unsqueeze(1)  # the pad row receives no gradient
        if self.pending_word_grad is None:
            self.pending_word_grad = (ids, rows)
        else:  # gradient accumulation over micro-batches
            self.pending_word_grad = (torch.cat([self.pending_word_grad[0], ids]),
                                      torch.cat([self.pending_word_grad[1], rows]))

    def next_seed(self):
        """(seed0, site -> seed): site s draws from seed0 * 4096 + s (the native stack uses sites 16 l + 3 .. 16 l + 5)."""
        self.calls += 1
        seed0 = (self.base_seed * 0x9E3779B1 + self.calls * 0x10001) & 0x7FFFFFFFFFFF
        self._seed0 = seed0
        return seed0, (lambda site: (seed0 * 4096 + site) & 0xFFFFFFFFFFFFFFFF)

    def last_seed(self, site):
        """Seed of `site` for the forward pass that is being built (the task head draws its dropout mask from it)."""
        return (self._seed0 * 4096 + site) & 0xFFFFFFFFFFFFFFFF

    def head_linears(self):
        """Leaf Linears of the task heads: prepared by the same launch as the trunk's weights."""
        return []

    def pooled_split(self, x):
        """(hi, lo) bf16 operand form of the pooled rows when `x` is the output of this engine's last forward in the
        pooled-row mode (the last LayerNorm of the stack wrote it next to the fp32 rows), else None."""
        ar = self._last_arena
        if ar is None or not getattr(ar, "pooled", False):
            return None
        L = ar.L
        out = ar.x32[L % 2]
        if x.data_ptr() != out.data_ptr() or x.shape[0] != ar.B or x.numel() != ar.B * out.shape[-1]:
            return None
        hi = (ar.x_hi[L] if ar.need_grad else ar.x_hi[L % 2])[:ar.B]
        return hi, ar.x_lo[L % 2][:ar.B]

    def prepared(self, device):
        if self._prepared is None or self._prepared["device"] != device:
            heads = self.head_linears()
            self._prepared = dict(device=device, img=PreparedWeight([self.image_linear()], device, need_t=False),
                                  layers=self.stack.make_prepared(device),
                                  head=[PreparedWeight([lin], device) for lin in heads])
            for lin, p in zip(heads, self._prepared["head"]):
                object.__setattr__(lin, "_vl_engine_pw", p)  # VLLinear._vl_prepared returns it
            self._dirty = True
        pw = self._prepared
        all_pw = [pw["img"]] + [lw[k] for lw in pw["layers"] for k in ("qkv", "o", "w1", "w2")] + pw["head"]
        pw["all_pw"] = all_pw
        if self._skip_prep_once and not self._dirty:
            # the optimizer prepared every chunk on the update stream (finish_prepare): nothing to launch, nothing to
            # fingerprint; the forward waits for the chunk events
            self._skip_prep_once = False
            return pw
        explicit = self._dirty  # set by the optimizer every step: no need to fingerprint versions to find that out
        if explicit or any(p._key() != p.key for p in all_pw):
            # the device table only depends on where the source weights / masks live: compare those pointers first
            ident = tuple(q for p in all_pw for l in p.linears for q in _src_ptrs(l) + (l.bias.data_ptr(),))
            if pw.get("table_ident") != ident:
                rows, tile0 = [], 0
                for r in (r for p in all_pw for r in p.descriptors()):
                    rows.append(r + [tile0])
                    tile0 += ((r[5] + 63) // 64) * ((r[6] + 63) // 64)
                pw["table"] = torch.tensor(rows, dtype=torch.int64).to(device)
                pw["table_ident"], pw["table_tiles"] = ident, tile0
                # the packed [Q|K|V] bias copies of all layers: (destination views, source biases) built once per
                # parameter placement -- ~150 tensor slices / detaches per step otherwise, on the host's critical path
                # at the step boundary
                pairs = [pr for p in all_pw for pr in p.bias_pairs()]
                pw["bias_dst"], pw["bias_src"] = [d for d, _ in pairs], [s_ for _, s_ in pairs]
                # the same table cut into chunks (embeddings | layer l | heads) for the optimizer's pipelined update
                groups = [[pw["img"]]] + [[lw[k] for k in ("qkv", "o", "w1", "w2")] for lw in pw["layers"]] + [pw["head"]]
                chunks = []
                for grp in groups:
                    rws, t0 = [], 0
                    for r in (r for p in grp for r in p.descriptors()):
                        rws.append(r + [t0])
                        t0 += ((r[5] + 63) // 64) * ((r[6] + 63) // 64)
                    prs = [pr for p in grp for pr in p.bias_pairs()]
                    chunks.append(dict(table=torch.tensor(rws, dtype=torch.int64).to(device) if rws else None, nrows=len(rws),
                                       tiles=t0, bias_dst=[d for d, _ in prs], bias_src=[s_ for _, s_ in prs]))
                pw["chunks"] = chunks
            ops.weight_prep_multi(pw["table"], pw["table"].shape[0], pw["table_tiles"])  # one launch per step
            if pw["bias_dst"]:  # one multi-tensor copy
                torch._foreach_copy_(pw["bias_dst"], pw["bias_src"])
            for p in all_pw:
                p.refresh_bias(fingerprint=not explicit, pack=False)
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
        self.embed_overlap = True  # token / box-location embeddings on the side stream beside the feature projection
        self.supports_update_pipeline = True
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

    def head_linears(self):
        lins = [self.model.bert.t_pooler.dense]
        for clf in self.model.clfs_dict.values():
            lins += [clf.logit_fc[0], clf.logit_fc[3]]
        return lins

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
    def forward(self, ids, feats, locs, seg, tmask, imask, training, need_grad=True):
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
        seed0, seed = self.next_seed()
        pw = self.prepared(dev)
        # (an optimizer update may still be running on the update stream: embeddings wait for chunk 0, every layer of the
        # stack for its own chunk, the task head for the last one)
        self.wait_chunk(0)
        self.stack.layer_ready_events = self.chunk_events[1:-1] if self.chunks_pending and self.chunk_events else None
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        b16 = lambda *s: torch.empty(*s, dtype=BF16, device=dev)  # noqa: E731
        ids = ids.contiguous(); seg = seg.contiguous()
        tmask = tmask.contiguous().to(torch.int64); imask = imask.contiguous().to(torch.int64)
        feats2 = feats.contiguous().view(BV, F)
        locs2 = locs.contiguous().view(BV, L)
        ar = self.stack.arena(B, S, dev, need_grad)
        self._last_arena = ar
        sv = dict(B=B, T=T, V=V, F=F, L=L, S=S, p_hid=p_hid, p_att=p_att, seed=seed, seed0=seed0, ids=ids, seg=seg,
                  locs=locs2, pw=pw, arena=ar)

        # the embeddings are ~10 small launches: the token rows and the box-location rows (no GEMM) run on the side stream
        # beside the region-feature projection (split + GEMM + LayerNorm) of the main stream
        main = torch.cuda.current_stream()
        side = self.stack.side_stream(dev) if self.embed_overlap else None
        x32, x_hi, x_lo = self.stack.input_buffers(ar)
        type_w = emb.new_token_type_embeddings.weight.detach()
        z_t, mean_t, rstd_t = f32(BT, H), f32(BT), f32(BT)
        z_l, mean_l, rstd_l, b32 = f32(BV, H), f32(BV), f32(BV), f32(BV, H)
        if side is not None:
            side.wait_stream(main)  # parameters (optimizer), batch tensors, buffers of the previous step
            ops.set_stream(side.cuda_stream)
        am = ar.addmask
        ops.addmask(tmask, imask, am, B, T, V)
        # text rows: word + position + type -> LN -> dropout   (embeddings.py:648-655)
        ops.embed_text_fwd(ids, seg, emb.word_embeddings.weight.detach(), emb.position_embeddings.weight.detach(),
                           type_w, z_t, B, T, H, int(cfg.pad_token_id))
        ops.ln_fwd(z_t, None, None, emb.LayerNorm.weight.detach(), emb.LayerNorm.bias.detach(), self.eps, x32, x_hi,
                   x_lo, mean_t, rstd_t, BT, H, group=T, out_stride=S, out_off=0, p_post=p_hid, seed=seed(1))
        # box locations: LN(loc W^T + b)   (embeddings.py:661-664)
        ops.loc_linear_fwd(locs2, emb.image_location_embeddings.weight.detach(),
                           emb.image_location_embeddings.bias.detach(), z_l, BV, L, H)
        ops.ln_fwd(z_l, None, None, emb.image_location_layer_norm.weight.detach(),
                   emb.image_location_layer_norm.bias.detach(), self.eps, b32, None, None, mean_l, rstd_l, BV, H)
        if side is not None:
            ops.set_stream(main.cuda_stream)
        # box rows: LN(feat W^T + b) + LN(loc ...) + type[1] -> LN -> dropout   (embeddings.py:660-667)
        f_hi, f_lo = b16(BV, F), b16(BV, F)
        ops.split_f32(feats2, f_hi, f_lo)
        z_i, mean_i, rstd_i, a32 = f32(BV, H), f32(BV), f32(BV), f32(BV, H)
        ops.gemm_nt(f_hi, f_lo, pw["img"].hi, pw["img"].lo, BV, H, F, 3, EPI_F32, bias=pw["img"].bias, out32=z_i)
        ops.ln_fwd(z_i, None, None, emb.image_layer_norm.weight.detach(), emb.image_layer_norm.bias.detach(),
                   self.eps, a32, None, None, mean_i, rstd_i, BV, H)
        if side is not None:
            main.wait_stream(side)
        mean_v, rstd_v = f32(BV), f32(BV)
        ops.ln_fwd(a32, b32, type_w[1], emb.v_LayerNorm.weight.detach(), emb.v_LayerNorm.bias.detach(), self.eps,
                   x32, x_hi, x_lo, mean_v, rstd_v, BV, H, group=V, out_stride=S, out_off=T, p_post=p_hid,
                   seed=seed(2))
        sv.update(z_t=z_t, mean_t=mean_t, rstd_t=rstd_t, f_hi=f_hi, z_i=z_i, mean_i=mean_i, rstd_i=rstd_i,
                  z_l=z_l, mean_l=mean_l, rstd_l=rstd_l, z_v=a32, mean_v=mean_v, rstd_v=rstd_v)
        return self.stack.forward(ar, pw["layers"], p_hid, p_att, seed0), sv

    # ---- backward ------------------------------------------------------------------------------------------------
    def backward(self, sv, dx):
        """dx: [B,S,H] fp32 gradient of the trunk output.  Returns grads in ``param_list`` order."""
        cfg = self.model.config
        emb = self.model.bert.embeddings
        B, T, V, F, L, S = sv["B"], sv["T"], sv["V"], sv["F"], sv["L"], sv["S"]
        H = self.H
        M, BT, BV = B * S, B * T, B * V
        dev = dx.device
        p_hid, p_att, seed, pw = sv["p_hid"], sv["p_att"], sv["seed"], sv["pw"]
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
        b16 = lambda *s: torch.empty(*s, dtype=BF16, device=dev)  # noqa: E731
        ws = ops.ln_bwd_ws(M, H, dev)
        dy, layer_grads = self.stack.backward(sv["arena"], pw["layers"], dx.contiguous().view(-1, H), p_hid, p_att,
                                              sv["seed0"])

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
        dWimg = dw_gemm(dimg16, sv["f_hi"], BV, H, F, mask=linear_params(emb.image_embeddings)[1])
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
        # (fixed summation order: sort + run sums, csrc/scatter.hip; beyond 16384 text rows the atomic kernel)
        text_bwd = ops.embed_text_bwd_det if ops.DETERMINISTIC_EMBED_BWD and BT <= 16384 else ops.embed_text_bwd
        text_bwd(sv["ids"], sv["seg"], dz_t, dword, dpos, dtype_, B, T, H, int(cfg.pad_token_id),
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
    """autograd boundary of a native trunk: (batch, *params) -> X_final [B,S,H], or its pooled rows [B,1,H] when the
    stack runs in the pooled-row mode (the heads only ever read X_final[:, 0])."""

    @staticmethod
    def forward(ctx, engine, training, ids, feats, locs, seg, tmask, imask, *params):
        ops.set_stream(torch.cuda.current_stream().cuda_stream)  # one stream lookup for all launches of the pass
        try:
            # whether a backward can follow: the caller's grad mode (inside autograd.Function.forward it is always off,
            # and ctx.needs_input_grad ignores torch.no_grad()) and whether any parameter wants a gradient
            out, sv = engine.forward(ids, feats, locs, seg, tmask, imask, training,
                                     need_grad=engine.grad_mode and any(ctx.needs_input_grad))
        finally:
            ops.set_stream(None)
        ctx.engine = engine
        ctx.sv = sv
        arena_claim(ctx, sv["arena"])
        return out

    @staticmethod
    def backward(ctx, dx):
        if ctx.sv is None:
            raise RuntimeError("clg_vqa_amd: the trunk's saved activations were released by its first backward "
                               "(retain_graph / double backward are not supported by the native trunk)")
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

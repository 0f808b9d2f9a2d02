"""Thin tensor-level wrappers over the C ABI (include/vlhip.h).  Tensors are torch CUDA(=HIP) tensors used
purely as device-memory handles; every op enqueues on torch's current stream.  CPU tensors are rejected."""
import ctypes
import threading

import torch

from . import _lib

BF16 = torch.bfloat16
EPI_F32, EPI_GELU_SPLIT, EPI_DGELU_BF16, EPI_BF16, EPI_SPLIT = 0, 1, 2, 3, 4
GEMM_TILE = 0  # default `tile` of gemm_nt (micro-benchmarks set it; the native stack always uses the automatic choice)


class _Launch(threading.local):
    """Per-thread launch context.  `stream`: explicit HIP stream handle for every native launch (the engine sets it
    once per forward / backward: torch.cuda.current_stream() costs ~9 us per call, i.e. ~1 ms per step); `hold`: while
    launches go to a stream other than torch's current one, temporaries allocated inside the wrappers are appended
    here so that the caching allocator cannot hand their memory out again before the streams are joined."""
    stream = None
    hold = None


_launch = _Launch()


def set_stream(handle, hold=None):
    """Route the native launches of this thread to `handle` (None = torch's current stream again)."""
    _launch.stream, _launch.hold = handle, hold


def _stream():
    s = _launch.stream
    return s if s is not None else torch.cuda.current_stream().cuda_stream


def _tmp(t):
    if _launch.hold is not None:
        _launch.hold.append(t)
    return t


def _p(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("clg_vqa_amd: native ops need device (cuda/HIP) tensors; got a CPU tensor -- the product "
                           "path has no CPU fallback")
    if not t.is_contiguous():
        raise RuntimeError("clg_vqa_amd: native ops need contiguous tensors")
    return t.data_ptr()


def _pld(t):
    """pointer + leading dimension of a 2-D row-major tensor whose rows may be strided (a column slice)."""
    if t is None:
        return None, 0
    if not t.is_cuda:
        raise RuntimeError("clg_vqa_amd: native ops need device tensors (no CPU fallback)")
    assert t.dim() == 2 and t.stride(1) == 1, "need a row-major 2-D tensor"
    return t.data_ptr(), t.stride(0)


_SMALL_WS = {}
SMALL_GEMM = True  # A/B knob (bench.py BENCH_NO_SMALL_GEMM): False = batch-sized products on the big-tile kernels


def small_ws(device, floats, stream):
    """Grow-only fp32 workspace of the small-M GEMM path, one per (device, launch stream): launches on one stream are
    ordered, so consecutive products may share it."""
    key = (str(device), stream)
    t = _SMALL_WS.get(key)
    if t is None or t.numel() < floats:
        t = torch.empty(max(floats, 1 << 20), dtype=torch.float32, device=device)
        _SMALL_WS[key] = t
    return t


def gemm_nt(a_hi, a_lo, b_hi, b_lo, M, N, K, passes, epilogue, bias=None, resid=None, out32=None, out_hi=None,
            out_lo=None, aux16=None, tile=0, image=None, image_cols=0, colsum=None, persist=0):
    """C[M,N] = A[M,K] . B[N,K]^T (+epilogue); operands are bf16 2-D tensors (row-major, ld = stride(0)).  Products
    with a small M (the batch-sized products of the head) get a workspace and take the library's small-M path.
    image / image_cols: also store out_hi as the K-major image of the weight-gradient GEMM; colsum [rows, N] fp32:
    column-sum partials of out_hi (DGELU epilogue, ping-pong kernel) -- returns the number of partial rows written."""
    import ctypes
    pa, lda = _pld(a_hi)
    pb, ldb = _pld(b_hi)
    pal = _pld(a_lo)[0] if a_lo is not None else None
    pbl = _pld(b_lo)[0] if b_lo is not None else None
    po32, ldc = _pld(out32)
    ph, ld16 = _pld(out_hi)
    pl = _pld(out_lo)[0] if out_lo is not None else None
    px = _pld(aux16)[0] if aux16 is not None else None
    if resid is not None:
        assert resid.stride(0) == ldc
    L = _lib.lib()
    tile = tile or GEMM_TILE
    st = _stream()
    extra = (ctypes.c_int64 * 8)()
    extra[0] = tile
    extra[7] = persist  # > 0: persistent form of the ping-pong kernel on that many workgroups (VL_GX_PERSIST)
    if M <= 8192 and tile in (0, 8) and colsum is None and (SMALL_GEMM or tile == 8):
        nws = L.vl_gemm_small_ws_floats(M, N, K)
        extra[1], extra[2] = small_ws(a_hi.device, nws, st).data_ptr(), nws
    if image is not None:
        extra[3], extra[4] = _p(image), image_cols or N
    if colsum is not None:
        assert colsum.shape[1] == N
        extra[5] = _p(colsum)
    _lib.check(L.vl_gemm_nt_ex(pa, pal, lda, pb, pbl, ldb, M, N, K, passes, epilogue, _p(bias),
                               _pld(resid)[0] if resid is not None else None, po32, ldc, ph, pl, px, ld16,
                               ctypes.cast(extra, ctypes.c_void_p), st), "vl_gemm_nt")
    return int(extra[6])


def gemm_nt_splitk(a_hi, b_hi, M, N, K, out32, splits=None):
    """out32[M,N] = A[M,K] . B[N,K]^T with the K range split over enough workgroups to fill the 256 CUs."""
    L = _lib.lib()
    if splits is None:
        splits = L.vl_gemm_splitk_plan(M, N, K)
    ws = None
    if splits > 1:
        ws = _tmp(torch.empty(L.vl_gemm_splitk_ws_floats(M, N, splits), dtype=torch.float32, device=out32.device))
    pa, lda = _pld(a_hi)
    pb, ldb = _pld(b_hi)
    assert out32.is_contiguous() and out32.shape[-1] == N
    _lib.check(L.vl_gemm_nt_splitk(pa, lda, pb, ldb, M, N, K, splits, _p(ws), _p(out32), _stream()),
               "vl_gemm_nt_splitk")


def gemm_tn_splitk(a, b, M, N, K, out32, splits=None):
    """out32[M,N] = A[K,M]^T . B[K,N] (row-major bf16 activations).  Returns False when the shape is outside the TN
    fast path (the caller then transposes and uses gemm_nt_splitk)."""
    L = _lib.lib()
    if splits is None:
        splits = L.vl_gemm_splitk_plan(M, N, K)
    ws = None
    if splits > 1:
        ws = _tmp(torch.empty(L.vl_gemm_splitk_ws_floats(M, N, splits), dtype=torch.float32, device=out32.device))
    pa, lda = _pld(a)
    pb, ldb = _pld(b)
    assert out32.is_contiguous() and out32.shape[-1] == N
    rc = L.vl_gemm_tn_splitk(pa, lda, pb, ldb, M, N, K, splits, _p(ws), _p(out32), _stream())
    if rc == -2:
        return False
    _lib.check(rc, "vl_gemm_tn_splitk")
    return True


def stack_fwd(desc, layer_begin, layer_end, stream_side=None):
    """desc: numpy int64 descriptor (engine.LayerStack._descriptor)."""
    _lib.check(_lib.lib().vl_stack_fwd(desc.ctypes.data, layer_begin, layer_end, _stream(), stream_side), "vl_stack_fwd")


def stack_bwd(desc, layer_hi, layer_lo, stream_main, stream_side):
    _lib.check(_lib.lib().vl_stack_bwd(desc.ctypes.data, layer_hi, layer_lo, stream_main, stream_side), "vl_stack_bwd")


def transpose_blocked(entries, M, max_blocks=0):
    """entries = [(src [M,N] bf16 (row-major, ld = stride(0)), dst blocked image, colsum_partial or None), ...]"""
    import ctypes
    n = len(entries)
    arr = (ctypes.c_int64 * (6 * n))()
    for i, (src, dst, cs) in enumerate(entries):
        ps, ld = _pld(src)
        arr[6 * i:6 * i + 6] = [ps, ld, src.shape[1], _p(dst), 0 if cs is None else _p(cs), 0]
    _lib.check(_lib.lib().vl_transpose_blocked(ctypes.cast(arr, ctypes.c_void_p), n, M, max_blocks, _stream()),
               "vl_transpose_blocked")


def colsum_finalize(partial, nblk, N, outs, accumulate=False):
    import ctypes
    arr = (ctypes.c_void_p * len(outs))(*[_p(o) for o in outs])
    _lib.check(_lib.lib().vl_colsum_finalize(_p(partial), nblk, N, ctypes.cast(arr, ctypes.c_void_p), len(outs),
                                             1 if accumulate else 0, _stream()), "vl_colsum_finalize")


def colreduce_multi(sets, accumulate=False):
    """sets = [(src [nrows, ncols] fp32, seg, (out0, out1, out2) with None = skip), ...] (<= 4): one launch."""
    import ctypes
    n = len(sets)
    arr = (ctypes.c_int64 * (8 * n))()
    for i, (src, seg, outs) in enumerate(sets):
        o = [0 if t is None else _p(t) for t in outs] + [0] * (3 - len(outs))
        arr[8 * i:8 * i + 8] = [_p(src), src.shape[0], src.shape[1], seg] + o + [0]
    _lib.check(_lib.lib().vl_colreduce_multi(ctypes.cast(arr, ctypes.c_void_p), n, 1 if accumulate else 0, _stream()),
               "vl_colreduce_multi")


def dw_grouped(problems, K, accumulate=False):
    """problems = [(aT_ptr_tensor, a_row0, a_rows_total, bT_tensor, b_rows_total, out [M,N] fp32, mask or None, M, N)]:
    aT / bT are blocked images (transpose_blocked); a_row0 selects a row sub-range of the A image."""
    import ctypes
    n = len(problems)
    arr = (ctypes.c_int64 * (10 * n))()
    for i, (aT, a_row0, a_rows, bT, b_rows, out, mask, M, N) in enumerate(problems):
        assert out.dtype == torch.float32 and out.stride(1) == 1
        arr[10 * i:10 * i + 10] = [_p(aT) + 2 * 64 * a_row0, a_rows, _p(bT), b_rows, out.data_ptr(), out.stride(0),
                                   0 if mask is None else _p(mask), M, N, 0]
    _lib.check(_lib.lib().vl_dw_grouped(ctypes.cast(arr, ctypes.c_void_p), n, K, 1 if accumulate else 0, _stream()),
               "vl_dw_grouped")


def dw_grouped_rowmajor(problems, rows, accumulate=False):
    """problems = [(dY [rows, >= M] bf16 view (row-major, ld = stride(0)), X [rows, >= N] bf16, out [M,N] fp32, mask or None,
    M, N, colsum partials [ceil(N/256), M] fp32 or None)]: out (+)= dY[:, :M]^T . X[:, :N] from the row-major operands."""
    import ctypes
    n = len(problems)
    arr = (ctypes.c_int64 * (10 * n))()
    for i, (dy, x, out, mask, M, N, cs) in enumerate(problems):
        pa, lda = _pld(dy)
        pb, ldb = _pld(x)
        assert out.dtype == torch.float32 and out.stride(1) == 1
        arr[10 * i:10 * i + 10] = [pa, lda, pb, ldb, out.data_ptr(), out.stride(0), 0 if mask is None else _p(mask), M, N,
                                   0 if cs is None else _p(cs)]
    _lib.check(_lib.lib().vl_dw_grouped_rowmajor(ctypes.cast(arr, ctypes.c_void_p), n, rows, 1 if accumulate else 0,
                                                 _stream()), "vl_dw_grouped_rowmajor")


def dw_grouped_mixed(problems, rows, mode, accumulate=False, budget=0, ws=None):
    """problems = [(a, a_cols, b, b_cols, out, mask, M, N, colsum)]: an operand is a row-major 2-D bf16 view (its *_cols is
    None) when its mode bit is set (bit 0: dY, bit 1: X), else its K-major image (1-D tensor) with the image's column count.
    budget > 0: the stream-K form on that many workgroups with the zero-initialised workspace `ws` (dw_streamk_ws)."""
    import ctypes
    n = len(problems)
    arr = (ctypes.c_int64 * (10 * n))()
    for i, (a, a_cols, b, b_cols, out, mask, M, N, cs) in enumerate(problems):
        pa, lda = _pld(a) if mode & 1 else (_p(a), a_cols)
        pb, ldb = _pld(b) if mode & 2 else (_p(b), b_cols)
        arr[10 * i:10 * i + 10] = [pa, lda, pb, ldb, out.data_ptr(), out.stride(0), 0 if mask is None else _p(mask), M, N,
                                   0 if cs is None else _p(cs)]
    if budget:
        _lib.check(_lib.lib().vl_dw_grouped_streamk(ctypes.cast(arr, ctypes.c_void_p), n, rows, 1 if accumulate else 0, mode,
                                                    int(budget), _p(ws), ws.numel() * ws.element_size(), _stream()),
                   "vl_dw_grouped_streamk")
        return
    _lib.check(_lib.lib().vl_dw_grouped_mixed(ctypes.cast(arr, ctypes.c_void_p), n, rows, 1 if accumulate else 0, mode,
                                              _stream()), "vl_dw_grouped_mixed")


def dw_streamk_ws(budget, device):
    """Zero-initialised workspace of the stream-K weight-gradient GEMM for `budget` workgroups (hand-off flags + one
    partial-tile slot per workgroup); launches that share it must be ordered on one stream."""
    return torch.zeros(_lib.lib().vl_dw_streamk_ws_bytes(int(budget)), dtype=torch.uint8, device=device)


def attn2_fwd(qkv_hi, qkv_lo, addmask, ctx_hi, ctx_lo, lse, B, S, nh, dh, p_drop, seed, nq=None):
    _lib.check(_lib.lib().vl_attn2_fwd(_p(qkv_hi), _p(qkv_lo), _p(addmask), _p(ctx_hi), _p(ctx_lo), _p(lse), B, S, nh, dh,
                                       S if nq is None else nq, float(p_drop), int(seed), _stream()), "vl_attn2_fwd")


def attn2_bwd(qkv_hi, addmask, dctx16, lse, dqkv16, B, S, nh, dh, p_drop, seed, nq=None):
    _lib.check(_lib.lib().vl_attn2_bwd(_p(qkv_hi), _p(addmask), _p(dctx16), _p(lse), _p(dqkv16), B, S, nh, dh,
                                       S if nq is None else nq, float(p_drop), int(seed), _stream()), "vl_attn2_bwd")


def ln_fwd(y, resid, addvec, gamma, beta, eps, out32, out_hi, out_lo, mean, rstd, M, H, group=None, out_stride=0,
           out_off=0, p_pre=0.0, p_post=0.0, seed=0, row_pre=None, row_post=None, orig_row_stride=1, resid_row_stride=1,
           resid_ln=None):
    """resid_ln = (z32, mean, rstd, gamma, beta, row_post | None) of the LayerNorm call whose output is the residual: it is
    recomputed from those instead of read (vl_ln_fwd_rr); excludes `resid`."""
    group = M if group is None else group
    arows = 1 if addvec is None or addvec.dim() == 1 else addvec.shape[0]
    rl = None
    if resid_ln is not None:
        rl = (ctypes.c_int64 * 6)(*[0 if t is None else t.data_ptr() for t in resid_ln])
    _lib.check(_lib.lib().vl_ln_fwd_rr(_p(y), _p(resid), ctypes.cast(rl, ctypes.c_void_p) if rl is not None else None,
                                       _p(addvec), arows, _p(row_pre), _p(row_post), _p(gamma),
                                       _p(beta), float(eps), _p(out32),
                                       _p(out_hi), _p(out_lo), _p(mean), _p(rstd), M, H, group, out_stride, out_off,
                                       float(p_pre), float(p_post), int(seed), orig_row_stride, resid_row_stride, _stream()),
               "vl_ln_fwd")


def ln_bwd_ws(M, H, device):
    return torch.empty(_lib.lib().vl_ln_bwd_ws_floats(M, H), dtype=torch.float32, device=device)


def ln_bwd(dy, z, mean, rstd, gamma, dz, dpre16, dpre32, dgamma, dbeta, dbias, ws, M, H, group=None, out_stride=0,
           out_off=0, p_pre=0.0, p_post=0.0, seed=0, row_pre=None, row_post=None, orig_row_stride=1):
    group = M if group is None else group
    _lib.check(_lib.lib().vl_ln_bwd(_p(dy), _p(z), _p(mean), _p(rstd), _p(gamma), _p(row_pre), _p(row_post), _p(dz),
                                    _p(dpre16), _p(dpre32),
                                    _p(dgamma), _p(dbeta), _p(dbias), _p(ws), M, H, group, out_stride, out_off,
                                    float(p_pre), float(p_post), int(seed), orig_row_stride, _stream()), "vl_ln_bwd")


def ln_bwd_reduce(ws, M, H, dgamma, dbeta, dbias):
    """Column sums of the partials a ln_bwd(..., dgamma=None, dbeta=None, dbias=None) call left in ws."""
    _lib.check(_lib.lib().vl_ln_bwd_reduce(_p(ws), M, H, _p(dgamma), _p(dbeta), _p(dbias), _stream()), "vl_ln_bwd_reduce")


def ln_bwd_reduce2(ws_a, M_a, outs_a, ws_b, M_b, outs_b, H, accumulate=False):
    """Both column-sum reductions of a transformer layer in one launch; outs = (dgamma, dbeta, dbias)."""
    _lib.check(_lib.lib().vl_ln_bwd_reduce2(_p(ws_a), M_a, _p(outs_a[0]), _p(outs_a[1]), _p(outs_a[2]), _p(ws_b), M_b,
                                            _p(outs_b[0]), _p(outs_b[1]), _p(outs_b[2]), H, 1 if accumulate else 0,
                                            _stream()),
               "vl_ln_bwd_reduce2")


def gqa_loss(logits, target, distances, semantic_lambda, loss_score, dlogits, ws):
    B, C = logits.shape
    _lib.check(_lib.lib().vl_gqa_loss(_p(logits), _p(target), _p(distances), B, C, float(semantic_lambda), _p(loss_score),
                                      _p(dlogits), _p(ws), _stream()), "vl_gqa_loss")


def memset_zero(t):
    _lib.check(_lib.lib().vl_memset_zero(_p(t), t.numel() * t.element_size(), _stream()), "vl_memset_zero")


def mask_mul(a, m, out):
    """out = a (*) m : SFT weight_orig * weight_mask and grad (*) mask."""
    assert a.numel() == m.numel() == out.numel()
    _lib.check(_lib.lib().vl_mask_mul(_p(a), _p(m), _p(out), a.numel(), _stream()), "vl_mask_mul")
    return out


def weight_prep(w32, mask32, w_hi, w_lo, wt_hi):
    """W[N,K] (*mask) -> w_hi/w_lo (row slices of a packed [*,K] buffer) and wt_hi (column slice of [K,*])."""
    N, K = w32.shape
    ph, ldw = _pld(w_hi) if w_hi is not None else (None, K)
    pl = _pld(w_lo)[0] if w_lo is not None else None
    pt, ldt = _pld(wt_hi) if wt_hi is not None else (None, N)
    _lib.check(_lib.lib().vl_weight_prep(_p(w32), _p(mask32), ph, pl, pt, N, K, ldw, ldt, _stream()),
               "vl_weight_prep")


def imp_select(w_flat, mask_flat, new_mask_flat, k):
    """One IMP round on the flat concatenation: the k smallest |w| with mask == 1 get mask 0."""
    n = w_flat.numel()
    ws = torch.empty(_lib.lib().vl_imp_ws_bytes(n), dtype=torch.uint8, device=w_flat.device)
    _lib.check(_lib.lib().vl_imp_select(_p(w_flat), _p(mask_flat), _p(new_mask_flat), n, int(k), _p(ws), _stream()),
               "vl_imp_select")
    return new_mask_flat


def weight_prep_multi(table_dev, ndesc, total_tiles):
    _lib.check(_lib.lib().vl_weight_prep_multi(_p(table_dev), ndesc, total_tiles, _stream()), "vl_weight_prep_multi")


ACT_NONE, ACT_RELU, ACT_TANH, ACT_GELU = 0, 1, 2, 3


def act_fwd(z32, M, N, act, p_drop, seed, out32=None, out_hi=None, out_lo=None):
    """y = dropout(act(z)) -> fp32 and / or its (hi, lo) split with zero pad columns up to out_hi.stride(0)."""
    ph, ld16 = _pld(out_hi)
    assert out_lo is None or out_lo.stride(0) == ld16
    _lib.check(_lib.lib().vl_act_fwd(_p(z32), M, N, act, float(p_drop), int(seed), _p(out32), ph,
                                     _pld(out_lo)[0] if out_lo is not None else None, ld16, _stream()), "vl_act_fwd")


def act_bwd(dy32, z32, M, N, act, p_drop, seed, dz32=None, dz16=None):
    """dz = dy * dropout mask * act'(z) -> fp32 and / or bf16 with zero pad columns up to dz16.stride(0)."""
    ph, ld16 = _pld(dz16)
    _lib.check(_lib.lib().vl_act_bwd(_p(dy32), _p(z32), M, N, act, float(p_drop), int(seed), _p(dz32), ph, ld16,
                                     _stream()), "vl_act_bwd")


def split_f32(x32, hi, lo=None):
    _lib.check(_lib.lib().vl_split_f32(_p(x32), _p(hi), _p(lo), x32.numel(), _stream()), "vl_split_f32")


def transpose_bf16(src, dst, M, N):
    ps, ld_in = _pld(src)
    pd, ld_out = _pld(dst)
    _lib.check(_lib.lib().vl_transpose_bf16(ps, pd, M, N, ld_in, ld_out, _stream()), "vl_transpose_bf16")


def colsum_bf16(x16, M, N, out32):
    ws = _tmp(torch.empty(_lib.lib().vl_colsum_ws_floats(M, N), dtype=torch.float32, device=x16.device))
    px, ld = _pld(x16)
    _lib.check(_lib.lib().vl_colsum_bf16(px, M, N, ld, _p(ws), _p(out32), _stream()), "vl_colsum_bf16")
    return out32


def addmask(text_mask, img_mask, out, B, T, V):
    _lib.check(_lib.lib().vl_addmask(_p(text_mask), _p(img_mask), _p(out), B, T, V, _stream()), "vl_addmask")


def embed_text_fwd(ids, seg, word, pos, typ, z32, B, T, H, pad_id):
    _lib.check(_lib.lib().vl_embed_text_fwd(_p(ids), _p(seg), _p(word), _p(pos), _p(typ), _p(z32), B, T, H, pad_id,
                                            _stream()), "vl_embed_text_fwd")


def embed_text_bwd(ids, seg, dz32, dword, dpos, dtyp, B, T, H, pad_id, row_flags=None):
    _lib.check(_lib.lib().vl_embed_text_bwd(_p(ids), _p(seg), _p(dz32), _p(dword), _p(dpos), _p(dtyp), B, T, H,
                                            pad_id, _p(row_flags), _stream()), "vl_embed_text_bwd")


def embed_gather_fwd(ids, table, out32, R, H):
    _lib.check(_lib.lib().vl_embed_gather_fwd(_p(ids), _p(table), _p(out32), R, H, _stream()), "vl_embed_gather_fwd")


def embed_scatter_add(ids, dz32, dtable, R, H, pad_id=-1, row_flags=None):
    _lib.check(_lib.lib().vl_embed_scatter_add(_p(ids), _p(dz32), _p(dtable), R, H, pad_id, _p(row_flags), _stream()),
               "vl_embed_scatter_add")


def loc_linear_fwd(loc, w, b, y32, R, L, H):
    _lib.check(_lib.lib().vl_loc_linear_fwd(_p(loc), _p(w), _p(b), _p(y32), R, L, H, _stream()), "vl_loc_linear_fwd")


DETERMINISTIC_EMBED_BWD = True  # fixed-order reductions in the embedding backward (False: float atomics, arrival order)


_WS_CACHE = {}  # persistent workspaces of the fixed-order reductions, keyed by (kind, device, stream, size): launches of one
                # stream are ordered, so they can share one buffer -- no allocator traffic on the hot path


def _ws(kind, device, nbytes):
    key = (kind, str(device), _stream(), int(nbytes))
    t = _WS_CACHE.get(key)
    if t is None:
        if len(_WS_CACHE) > 64:
            _WS_CACHE.clear()
        t = _WS_CACHE[key] = torch.empty(int(nbytes), dtype=torch.uint8, device=device)
    return t


def loc_linear_bwd(loc, dy32, dw, db, R, L, H, deterministic=None):
    """dw [H, L], db [H] += ...; deterministic: row blocks summed in a fixed order through a workspace (default), else
    float atomics."""
    det = DETERMINISTIC_EMBED_BWD if deterministic is None else deterministic
    ws = _ws("loc", dy32.device, 4 * _lib.lib().vl_loc_bwd_ws_floats(R, H)) if det else None
    _lib.check(_lib.lib().vl_loc_linear_bwd(_p(loc), _p(dy32), _p(dw), _p(db), R, L, H, _p(ws), _stream()),
               "vl_loc_linear_bwd")


def scatter_add_det(tables, dz32, R, H):
    """Deterministic scatter-add of the rows of dz32 [R, H] into several tables at once (csrc/scatter.hip).
    tables = [(ids int64 [R], kind, table fp32 [rows, H], skip, row_flags or None, T)] -- see vl_scatter_add_det."""
    import ctypes
    n = len(tables)
    arr = (ctypes.c_int64 * (6 * n))()
    for i, (ids, kind, table, skip, flags, T) in enumerate(tables):
        assert ids.is_contiguous() and ids.dtype == torch.int64 and table.is_contiguous() and table.dtype == torch.float32
        arr[6 * i:6 * i + 6] = [_p(ids), kind, _p(table), skip, 0 if flags is None else _p(flags), T]
    L = _lib.lib()
    nbytes = L.vl_scatter_det_ws_bytes(n, R, H)
    ws = _ws("scatter", dz32.device, nbytes)
    _lib.check(L.vl_scatter_add_det(ctypes.cast(arr, ctypes.c_void_p), n, _p(dz32), R, H, _p(ws), nbytes, _stream()),
               "vl_scatter_add_det")


def embed_text_bwd_det(ids, seg, dz32, dword, dpos, dtyp, B, T, H, pad_id, row_flags=None):
    """embed_text_bwd with a fixed summation order (same destinations, same semantics: the pad row of the word table receives
    nothing, dword may be None)."""
    tabs = []
    if dword is not None:
        tabs.append((ids.view(-1), 0, dword, pad_id, row_flags, T))
    tabs.append((ids.view(-1), 1, dpos, pad_id, None, T))
    tabs.append((seg.view(-1), 0, dtyp, -1, None, T))
    scatter_add_det(tabs, dz32, B * T, H)


def adamw(param, grad, exp_avg, exp_avg_sq, seg_end, seg_lr, seg_wd, beta1, beta2, eps, step, correct_bias, lr_mult,
          grad_scale_dev=None, grad_scale=1.0, zero_grad=False, row_flags=None, flag_begin=0, flag_rows=0,
          flag_row_len=0, sumsq=None, max_norm=0.0, post=1.0, sumsq_next=None, seg_step=None):
    """sumsq (device scalar) switches the device-side clip: scale = min(1, max_norm / (sqrt(sumsq) * post + 1e-6)) * post.
    seg_step (device int64 [nseg], optional): per-segment step counts for the bias correction (default: `step` everywhere)."""
    _lib.check(_lib.lib().vl_adamw(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), _p(seg_end),
                                   _p(seg_lr), _p(seg_wd), seg_end.numel(), float(beta1), float(beta2), float(eps),
                                   int(step), _p(seg_step), int(bool(correct_bias)), float(lr_mult), _p(grad_scale_dev),
                                   float(grad_scale), _p(sumsq), float(max_norm), float(post), _p(sumsq_next),
                                   int(bool(zero_grad)), _p(row_flags), flag_begin, flag_rows,
                                   flag_row_len, _stream()), "vl_adamw")


def sumsq(x, out, row_flags=None, flag_begin=0, flag_rows=0, flag_row_len=0, ws=None):
    """out[0] += sum(x^2).  ws (fp32, >= vl_sumsq_ws_floats()): fixed-order two-launch reduction (bit-reproducible) instead
    of float atomics."""
    if row_flags is None:
        _lib.check(_lib.lib().vl_sumsq(_p(x), x.numel(), _p(out), _p(ws), _stream()), "vl_sumsq")
    else:
        _lib.check(_lib.lib().vl_sumsq_flagged(_p(x), x.numel(), _p(out), _p(row_flags), flag_begin, flag_rows,
                                               flag_row_len, _p(ws), _stream()), "vl_sumsq_flagged")

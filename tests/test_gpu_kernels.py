"""Per-kernel parity of the HIP path (through the C ABI, include/vlhip.h) against plain fp32/fp64 torch
restatements of the same op.  Needs a real MI355X: ``pytest -m gpu``."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from clg_vqa_amd import ops  # noqa: E402
from clg_vqa_amd.ops import BF16, EPI_BF16, EPI_DGELU_BF16, EPI_F32, EPI_GELU_SPLIT, EPI_SPLIT  # noqa: E402

DEV = "cuda"


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


def _split(x):
    hi = torch.empty_like(x, dtype=BF16)
    lo = torch.empty_like(x, dtype=BF16)
    ops.split_f32(x.contiguous(), hi, lo)
    return hi, lo


def _gelu(x):
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _gelu_grad(x):
    x = x.double()
    return (0.5 * (1 + torch.erf(x / math.sqrt(2))) + x * torch.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)).float()


def test_split_reconstructs_16_bits():
    x = _rand(1000, 37, seed=1)
    hi, lo = _split(x)
    assert torch.equal(hi, x.to(BF16))
    rec = hi.float() + lo.float()
    assert (rec - x).abs().max().item() <= 2.0 ** -16 * x.abs().max().item()


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (128, 128, 64), (300, 200, 264), (1024, 768, 768),
                                   (256, 1842, 768), (17, 9, 8), (512, 3072, 768), (512, 768, 3072), (700, 1000, 320),
                                   (257, 129, 64), (1000, 200, 128)])
def test_gemm_single_pass_bf16(M, N, K):
    a = _rand(M, K, seed=2).to(BF16)
    b = _rand(N, K, seed=3).to(BF16)
    bias = _rand(N, seed=4)
    out = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm_nt(a, None, b, None, M, N, K, 1, EPI_F32, bias=bias, out32=out)
    ref = (a.double() @ b.double().t() + bias.double()).float()
    # products of bf16 values are exact in fp32; only the fp32 accumulation order differs
    torch.testing.assert_close(out, ref, rtol=2e-5, atol=2e-4 * math.sqrt(K / 256))


@pytest.mark.parametrize("M,N,K", [(256, 256, 64), (256, 256, 128), (300, 260, 192), (1000, 700, 448), (2048, 768, 3072),
                                   (3584, 3072, 768), (513, 257, 2304),
                                   # more tiles than CUs, K-tile counts 1, 2, 3, 4 and 12 per tile
                                   (8192, 2560, 64), (8200, 2304, 128), (7000, 3072, 192), (14336, 3072, 768)])
@pytest.mark.parametrize("passes,width", [(1, 2), (1, 3), (3, 2), (3, 3)])
def test_gemm_pingpong_kernel(M, N, K, passes, width):
    """The 8-wave ping-pong kernel behind the 1-pass products (counted vmcnt + staggered barriers): edges, every K-tile
    count parity, and a repeat screen -- the kernel is deterministic, so any run-to-run difference is a staging race."""
    from clg_vqa_amd import _lib
    x, w = _rand(M, K, seed=20), _rand(N, K, seed=21)
    a, al = _split(x)
    b, bl = _split(w)
    if passes == 1:
        al = bl = None
        ref = (a.double() @ b.double().t()).float()
    else:
        ref = (x.double() @ w.double().t()).float()
    try:
        ops.GEMM_TILE = width  # 2: 256 x 256 tiles, 3: 256 x 192 tiles (per-call `tile` of vl_gemm_nt_ex)
        out = torch.full((M, N), float("nan"), device=DEV)
        ops.gemm_nt(a, al, b, bl, M, N, K, passes, EPI_F32, out32=out)
        if passes == 1:  # products of bf16 values are exact in fp32; only the accumulation order differs
            torch.testing.assert_close(out, ref, rtol=2e-5, atol=2e-4 * math.sqrt(max(K, 256) / 256))
        else:            # the dropped lo*lo term: 2^-16 relative per product
            scale = (x.abs().double() @ w.abs().double().t()).max().item()
            err = (out.double() - ref.double()).abs().max().item()
            assert err <= 6e-5 * scale / math.sqrt(K) + 1e-5, (err, scale)
        filler = torch.empty(64 << 20, device=DEV)
        for it in range(8):
            again = torch.full((M, N), float("nan"), device=DEV)
            filler.normal_()  # evict L2 / perturb timing between runs
            ops.gemm_nt(a, al, b, bl, M, N, K, passes, EPI_F32, out32=again)
            assert torch.equal(again, out), it
        # the 16-bit epilogue (paired n-tiles, permuted B rows) against the fp32 one
        sh, sl = torch.empty(M, N, dtype=BF16, device=DEV), torch.empty(M, N, dtype=BF16, device=DEV)
        ops.gemm_nt(a, al, b, bl, M, N, K, passes, EPI_SPLIT, out_hi=sh, out_lo=sl)
        eh, el = _split(out)
        assert torch.equal(sh, eh) and torch.equal(sl, el)
        ops.GEMM_TILE = 6  # the older single-barrier kernel adds the same products in the same k order
        old = torch.empty_like(out)
        ops.gemm_nt(a, al, b, bl, M, N, K, passes, EPI_F32, out32=old)
        if passes == 1:
            assert torch.equal(out, old)
        else:
            torch.testing.assert_close(out, old, rtol=1e-5, atol=1e-5 * math.sqrt(K))
    finally:
        ops.GEMM_TILE = 0


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (300, 200, 264), (1024, 2304, 768), (256, 1842, 768),
                                   (512, 768, 3072), (700, 768, 2048), (300, 200, 96), (999, 333, 32)])
def test_gemm_three_pass_is_fp32_grade(M, N, K):
    x = _rand(M, K, seed=5)
    w = _rand(N, K, seed=6, scale=0.05)
    bias = _rand(N, seed=7)
    resid = _rand(M, N, seed=8)
    xh, xl = _split(x)
    wh, wl = _split(w)
    out = torch.empty(M, N, device=DEV)
    ops.gemm_nt(xh, xl, wh, wl, M, N, K, 3, EPI_F32, bias=bias, resid=resid, out32=out)
    ref = (x.double() @ w.double().t() + bias.double() + resid.double())
    err = (out.double() - ref).abs().max().item()
    scale = (x.double().abs() @ w.double().abs().t()).max().item()
    assert err <= 6e-5 * scale / math.sqrt(K) + 1e-5, (err, scale)
    # and it is far better than a single bf16 pass
    out1 = torch.empty(M, N, device=DEV)
    ops.gemm_nt(xh, None, wh, None, M, N, K, 1, EPI_F32, bias=bias, resid=resid, out32=out1)
    err1 = (out1.double() - ref).abs().max().item()
    assert err < err1 / 20, (err, err1)


@pytest.mark.parametrize("M,N,K", [(256, 768, 3072), (256, 3072, 768), (256, 768, 768), (256, 1842, 1536), (256, 1024, 768),
                                   (128, 768, 768), (8, 1842, 1536), (70, 100, 72), (1000, 768, 3072), (2, 128, 128)])
@pytest.mark.parametrize("passes", [1, 3])
def test_gemm_small_m_path(M, N, K, passes):
    """tile 8: 64 x 64 tiles x K ranges through a workspace + the slab-sum / epilogue launch (the batch-sized products of
    the pooled last layer and the head).  Against fp64, every epilogue against the generic kernel's, and repeatable."""
    x, w = _rand(M, K, seed=50), _rand(N, K, seed=51, scale=0.05)
    bias, resid = _rand(N, seed=52), _rand(M, N, seed=53)
    a, al = _split(x)
    b, bl = _split(w)
    if passes == 1:
        al = bl = None
        ref = a.double() @ b.double().t() + bias.double() + resid.double()
    else:
        ref = x.double() @ w.double().t() + bias.double() + resid.double()
    out = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm_nt(a, al, b, bl, M, N, K, passes, EPI_F32, bias=bias, resid=resid, out32=out, tile=8)
    err = (out.double() - ref).abs().max().item()
    scale = (x.double().abs() @ w.double().abs().t()).max().item()
    assert err <= 6e-5 * scale / math.sqrt(K) + 1e-5, (err, scale)
    again = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm_nt(a, al, b, bl, M, N, K, passes, EPI_F32, bias=bias, resid=resid, out32=again, tile=8)
    assert torch.equal(out, again)
    Np = (N + 7) // 8 * 8  # 16-bit outputs need a leading dimension the vector stores can use
    for epi in (EPI_SPLIT, EPI_BF16, EPI_GELU_SPLIT, EPI_DGELU_BF16):
        got, want = [], []
        for tile, dst in ((8, got), (7, want)):
            hi, lo = torch.zeros(M, Np, dtype=BF16, device=DEV), torch.zeros(M, Np, dtype=BF16, device=DEV)
            aux = torch.zeros(M, Np, dtype=BF16, device=DEV)
            if epi == EPI_DGELU_BF16:
                aux[:, :N] = _rand(M, N, seed=54).to(BF16)
            ops.gemm_nt(a, al, b, bl, M, N, K, passes, epi, bias=None if epi == EPI_DGELU_BF16 else bias,
                        out_hi=hi, out_lo=lo if epi in (EPI_SPLIT, EPI_GELU_SPLIT) else None,
                        aux16=aux if epi in (EPI_GELU_SPLIT, EPI_DGELU_BF16) else None, tile=tile)
            dst += [hi.float(), lo.float(), aux.float()]
        # different accumulation order: fp32 values differ by ~1e-5, i.e. at most one bf16 rounding step in `hi` / `aux`
        # (`lo` alone is not comparable: it is the residual of whichever way `hi` rounded)
        for g, w_ in ((got[0], want[0]), (got[2], want[2])):
            assert ((g - w_).abs() <= 2.0 ** -7 * w_.abs() + 1e-4).all()
        torch.testing.assert_close(got[0] + got[1], want[0] + want[1], rtol=2 ** -7 if epi in (EPI_BF16, EPI_DGELU_BF16) else 1e-4,
                                   atol=1e-4)


@pytest.mark.parametrize("M,N,K", [(768, 768, 14336), (2304, 768, 14336), (768, 3072, 4096), (200, 136, 1000), (128, 128, 64)])
def test_gemm_splitk(M, N, K):
    a = _rand(M, K, seed=40).to(BF16)
    b = _rand(N, K, seed=41).to(BF16)
    out = torch.full((M, N), float("nan"), device=DEV)
    ops.gemm_nt_splitk(a, b, M, N, K, out)
    ref = (a.double() @ b.double().t()).float()
    torch.testing.assert_close(out, ref, rtol=2e-5, atol=2e-4 * math.sqrt(K / 256))


@pytest.mark.parametrize("M,N,K", [(768, 768, 14336), (2304, 768, 1024), (768, 3072, 2048), (3072, 768, 14336),
                                   (264, 136, 128), (776, 1032, 256)])
def test_gemm_tn_splitk(M, N, K):
    """dW = A^T B straight from [K,M] / [K,N] row-major operands (transposing LDS reads)."""
    a = _rand(K, M, seed=42).to(BF16)
    b = _rand(K, N, seed=43).to(BF16)
    out = torch.full((M, N), float("nan"), device=DEV)
    assert ops.gemm_tn_splitk(a, b, M, N, K, out)
    ref = (a.double().t() @ b.double()).float()
    torch.testing.assert_close(out, ref, rtol=2e-5, atol=2e-4 * math.sqrt(K / 256))
    assert not ops.gemm_tn_splitk(a[:, :64].contiguous(), b, 64, N, K, out[:64].contiguous())  # outside the fast path


@pytest.mark.parametrize("M,N,K", [(384, 512, 256), (300, 200, 128), (512, 3072, 768), (260, 776, 64), (100, 72, 64)])
def test_gemm_epilogues(M, N, K):
    x, w, bias = _rand(M, K, seed=9), _rand(N, K, seed=10, scale=0.1), _rand(N, seed=11)
    xh, xl = _split(x)
    wh, wl = _split(w)
    u_ref = (x.double() @ w.double().t() + bias.double()).float()
    # GELU_SPLIT
    u16 = torch.empty(M, N, dtype=BF16, device=DEV)
    hh, hl = torch.empty_like(u16), torch.empty_like(u16)
    ops.gemm_nt(xh, xl, wh, wl, M, N, K, 3, EPI_GELU_SPLIT, bias=bias, out_hi=hh, out_lo=hl, aux16=u16)
    torch.testing.assert_close(u16.float(), _gelu_grad(u_ref), rtol=2 ** -8, atol=1e-4)  # aux16 = bf16(GELU'(u))
    torch.testing.assert_close(hh.float() + hl.float(), _gelu(u_ref), rtol=1e-4, atol=1e-4)
    # SPLIT and BF16
    sh, sl = torch.empty_like(u16), torch.empty_like(u16)
    ops.gemm_nt(xh, xl, wh, wl, M, N, K, 3, EPI_SPLIT, bias=bias, out_hi=sh, out_lo=sl)
    torch.testing.assert_close(sh.float() + sl.float(), u_ref, rtol=1e-4, atol=1e-4)
    bh = torch.empty_like(u16)
    ops.gemm_nt(xh, None, wh, None, M, N, K, 1, EPI_BF16, bias=bias, out_hi=bh)
    torch.testing.assert_close(bh.float(), u_ref, rtol=2e-2, atol=5e-2)
    # DGELU: out = bf16(acc * aux16), aux16 = the derivative the forward epilogue saved
    dh = torch.empty_like(u16)
    ops.gemm_nt(xh, None, wh, None, M, N, K, 1, EPI_DGELU_BF16, out_hi=dh, aux16=u16)
    acc = xh.double() @ wh.double().t()
    torch.testing.assert_close(dh.float(), (acc * u16.double()).float(), rtol=2 ** -7, atol=1e-3)


@pytest.mark.parametrize("M,N,K,persist", [(2048, 1536, 768, 8), (2048, 1536, 768, 13), (3584, 3072, 256, 40), (1000, 700, 128, 3),
                                            (14336, 768, 768, 256), (4096, 2304, 64, 16)])
def test_persistent_gemm_is_bit_identical_to_one_tile_per_workgroup(M, N, K, persist):
    """VL_GX_PERSIST: a workgroup walks several tiles and prefetches the next tile's operands under the epilogue of the
    finished one -- same tiles, same arithmetic: every epilogue and tile configuration the layer stack uses must give the
    same bits as the one-tile-per-workgroup launch (ragged edges, a single K-tile and uneven tile counts included)."""
    x, w, bias = _rand(M, K, seed=19), _rand(N, K, seed=20, scale=0.1), _rand(N, seed=21)
    resid = _rand(M, N, seed=22)
    xh, xl = _split(x)
    wh, wl = _split(w)
    for tile in (2, 3, 5):
        def run(p_):
            o = {}
            o["f32_3"] = torch.full((M, N), float("nan"), device=DEV)
            ops.gemm_nt(xh, xl, wh, wl, M, N, K, 3, EPI_F32, bias=bias, out32=o["f32_3"], tile=tile, persist=p_)
            o["sh"], o["sl"] = (torch.zeros(M, N, dtype=BF16, device=DEV) for _ in range(2))
            ops.gemm_nt(xh, xl, wh, wl, M, N, K, 3, EPI_SPLIT, bias=bias, out_hi=o["sh"], out_lo=o["sl"], tile=tile, persist=p_)
            o["u16"], o["hh"], o["hl"] = (torch.zeros(M, N, dtype=BF16, device=DEV) for _ in range(3))
            ops.gemm_nt(xh, xl, wh, wl, M, N, K, 3, EPI_GELU_SPLIT, bias=bias, out_hi=o["hh"], out_lo=o["hl"], aux16=o["u16"],
                        tile=tile, persist=p_)
            o["f32_1"] = torch.full((M, N), float("nan"), device=DEV)
            ops.gemm_nt(xh, None, wh, None, M, N, K, 1, EPI_F32, resid=resid, out32=o["f32_1"], tile=tile, persist=p_)
            o["bh"] = torch.zeros(M, N, dtype=BF16, device=DEV)
            ops.gemm_nt(xh, None, wh, None, M, N, K, 1, EPI_BF16, out_hi=o["bh"], tile=tile, persist=p_)
            o["dh"] = torch.zeros(M, N, dtype=BF16, device=DEV)
            ops.gemm_nt(xh, None, wh, None, M, N, K, 1, EPI_DGELU_BF16, out_hi=o["dh"], aux16=o["u16"], tile=tile, persist=p_)
            return o
        ref, per = run(0), run(persist)
        for k in ref:
            assert torch.equal(ref[k].view(torch.int16) if ref[k].dtype == BF16 else ref[k], per[k].view(torch.int16) if per[k].dtype == BF16 else per[k]), (tile, k)
    torch.testing.assert_close(ref["f32_3"].double(), x.double() @ w.double().t() + bias.double(), rtol=1e-4, atol=1e-4)


def test_gemm_rejects_bad_arguments():
    a = torch.zeros(16, 12, dtype=BF16, device=DEV)
    out = torch.zeros(16, 16, device=DEV)
    with pytest.raises(RuntimeError, match="multiples of 8"):
        ops.gemm_nt(a, None, a, None, 16, 16, 12, 1, EPI_F32, out32=out)
    with pytest.raises(RuntimeError, match="no CPU fallback|device"):
        ops.gemm_nt(a.cpu(), None, a.cpu(), None, 16, 16, 8, 1, EPI_F32, out32=out)


def _attn_ref(qkv, addmask, B, S, nh, keep=None):
    H = nh * 64
    q, k, v = [t.view(B, S, nh, 64).permute(0, 2, 1, 3) for t in qkv.view(B, S, 3 * H).split(H, dim=-1)]
    s = q @ k.transpose(-1, -2) / 8.0 + addmask.view(B, 1, 1, S)
    p = torch.softmax(s, dim=-1)
    lse = torch.logsumexp(s, dim=-1)
    if keep is not None:
        p = p * keep
    return (p @ v).permute(0, 2, 1, 3).reshape(B * S, H), lse


@pytest.mark.parametrize("H", [256, 768, 1536])
def test_layernorm_forward_backward(H):
    M = 333
    y, resid, addvec = _rand(M, H, seed=17), _rand(M, H, seed=18), _rand(H, seed=19)
    gamma, beta = 1 + 0.1 * _rand(H, seed=20), 0.1 * _rand(H, seed=21)
    eps = 1e-5
    z = y.clone()
    out, mean, rstd = torch.empty(M, H, device=DEV), torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    hi, lo = torch.empty(M, H, dtype=BF16, device=DEV), torch.empty(M, H, dtype=BF16, device=DEV)
    ops.ln_fwd(z, resid, addvec, gamma, beta, eps, out, hi, lo, mean, rstd, M, H)
    zr = (y.double() + resid.double() + addvec.double()).requires_grad_(True)
    g64, b64 = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    u = zr.mean(-1, keepdim=True)
    s = (zr - u).pow(2).mean(-1, keepdim=True)
    ref = g64 * ((zr - u) / torch.sqrt(s + eps)) + b64
    torch.testing.assert_close(z.double(), zr.detach(), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(out.double(), ref.detach(), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(hi.double() + lo.double(), ref.detach(), rtol=1e-4, atol=1e-4)
    dy = _rand(M, H, seed=22)
    ref.backward(dy.double())
    dz, dpre16, dpre32 = torch.empty(M, H, device=DEV), torch.empty(M, H, dtype=BF16, device=DEV), torch.empty(M, H, device=DEV)
    dg, db, dbias = torch.empty(H, device=DEV), torch.empty(H, device=DEV), torch.empty(H, device=DEV)
    ops.ln_bwd(dy, z, mean, rstd, gamma, dz, dpre16, dpre32, dg, db, dbias, ops.ln_bwd_ws(M, H, DEV), M, H)
    torch.testing.assert_close(dz.double(), zr.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dpre32, dz)
    torch.testing.assert_close(dpre16.float(), dz, rtol=2 ** -7, atol=1e-6)
    torch.testing.assert_close(dg.double(), g64.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db.double(), b64.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dbias.double(), zr.grad.sum(0), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("H,stride", [(768, 1), (768, 7), (128, 1)])
def test_layernorm_recomputed_residual_is_bit_identical(H, stride):
    """vl_ln_fwd_rr: the residual re-evaluated from the producing LayerNorm's (z, mean, rstd, gamma, beta, row_post) gives the
    bits of the call that reads the stored fp32 output -- dense rows and the strided rows of the pooled last layer; the
    producer then runs without an fp32 output at all."""
    M0 = 7 * 40
    M = M0 // stride
    g = torch.Generator().manual_seed(5)
    y0 = _rand(M0, H, seed=61)
    res0 = _rand(M0, H, seed=62)
    g0, b0 = (1.0 + 0.1 * torch.randn(H, generator=g)).to(DEV), (0.1 * torch.randn(H, generator=g)).to(DEV)
    g1, b1 = (1.0 + 0.1 * torch.randn(H, generator=g)).to(DEV), (0.1 * torch.randn(H, generator=g)).to(DEV)
    row_post = (torch.rand(M0, generator=g) > 0.2).float().to(DEV)
    mean0, rstd0 = torch.empty(M0, device=DEV), torch.empty(M0, device=DEV)
    out0 = torch.empty(M0, H, device=DEV)
    h0, l0 = torch.empty(M0, H, dtype=BF16, device=DEV), torch.empty(M0, H, dtype=BF16, device=DEV)
    z0 = y0.clone()
    ops.ln_fwd(z0, res0, None, g0, b0, 1e-12, out0, h0, l0, mean0, rstd0, M0, H, p_pre=0.1, seed=9, row_post=row_post)
    # the producer again, without its fp32 output
    z0b, m0b, r0b = y0.clone(), torch.empty_like(mean0), torch.empty_like(rstd0)
    h0b, l0b = torch.empty_like(h0), torch.empty_like(l0)
    ops.ln_fwd(z0b, res0, None, g0, b0, 1e-12, None, h0b, l0b, m0b, r0b, M0, H, p_pre=0.1, seed=9, row_post=row_post)
    assert torch.equal(z0b, z0) and torch.equal(m0b, mean0) and torch.equal(h0b, h0) and torch.equal(l0b, l0)
    # the consumer: rows r * stride of the producer are its residual
    y1 = _rand(M, H, seed=63)
    outs = []
    for mode in ("stored", "recomputed"):
        z1, m1, r1 = y1.clone(), torch.empty(M, device=DEV), torch.empty(M, device=DEV)
        o1 = torch.empty(M, H, device=DEV)
        h1, l1 = torch.empty(M, H, dtype=BF16, device=DEV), torch.empty(M, H, dtype=BF16, device=DEV)
        kw = dict(resid_row_stride=stride, orig_row_stride=stride, p_pre=0.1, seed=11)
        if mode == "stored":
            ops.ln_fwd(z1, out0, None, g1, b1, 1e-12, o1, h1, l1, m1, r1, M, H, **kw)
        else:
            ops.ln_fwd(z1, None, None, g1, b1, 1e-12, o1, h1, l1, m1, r1, M, H, resid_ln=(z0b, m0b, r0b, g0, b0, row_post), **kw)
        outs.append((z1, m1, r1, o1, h1, l1))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    with pytest.raises(RuntimeError):
        ops.ln_fwd(y1.clone(), out0, None, g1, b1, 1e-12, o1, None, None, m1, r1, M, H, resid_ln=(z0b, m0b, r0b, g0, b0, None))


def test_layernorm_row_map_and_dropout():
    B, T, V, H = 3, 5, 7, 256
    S = T + V
    y = _rand(B * V, H, seed=23)
    gamma, beta = torch.ones(H, device=DEV), torch.zeros(H, device=DEV)
    out = torch.zeros(B * S, H, device=DEV)
    mean, rstd = torch.empty(B * V, device=DEV), torch.empty(B * V, device=DEV)
    z = y.clone()
    ops.ln_fwd(z, None, None, gamma, beta, 1e-5, out, None, None, mean, rstd, B * V, H, group=V, out_stride=S,
               out_off=T)
    ref = torch.nn.functional.layer_norm(y, (H,), eps=1e-5)
    o3 = out.view(B, S, H)
    assert o3[:, :T].abs().max().item() == 0.0
    torch.testing.assert_close(o3[:, T:].reshape(B * V, H), ref, rtol=1e-5, atol=1e-5)
    # dropout: p_pre drops entries of y before the residual add; p_post drops LN outputs
    p = 0.3
    z2 = y.clone()
    resid = torch.zeros_like(y)
    out2 = torch.empty(B * V, H, device=DEV)
    ops.ln_fwd(z2, resid, None, gamma, beta, 1e-5, out2, None, None, mean, rstd, B * V, H, p_pre=p, seed=5)
    kept = z2 != 0
    assert abs(kept.float().mean().item() - (1 - p)) < 0.03
    torch.testing.assert_close(z2[kept], (y / (1 - p))[kept], rtol=1e-6, atol=1e-6)
    # backward regenerates the same masks
    dy = _rand(B * V, H, seed=24)
    dz, dpre = torch.empty_like(y), torch.empty_like(y)
    dg, db, dbias = torch.empty(H, device=DEV), torch.empty(H, device=DEV), torch.empty(H, device=DEV)
    ops.ln_bwd(dy, z2, mean, rstd, gamma, dz, None, dpre, dg, db, dbias, ops.ln_bwd_ws(B * V, H, DEV), B * V, H,
               p_pre=p, seed=5)
    assert torch.equal(dpre != 0, kept & (dz != 0))
    # the two halves of the backward on their own: partials only, then the column sums -- identical to the fused call
    ws2 = ops.ln_bwd_ws(B * V, H, DEV)
    dz_b, dpre_b = torch.empty_like(y), torch.empty_like(y)
    ops.ln_bwd(dy, z2, mean, rstd, gamma, dz_b, None, dpre_b, None, None, None, ws2, B * V, H, p_pre=p, seed=5)
    dg_b, db_b, dbias_b = torch.empty(H, device=DEV), torch.empty(H, device=DEV), torch.empty(H, device=DEV)
    ops.ln_bwd_reduce(ws2, B * V, H, dg_b, db_b, dbias_b)
    assert torch.equal(dz_b, dz) and torch.equal(dg_b, dg) and torch.equal(db_b, db) and torch.equal(dbias_b, dbias)
    torch.testing.assert_close(dpre[kept], (dz / (1 - p))[kept], rtol=1e-6, atol=1e-7)
    z3 = y.clone()
    out3 = torch.empty(B * V, H, device=DEV)
    ops.ln_fwd(z3, None, None, gamma, beta, 1e-5, out3, None, None, mean, rstd, B * V, H, p_post=p, seed=9)
    kept3 = out3 != 0
    assert abs(kept3.float().mean().item() - (1 - p)) < 0.03
    torch.testing.assert_close(out3[kept3], (ref / (1 - p))[kept3], rtol=1e-5, atol=1e-5)


def test_mask_mul_weight_prep_transpose_colsum():
    N, K = 200, 136
    w = _rand(N, K, seed=25)
    mask = (torch.rand(N, K, generator=torch.Generator().manual_seed(26)) < 0.6).float().to(DEV)
    out = torch.empty_like(w)
    ops.mask_mul(w, mask, out)
    assert torch.equal(out, w * mask)
    n_odd = torch.arange(1003, device=DEV, dtype=torch.float32)
    o2 = torch.empty_like(n_odd)
    ops.mask_mul(n_odd, n_odd, o2)
    assert torch.equal(o2, n_odd * n_odd)
    hi, lo = torch.empty(N, K, dtype=BF16, device=DEV), torch.empty(N, K, dtype=BF16, device=DEV)
    t_hi = torch.zeros(K, N + 8, dtype=BF16, device=DEV)
    ops.weight_prep(w, mask, hi, lo, t_hi[:, :N])
    wm = w * mask
    assert torch.equal(hi, wm.to(BF16))
    assert (hi.float() + lo.float() - wm).abs().max().item() <= 2.0 ** -16 * wm.abs().max().item()
    assert torch.equal(t_hi[:, :N], wm.to(BF16).t())
    assert t_hi[:, N:].abs().max().item() == 0
    x = _rand(300, 72, seed=27).to(BF16)
    xt = torch.empty(72, 304, dtype=BF16, device=DEV)
    ops.transpose_bf16(x, xt, 300, 72)
    assert torch.equal(xt[:, :300], x.t())
    cs = torch.empty(72, device=DEV)
    ops.colsum_bf16(x, 300, 72, cs)
    torch.testing.assert_close(cs, x.float().sum(0), rtol=1e-5, atol=1e-4)
    for M, N, off in [(14336, 3072, 0), (1000, 2304, 0), (77, 20, 0), (513, 768, 768)]:  # 16-byte path, ragged, column slice
        wide = _rand(M, N + off + 8, seed=29).to(BF16)
        xs = wide[:, off:off + N]
        cs = torch.full((N,), float("nan"), device=DEV)
        ops.colsum_bf16(xs, M, N, cs)
        torch.testing.assert_close(cs, xs.double().sum(0).float(), rtol=1e-5, atol=1e-3 * math.sqrt(M / 256))


def test_addmask_embeddings_loc():
    B, T, V, H, vocab, L = 3, 9, 5, 256, 50, 7
    g = torch.Generator().manual_seed(28)
    ids = torch.randint(2, vocab, (B, T), generator=g)
    ids[0, 6:] = 1
    ids[2, 3:] = 1
    ids = ids.to(DEV)
    seg = torch.zeros(B, T, dtype=torch.int64, device=DEV)
    tm = (ids != 1).long()
    im = torch.ones(B, V, dtype=torch.int64, device=DEV)
    im[1, 3:] = 0
    am = torch.empty(B * (T + V), device=DEV)
    ops.addmask(tm, im, am, B, T, V)
    ref = (1.0 - torch.cat([tm, im], 1).float()) * -10000.0
    assert torch.equal(am.view(B, T + V), ref)
    word, pos, typ = _rand(vocab, H, seed=29), _rand(T + 2, H, seed=30), _rand(2, H, seed=31)
    z = torch.empty(B * T, H, device=DEV)
    ops.embed_text_fwd(ids, seg, word, pos, typ, z, B, T, H, 1)
    mask = ids.ne(1).int()
    pid = (torch.cumsum(mask, 1) * mask).long() + 1
    zref = word[ids] + pos[pid] + typ[seg]
    torch.testing.assert_close(z.view(B, T, H), zref, rtol=1e-6, atol=1e-6)
    dz = _rand(B * T, H, seed=32)
    dword, dpos, dtyp = torch.zeros_like(word), torch.zeros_like(pos), torch.zeros_like(typ)
    ops.embed_text_bwd(ids, seg, dz, dword, dpos, dtyp, B, T, H, 1)
    rw = torch.zeros_like(word).index_add_(0, ids.view(-1), dz)
    rw[1] = 0  # padding_idx row gets no gradient
    torch.testing.assert_close(dword, rw, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dpos, torch.zeros_like(pos).index_add_(0, pid.view(-1), dz), rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(dtyp, torch.zeros_like(typ).index_add_(0, seg.view(-1), dz), rtol=1e-5, atol=1e-5)
    R = B * V
    loc, wl, bl = _rand(R, L, seed=33), _rand(H, L, seed=34), _rand(H, seed=35)
    y = torch.empty(R, H, device=DEV)
    ops.loc_linear_fwd(loc, wl, bl, y, R, L, H)
    torch.testing.assert_close(y, loc @ wl.t() + bl, rtol=1e-5, atol=1e-5)
    dy = _rand(R, H, seed=36)
    dw, db = torch.zeros(H, L, device=DEV), torch.zeros(H, device=DEV)
    ops.loc_linear_bwd(loc, dy, dw, db, R, L, H)
    torch.testing.assert_close(dw, dy.t() @ loc, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(db, dy.sum(0), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,T,H,vocab", [(256, 20, 768, 250002), (37, 23, 128, 50), (2, 5, 64, 9), (64, 40, 256, 300)])
def test_deterministic_scatter_add_matches_index_add_and_is_reproducible(B, T, H, vocab):
    """vl_scatter_add_det (sort + ordered run sums, one owner per table row) = nn.Embedding's dense backward for the word
    (padding_idx skipped), RoBERTa-position and token-type tables; bit-equal across runs; rows add to what is there."""
    g = torch.Generator().manual_seed(B + T)
    pad = 1
    ids = torch.randint(2, min(vocab, 40), (B, T), generator=g)   # few distinct ids: long runs, many block crossings
    ids[:, 0] = 0
    lens = torch.randint(1, T + 1, (B,), generator=g)
    for b in range(B):
        ids[b, int(lens[b]):] = pad
    if B > 2:
        ids[1, 2] = pad  # a pad in the middle of a sample: the position ids must follow the cumulative count
    seg = torch.randint(0, 2, (B, T), generator=g)
    ids, seg = ids.to(DEV), seg.to(DEV)
    R = B * T
    dz = _rand(R, H, seed=5)
    nz = (ids != pad).long()
    pos_ids = (torch.cumsum(nz, 1) * nz + pad).view(-1)          # embeddings.py:157-170
    base = [_rand(vocab if vocab < 1000 else 64, H, seed=6), _rand(T + 3, H, seed=7), _rand(2, H, seed=8)]
    if vocab >= 1000:
        base[0] = torch.zeros(vocab, H, device=DEV)
    outs = []
    for rep in range(2):
        dword, dpos, dtyp = (t.clone() for t in base)
        flags = torch.zeros(dword.shape[0], dtype=torch.uint8, device=DEV)
        ops.embed_text_bwd_det(ids, seg, dz, dword, dpos, dtyp, B, T, H, pad, row_flags=flags)
        outs.append((dword, dpos, dtyp, flags))
    for a_, b_ in zip(outs[0], outs[1]):
        assert torch.equal(a_, b_)
    dword, dpos, dtyp, flags = outs[0]
    keep = (ids.view(-1) != pad)
    ref_w = base[0].double().index_add_(0, ids.view(-1)[keep], dz.double()[keep])
    ref_p = base[1].double().index_add_(0, pos_ids, dz.double())
    ref_t = base[2].double().index_add_(0, seg.view(-1), dz.double())
    tol = 2e-6 * math.sqrt(R)
    torch.testing.assert_close(dword.double(), ref_w, rtol=1e-6, atol=tol)
    torch.testing.assert_close(dpos.double(), ref_p, rtol=1e-6, atol=tol * 4)
    torch.testing.assert_close(dtyp.double(), ref_t, rtol=1e-6, atol=tol * 16)
    touched = torch.zeros_like(flags)
    touched[ids.view(-1)[keep]] = 1
    assert torch.equal(flags, touched)


def test_deterministic_loc_bwd_and_sumsq():
    R, L, H = 9216, 7, 768
    loc, dy = _rand(R, L, seed=33), _rand(R, H, seed=36)
    res = []
    for rep in range(2):
        dw, db = torch.ones(H, L, device=DEV), torch.ones(H, device=DEV)
        ops.loc_linear_bwd(loc, dy, dw, db, R, L, H, deterministic=True)
        res.append((dw, db))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    torch.testing.assert_close(res[0][0].double(), 1.0 + dy.double().t() @ loc.double(), rtol=1e-5, atol=1e-3)
    torch.testing.assert_close(res[0][1].double(), 1.0 + dy.double().sum(0), rtol=1e-5, atol=1e-3)
    x = _rand(3_000_001, seed=44)
    ws = torch.empty(2048, device=DEV)
    o = [torch.full((1,), 2.0, device=DEV) for _ in range(3)]
    ops.sumsq(x, o[0], ws=ws)
    ops.sumsq(x, o[1], ws=ws)
    assert torch.equal(o[0], o[1])
    torch.testing.assert_close(o[0].double().cpu(), 2.0 + (x.double() ** 2).sum().view(1).cpu(), rtol=1e-5, atol=1e-3)


def test_adamw_and_sumsq():
    n = 5000
    p, g = _rand(n, seed=37), _rand(n, seed=38)
    m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    seg_end = torch.tensor([1000, 1004, 5000], dtype=torch.int64, device=DEV)
    seg_lr = torch.tensor([4e-5, 1e-4, 4e-5], device=DEV)
    seg_wd = torch.tensor([1e-4, 0.0, 1e-4], device=DEV)
    p0, g0 = p.double().cpu(), g.double().cpu()
    rp, rm, rv = p0.clone(), torch.zeros(n, dtype=torch.float64), torch.zeros(n, dtype=torch.float64)
    lr = torch.cat([torch.full((1000,), 4e-5), torch.full((4,), 1e-4), torch.full((3996,), 4e-5)]).double()
    wd = torch.cat([torch.full((1000,), 1e-4), torch.zeros(4), torch.full((3996,), 1e-4)]).double()
    b1, b2, eps, lr_mult, gs = 0.9, 0.999, 1e-6, 0.5, 0.7
    for step in (1, 2, 3):
        ops.adamw(p, g, m, v, seg_end, seg_lr, seg_wd, b1, b2, eps, step, True, lr_mult, None, gs, False)
        gg = g0 * gs
        rm = b1 * rm + (1 - b1) * gg
        rv = b2 * rv + (1 - b2) * gg * gg
        ss = lr * lr_mult * math.sqrt(1 - b2 ** step) / (1 - b1 ** step)
        rp = rp - ss * rm / (rv.sqrt() + eps)
        rp = rp - lr * lr_mult * wd * rp
    torch.testing.assert_close(p.double().cpu(), rp, rtol=1e-5, atol=1e-7)
    # a negative segment learning rate = "this parameter received no gradient": p, m, v (and g) of the segment are not
    # touched at all (pytorch_transformers.AdamW: `if p.grad is None: continue`), the others update as before
    ps, ms, vs, gs_ = p.clone(), m.clone(), v.clone(), g.clone()
    ops.adamw(p, g, m, v, seg_end, torch.tensor([4e-5, -1.0, 4e-5], device=DEV), seg_wd, b1, b2, eps, 4, True, lr_mult, None, gs,
              True)
    for t, t0 in ((p, ps), (m, ms), (v, vs), (g, gs_)):
        assert torch.equal(t[1000:1004], t0[1000:1004])
    assert not torch.equal(p[:1000], ps[:1000]) and not torch.equal(m[1004:], ms[1004:])
    assert (g[:1000] == 0).all() and (g[1004:] == 0).all()  # zero_grad on the active segments
    g.copy_(gs_)
    out = torch.zeros(1, device=DEV)
    ops.sumsq(g, out)
    torch.testing.assert_close(out.double().cpu(), (g0 * g0).sum().view(1), rtol=1e-5, atol=1e-5)
    # flagged form: a table range whose unflagged rows are exact zeros (and are not read)
    rows, rl, beg = 37, 24, 40
    x = _rand(beg + rows * rl + 19, seed=31)
    flags = (torch.rand(rows, generator=torch.Generator().manual_seed(32)) < 0.4).to(torch.uint8).to(DEV)
    tbl = x[beg:beg + rows * rl].view(rows, rl)
    tbl[flags == 0] = 0
    o1, o2 = torch.zeros(1, device=DEV), torch.zeros(1, device=DEV)
    ops.sumsq(x, o1)
    ops.sumsq(x, o2, row_flags=flags, flag_begin=beg, flag_rows=rows, flag_row_len=rl)
    torch.testing.assert_close(o2, o1, rtol=1e-5, atol=1e-6)
    tbl[flags == 0] = 7.0  # proves those rows are skipped
    o3 = torch.zeros(1, device=DEV)
    ops.sumsq(x, o3, row_flags=flags, flag_begin=beg, flag_rows=rows, flag_row_len=rl)
    torch.testing.assert_close(o3, o1, rtol=1e-5, atol=1e-6)


def test_adamw_untouched_row_fast_path_is_bit_identical():
    rows, H, extra = 50, 64, 40
    n = rows * H + extra
    p0 = _rand(n, seed=50)
    g = _rand(n, seed=51)
    touched = torch.zeros(rows, dtype=torch.uint8, device=DEV)
    touched[[3, 17, 18, 49]] = 1
    gt = g[:rows * H].view(rows, H)
    gt[touched == 0] = 0  # untouched rows have no gradient
    seg_end = torch.tensor([rows * H, n], dtype=torch.int64, device=DEV)
    seg_lr = torch.tensor([4e-5, 1e-4], device=DEV)
    seg_wd = torch.tensor([1e-2, 0.0], device=DEV)
    res = []
    for flags in (None, touched):
        p, gg = p0.clone(), g.clone()
        m, v = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        for step in (1, 2, 3):
            kw = {} if flags is None else dict(row_flags=flags, flag_begin=0, flag_rows=rows, flag_row_len=H)
            ops.adamw(p, gg, m, v, seg_end, seg_lr, seg_wd, 0.9, 0.999, 1e-6, step, True, 1.0, None, 1.0, False, **kw)
        res.append((p, m, v))
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("M,N,K", [(448, 512, 256), (300, 260, 192), (1000, 700, 64), (3584, 3072, 768), (2304, 520, 96)])
def test_gemm_224_row_tiles(M, N, K):
    """The 224 x 256 ping-pong tile (FFN1 forward / its dX backward at B*S = 14336: 3 full rounds instead of 2.6) computes
    the same sums in the same order as the 256 x 256 tile: bit-identical outputs for both fused epilogues."""
    x, w, bias = _rand(M, K, seed=70), _rand(N, K, seed=71, scale=0.1), _rand(N, seed=72)
    xh, xl = _split(x)
    wh, wl = _split(w)
    outs = {}
    try:
        for width in (2, 5):
            ops.GEMM_TILE = width
            u16 = torch.full((M, N), float("nan"), dtype=BF16, device=DEV)
            hh, hl, dh = torch.full_like(u16, float("nan")), torch.full_like(u16, float("nan")), torch.full_like(u16, float("nan"))
            if K % 32 == 0:
                ops.gemm_nt(xh, xl, wh, wl, M, N, K, 3, EPI_GELU_SPLIT, bias=bias, out_hi=hh, out_lo=hl, aux16=u16)
            if K % 64 == 0:
                aux = _rand(M, N, seed=73).to(BF16)
                ops.gemm_nt(xh, None, wh, None, M, N, K, 1, EPI_DGELU_BF16, out_hi=dh, aux16=aux)
            outs[width] = (u16, hh, hl, dh)
    finally:
        ops.GEMM_TILE = 0
    for a, b in zip(outs[2], outs[5]):
        assert torch.equal(a.view(torch.int16), b.view(torch.int16))  # bit patterns (NaN-filled where not computed)
    if K % 32 == 0:
        u_ref = (x.double() @ w.double().t() + bias.double()).float()
        torch.testing.assert_close(outs[5][0].float(), _gelu_grad(u_ref), rtol=2 ** -8, atol=1e-4)
        torch.testing.assert_close(outs[5][1].float() + outs[5][2].float(), _gelu(u_ref), rtol=1e-4, atol=1e-4)


def _attn2_inputs(B, S, nh, seed, scale=1.5):
    H = nh * 64
    qkv = _rand(B * S, 3 * H, seed=seed, scale=scale)
    hi, lo = _split(qkv)
    return qkv, hi, lo


@pytest.mark.parametrize("B,T,V", [(3, 20, 36), (2, 40, 36), (2, 20, 100), (2, 40, 100), (1, 13, 20), (1, 60, 100), (2, 1, 1),
                                   (1, 30, 50)])
def test_attention2_forward_backward(B, T, V):
    """bf16-pipe attention: forward (3-term split) is fp32-grade against fp64 math on the same (hi + lo) inputs; the
    backward (single-pass bf16) against autograd within bf16 operand rounding."""
    S, nh = T + V, 12
    H = nh * 64
    qkv, hi, lo = _attn2_inputs(B, S, nh, 12)
    m = torch.ones(B, S, device=DEV)
    if T > 6:
        m[0, T - 5:T] = 0  # padded text tokens in sample 0
    addmask = ((1 - m) * -10000.0).reshape(-1).contiguous()
    ctx_hi = torch.full((B * S, H), float("nan"), dtype=BF16, device=DEV)
    ctx_lo = torch.full_like(ctx_hi, float("nan"))
    lse = torch.empty(B * nh * S, device=DEV)
    ops.attn2_fwd(hi, lo, addmask, ctx_hi, ctx_lo, lse, B, S, nh, 64, 0.0, 1)
    q16 = (hi.double() + lo.double())
    qd = q16.clone().requires_grad_(True)
    ref, lse_ref = _attn_ref(qd, addmask.double().view(B, S), B, S, nh)
    got = ctx_hi.double() + ctx_lo.double()
    err = (got - ref.detach()).abs().max().item()
    assert err < 3e-5 * max(1.0, ref.abs().max().item()), err
    torch.testing.assert_close(lse.double().view(B, nh, S), lse_ref.detach(), rtol=1e-5, atol=2e-4)
    # backward: reference = autograd at the bf16-rounded operands the kernel reads (hi halves, bf16 dO)
    dctx = _rand(B * S, H, seed=13)
    d16 = dctx.to(BF16)
    qh = hi.double().clone().requires_grad_(True)
    ref_h, _ = _attn_ref(qh, addmask.double().view(B, S), B, S, nh)
    ref_h.backward(d16.double())
    dqkv = torch.full((B * S, 3 * H), float("nan"), dtype=BF16, device=DEV)
    ops.attn2_bwd(hi, addmask, d16, lse, dqkv, B, S, nh, 64, 0.0, 1)
    g = qh.grad
    assert torch.isfinite(dqkv.float()).all()
    rel = (dqkv.double() - g).norm().item() / g.norm().item()
    err = (dqkv.double() - g).abs().max().item()
    print("attn2 bwd B=%d S=%d: rel-L2 %.3e, max abs %.3e (grad max %.3e)" % (B, S, rel, err, g.abs().max().item()))
    assert rel <= 2e-2 and err <= 3e-2 * g.abs().max().item()  # P / dS are rounded to bf16 inside (2^-9 relative each)


@pytest.mark.parametrize("B,S", [(3, 56), (2, 120), (1, 10), (2, 140), (1, 156)])  # (4 / 8 / 9 / 10 key-tile instances)
def test_attention2_head_dim_32(B, S):
    """head dim 32 (the tiny c1 config: hidden 128, 4 heads)."""
    nh, dh = 4, 32
    H = nh * dh
    qkv = _rand(B * S, 3 * H, seed=44, scale=1.5)
    hi, lo = _split(qkv)
    addmask = torch.zeros(B * S, device=DEV)
    addmask[S - 2:S] = -10000.0
    ctx_hi = torch.full((B * S, H), float("nan"), dtype=BF16, device=DEV)
    ctx_lo = torch.full_like(ctx_hi, float("nan"))
    lse = torch.empty(B * nh * S, device=DEV)
    ops.attn2_fwd(hi, lo, addmask, ctx_hi, ctx_lo, lse, B, S, nh, dh, 0.0, 1)

    def ref_fn(x):
        x = x.view(B, S, 3, nh, dh)
        q, k, v = x[:, :, 0].permute(0, 2, 1, 3), x[:, :, 1].permute(0, 2, 1, 3), x[:, :, 2].permute(0, 2, 1, 3)
        sc = q @ k.transpose(-1, -2) / math.sqrt(dh) + addmask.double().view(B, 1, 1, S)
        return (torch.softmax(sc, -1) @ v).permute(0, 2, 1, 3).reshape(B * S, H)

    ref = ref_fn(hi.double() + lo.double())
    got = ctx_hi.double() + ctx_lo.double()
    assert (got - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item())
    d16 = _rand(B * S, H, seed=45).to(BF16)
    qh = hi.double().clone().requires_grad_(True)
    ref_fn(qh).backward(d16.double())
    dqkv = torch.full((B * S, 3 * H), float("nan"), dtype=BF16, device=DEV)
    ops.attn2_bwd(hi, addmask, d16, lse, dqkv, B, S, nh, dh, 0.0, 1)
    rel = (dqkv.double() - qh.grad).norm().item() / qh.grad.norm().item()
    assert rel <= 2e-2, rel


@pytest.mark.parametrize("H", [128])
def test_layernorm_hidden_128(H):
    M = 37
    y, resid = _rand(M, H, seed=5), _rand(M, H, seed=6)
    gamma, beta = _rand(H, seed=7) + 1.0, _rand(H, seed=8)
    out, mean, rstd = torch.empty(M, H, device=DEV), torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    oh, ol = torch.empty(M, H, dtype=BF16, device=DEV), torch.empty(M, H, dtype=BF16, device=DEV)
    z = y.clone()
    ops.ln_fwd(z, resid, None, gamma, beta, 1e-5, out, oh, ol, mean, rstd, M, H)
    zr = (y + resid).double().requires_grad_(True)
    gd, bd = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(zr, (H,), gd, bd, 1e-5)
    torch.testing.assert_close(out.double(), ref.detach(), rtol=1e-5, atol=1e-5)
    assert (oh.double() + ol.double() - ref.detach()).abs().max().item() < 1e-4
    dy = _rand(M, H, seed=9)
    ref.backward(dy.double())
    dz, dg, db, dbias = torch.empty(M, H, device=DEV), torch.empty(H, device=DEV), torch.empty(H, device=DEV), torch.empty(H, device=DEV)
    d16 = torch.empty(M, H, dtype=BF16, device=DEV)
    ops.ln_bwd(dy, z, mean, rstd, gamma, dz, d16, None, dg, db, dbias, ops.ln_bwd_ws(M, H, DEV), M, H)
    torch.testing.assert_close(dz.double(), zr.grad, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(dg.double(), gd.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(db.double(), bd.grad, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(dbias.double(), zr.grad.sum(0), rtol=1e-4, atol=1e-4)


def test_attention2_pooled_row_mode_equals_the_dense_run():
    """nq = 1 (only query 0 of every sample is live): ctx row 0 bit-equal to the dense run; the gradient equals the
    dense run fed with dO = 0 everywhere but row 0."""
    B, S, nh = 3, 56, 12
    H = nh * 64
    qkv, hi, lo = _attn2_inputs(B, S, nh, 31)
    addmask = torch.zeros(B * S, device=DEV)
    addmask[S - 4:S] = -10000.0
    ctx_hi, ctx_lo = torch.empty(B * S, H, dtype=BF16, device=DEV), torch.empty(B * S, H, dtype=BF16, device=DEV)
    lse = torch.empty(B * nh * S, device=DEV)
    ops.attn2_fwd(hi, lo, addmask, ctx_hi, ctx_lo, lse, B, S, nh, 64, 0.1, 5)
    c_hi, c_lo = torch.empty(B, H, dtype=BF16, device=DEV), torch.empty(B, H, dtype=BF16, device=DEV)
    lse1 = torch.zeros(B * nh * S, device=DEV)
    ops.attn2_fwd(hi, lo, addmask, c_hi, c_lo, lse1, B, S, nh, 64, 0.1, 5, nq=1)
    assert torch.equal(c_hi, ctx_hi.view(B, S, H)[:, 0]) and torch.equal(c_lo, ctx_lo.view(B, S, H)[:, 0])
    assert torch.equal(lse1.view(B, nh, S)[:, :, 0], lse.view(B, nh, S)[:, :, 0])
    d0 = _rand(B, H, seed=32).to(BF16)
    dfull = torch.zeros(B, S, H, dtype=BF16, device=DEV)
    dfull[:, 0] = d0
    g_dense = torch.empty(B * S, 3 * H, dtype=BF16, device=DEV)
    ops.attn2_bwd(hi, addmask, dfull.view(B * S, H), lse, g_dense, B, S, nh, 64, 0.1, 5)
    g_c = torch.full((B * S, 3 * H), float("nan"), dtype=BF16, device=DEV)
    ops.attn2_bwd(hi, addmask, d0, lse1, g_c, B, S, nh, 64, 0.1, 5, nq=1)
    assert torch.equal(g_c, g_dense)


def test_attention2_dropout_is_consistent_between_forward_and_backward():
    B, T, V, nh, p = 2, 20, 36, 4, 0.25
    S, H = T + V, nh * 64
    qkv, hi, lo = _attn2_inputs(B, S, nh, 14, scale=1.0)
    addmask = torch.zeros(B * S, device=DEV)
    ctx_hi = torch.empty(B * S, H, dtype=BF16, device=DEV)
    ctx_lo = torch.empty_like(ctx_hi)
    lse = torch.empty(B * nh * S, device=DEV)
    q2 = qkv.clone().view(B, S, 3, nh, 64)
    q2[:, :, 2] = 1.0  # V = 1: ctx = rowsum(P * keep)
    h2, l2 = _split(q2.view(B * S, 3 * H).contiguous())
    ops.attn2_fwd(h2, l2, addmask, ctx_hi, ctx_lo, lse, B, S, nh, 64, p, 77)
    rowsum = (ctx_hi.float() + ctx_lo.float()).view(B, S, nh, 64)[..., 0]
    assert abs(rowsum.mean().item() - 1.0) < 0.05 and (rowsum - 1.0).abs().max().item() > 1e-3
    c2h, c2l = torch.empty_like(ctx_hi), torch.empty_like(ctx_hi)
    ops.attn2_fwd(h2, l2, addmask, c2h, c2l, lse, B, S, nh, 64, p, 77)
    assert torch.equal(c2h, ctx_hi)
    ops.attn2_fwd(h2, l2, addmask, c2h, c2l, lse, B, S, nh, 64, p, 78)
    assert not torch.equal(c2h, ctx_hi)
    # keep rate: V = 1 and uniform P (Q = 0) -> ctx = (#kept keys) / S / (1 - p)
    z = torch.zeros(B * S, 3 * H, device=DEV).view(B, S, 3, nh, 64)
    z[:, :, 2] = 1.0
    zh, zl = _split(z.view(B * S, 3 * H).contiguous())
    ops.attn2_fwd(zh, zl, addmask, ctx_hi, ctx_lo, lse, B, S, nh, 64, p, 123)
    kept = (ctx_hi.float() + ctx_lo.float()).view(B, S, nh, 64)[..., 0] * (1 - p)
    assert abs(kept.mean().item() - (1 - p)) < 0.01  # B*S*nh*S = 25 088 Bernoulli draws: sigma 0.003
    # backward consistency via a directional finite difference of f(qkv) = sum(ctx * w) at the hi-rounded point
    base = hi.float()
    bh, bl = _split(base)
    ops.attn2_fwd(bh, bl, addmask, ctx_hi, ctx_lo, lse, B, S, nh, 64, p, 77)
    w = _rand(B * S, H, seed=15).to(BF16)
    dqkv = torch.empty(B * S, 3 * H, dtype=BF16, device=DEV)
    ops.attn2_bwd(bh, addmask, w, lse, dqkv, B, S, nh, 64, p, 77)
    d = torch.sign(dqkv.float()) * (0.5 + torch.rand(B * S, 3 * H, generator=torch.Generator().manual_seed(16)).to(DEV))
    eps = 4e-3

    def f(x):
        xh, xl = _split(x.contiguous())
        h_, l_ = torch.empty_like(ctx_hi), torch.empty_like(ctx_hi)
        ops.attn2_fwd(xh, xl, addmask, h_, l_, lse.clone(), B, S, nh, 64, p, 77)
        return ((h_.double() + l_.double()) * w.double()).sum().item()

    fd = (f(base + eps * d) - f(base - eps * d)) / (2 * eps)
    an = (dqkv.double() * d.double()).sum().item()
    assert abs(fd - an) <= 3e-2 * max(1.0, abs(fd)), (fd, an)


def test_attention2_dropout_mask_statistics():
    """The keep mask itself, read back through one-hot V rows (S <= head dim: ctx[q, k] = P[q, k] * keep[q, k] / (1 - p) with
    uniform P): keep rate overall and per key position, independence of neighbouring keys (the two 16-bit halves of one hash
    word and consecutive hash words), of neighbouring query rows and of neighbouring seeds (seeds of consecutive layers differ by
    16 * 4096, of consecutive sites by 4096: stack.hip seed_of)."""
    B, S, nh, p = 8, 56, 12, 0.1
    H = nh * 64
    z = torch.zeros(B, S, 3, nh, 64, device=DEV)
    z[:, :, 2] = torch.eye(S, 64, device=DEV).view(1, S, 1, 64)  # V[key] = e_key
    zh, zl = _split(z.view(B * S, 3 * H).contiguous())
    addmask = torch.zeros(B * S, device=DEV)
    ctx_hi, ctx_lo = torch.empty(B * S, H, dtype=BF16, device=DEV), torch.empty(B * S, H, dtype=BF16, device=DEV)
    lse = torch.empty(B * nh * S, device=DEV)

    def mask_of(seed):
        ops.attn2_fwd(zh, zl, addmask, ctx_hi, ctx_lo, lse, B, S, nh, 64, p, seed)
        c = (ctx_hi.float() + ctx_lo.float()).view(B, S, nh, 64)[..., :S] * S * (1 - p)  # [b, q, h, k] in {0, 1}
        assert ((c - c.round()).abs().max().item() < 1e-2) and c.min().item() > -0.01 and c.max().item() < 1.01
        return c.round().permute(0, 2, 1, 3).contiguous()  # [b, h, q, k]

    seed0 = 123456789 * 4096
    m = mask_of(seed0 + 3)
    n = m.numel()
    sig = (p * (1 - p) / n) ** 0.5
    assert abs(m.mean().item() - (1 - p)) < 4 * sig, m.mean().item()
    per_key = m.mean(dim=(0, 1, 2))
    assert (per_key - (1 - p)).abs().max().item() < 5 * (p * (1 - p) / (n / S)) ** 0.5, per_key
    agree = p * p + (1 - p) * (1 - p)
    sa = (agree * (1 - agree)) ** 0.5

    def check_agree(a, b, what):
        r = (a == b).float().mean().item()
        assert abs(r - agree) < 5 * sa / a.numel() ** 0.5, (what, r, agree)

    check_agree(m[..., 0::2], m[..., 1::2], "the two halves of a hash word")
    check_agree(m[..., 1:-1:2], m[..., 2::2], "consecutive hash words")
    check_agree(m[:, :, :-1], m[:, :, 1:], "neighbouring query rows")
    check_agree(m[:, :-1], m[:, 1:], "neighbouring heads")
    for ds, what in ((1, "next seed"), (4096, "next site"), (16 * 4096, "next layer"), (4096 * 4096, "next step")):
        check_agree(m, mask_of(seed0 + 3 + ds), what)


def _blocked_ref(x, M, N):
    """XT[mb][n][mi] = X[64 mb + mi][n], zero rows past M."""
    mb = (M + 63) // 64
    pad = torch.zeros(mb * 64, N, dtype=x.dtype, device=x.device)
    pad[:M] = x[:M, :N]
    return pad.view(mb, 64, N).permute(0, 2, 1).contiguous()


@pytest.mark.parametrize("M", [64, 200, 14336, 1000])
def test_transpose_blocked_and_column_sums(M):
    xs = [_rand(M, n, seed=40 + i).to(BF16) for i, n in enumerate((768, 3072, 2304, 64))]
    wide = _rand(M, 1024, seed=50).to(BF16)
    xs.append(wide[:, 128:128 + 256])  # a column slice: ld > N
    dsts = [torch.full((ops._lib.lib().vl_blocked_elems(M, x.shape[1]),), float("nan"), dtype=BF16, device=DEV) for x in xs]
    mb = (M + 63) // 64
    cs = [None, torch.full((mb, 3072), float("nan"), device=DEV), torch.full((mb, 2304), float("nan"), device=DEV), None, None]
    ops.transpose_blocked([(x, d, c) for x, d, c in zip(xs, dsts, cs)], M)
    for x, d in zip(xs, dsts):
        assert torch.equal(d.view(mb, x.shape[1], 64), _blocked_ref(x, M, x.shape[1]))
    for x, c in zip(xs, cs):
        if c is None:
            continue
        N = x.shape[1]
        outs = [torch.full((N // 3,), float("nan"), device=DEV) for _ in range(3)]
        ops.colsum_finalize(c, mb, N, outs)
        ref = x.double().sum(0)
        got = torch.cat(outs).double()
        assert (got - ref).abs().max().item() <= 1e-5 * x.float().abs().sum(0).max().item() + 1e-4
        ops.colsum_finalize(c, mb, N, outs, accumulate=True)
        assert (torch.cat(outs).double() - 2 * ref).abs().max().item() <= 2e-5 * x.float().abs().sum(0).max().item() + 2e-4


@pytest.mark.parametrize("K", [64, 200, 1024, 14336])
def test_dw_grouped_matches_fp64(K):
    """The six weight-gradient products of a layer in one launch on blocked-transposed operands, incl. the packed
    [Q|K|V] gradient as three row sub-ranges of one image, an SFT mask, accumulation, and ragged tile edges."""
    H, I = 768, 3072
    dqkv, dt1, du, dt2 = (_rand(K, n, seed=60 + i, scale=0.5).to(BF16) for i, n in enumerate((3 * H, H, I, H)))
    x, ctx, x1, hh = (_rand(K, n, seed=70 + i).to(BF16) for i, n in enumerate((H, H, H, I)))
    odd_a, odd_b = _rand(K, 320, seed=80).to(BF16), _rand(K, 192, seed=81).to(BF16)  # 320 x 192: ragged 256-tiles
    mats = [dqkv, dt1, du, dt2, x, ctx, x1, hh]
    imgs = [torch.empty(ops._lib.lib().vl_blocked_elems(K, m.shape[1]), dtype=BF16, device=DEV) for m in mats]
    ops.transpose_blocked([(m, d, None) for m, d in zip(mats, imgs)], K)
    oimg = [torch.empty(ops._lib.lib().vl_blocked_elems(K, m.shape[1]), dtype=BF16, device=DEV) for m in (odd_a, odd_b)]
    ops.transpose_blocked([(odd_a, oimg[0], None), (odd_b, oimg[1], None)], K)
    Tqkv, Tt1, Tu, Tt2, Tx, Tctx, Tx1, Th = imgs
    mask = (torch.rand(I, H, generator=torch.Generator().manual_seed(9)) < 0.6).float().to(DEV)
    outs = dict(q=torch.full((H, H), float("nan"), device=DEV), k=torch.full((H, H), float("nan"), device=DEV),
                v=torch.full((H, H), float("nan"), device=DEV), o=torch.full((H, H), float("nan"), device=DEV),
                w1=torch.full((I, H), float("nan"), device=DEV), w2=torch.full((H, I), float("nan"), device=DEV))
    probs = [(Tqkv, 0, 3 * H, Tx, H, outs["q"], None, H, H), (Tqkv, H, 3 * H, Tx, H, outs["k"], None, H, H),
             (Tqkv, 2 * H, 3 * H, Tx, H, outs["v"], None, H, H), (Tt1, 0, H, Tctx, H, outs["o"], None, H, H),
             (Tu, 0, I, Tx1, H, outs["w1"], mask, I, H), (Tt2, 0, H, Th, I, outs["w2"], None, H, I)]
    ops.dw_grouped(probs, K)
    refs = dict(q=dqkv[:, :H].double().t() @ x.double(), k=dqkv[:, H:2 * H].double().t() @ x.double(),
                v=dqkv[:, 2 * H:].double().t() @ x.double(), o=dt1.double().t() @ ctx.double(),
                w1=(du.double().t() @ x1.double()) * mask.double(), w2=dt2.double().t() @ hh.double())
    tol = 3e-4 * math.sqrt(K / 256)
    for n in outs:
        torch.testing.assert_close(outs[n].double(), refs[n], rtol=2e-5, atol=tol, msg=lambda m, n=n: "%s: %s" % (n, m))
    first = {n: o.clone() for n, o in outs.items()}
    ops.dw_grouped(probs, K, accumulate=True)
    for n in outs:
        assert torch.equal(outs[n], first[n] + first[n]), n  # deterministic, and accumulate adds exactly
    o2 = torch.full((320, 192), float("nan"), device=DEV)
    ops.dw_grouped([(oimg[0], 0, 320, oimg[1], 192, o2, None, 320, 192)], K)
    torch.testing.assert_close(o2.double(), odd_a.double().t() @ odd_b.double(), rtol=2e-5, atol=tol)


def test_colreduce_multi_matches_fp64_and_accumulates():
    a = _rand(513, 2304, seed=90)
    b = _rand(224, 3072, seed=91)
    c = _rand(7, 64, seed=92)
    oa = [torch.full((768,), float("nan"), device=DEV) for _ in range(3)]
    ob = [torch.full((3072,), float("nan"), device=DEV)]
    oc = [torch.full((32,), float("nan"), device=DEV), None]
    ops.colreduce_multi([(a, 768, oa), (b, 3072, ob), (c, 32, oc)])
    torch.testing.assert_close(torch.cat(oa).double(), a.double().sum(0), rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(ob[0].double(), b.double().sum(0), rtol=1e-5, atol=1e-4)
    torch.testing.assert_close(oc[0].double(), c.double().sum(0)[:32], rtol=1e-5, atol=1e-5)
    first = torch.cat(oa).clone()
    ops.colreduce_multi([(a, 768, oa)], accumulate=True)
    assert torch.equal(torch.cat(oa), first + first)


@pytest.mark.parametrize("B,C", [(256, 1842), (4, 1842), (7, 300), (3, 4096)])
def test_gqa_loss_kernel_matches_the_reference_arithmetic(B, C):
    """vl_gqa_loss against the eager arithmetic of task_utils.py:413-428 + :706-711 (torch autograd for the gradient)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(B + C)
    logits = (torch.randn(B, C, generator=g) * 2.0).to(DEV)
    labels = torch.randint(0, C, (B,), generator=g)
    target = torch.zeros(B, C)
    target[torch.arange(B), labels] = 1.0
    target[0, labels[0]] = 0.5  # a fractional score: .long() truncates it to 0 -> label 0 for that row, like the reference
    dist = torch.rand(B, C, generator=g)
    dist[torch.arange(B), labels] = 0.0
    # make half of the rows "correct" so that the score is not trivially zero
    for b in range(0, B, 2):
        logits[b, labels[b]] = 50.0
    target, dist = target.to(DEV), dist.to(DEV)
    lam = 10.0
    z = logits.double().clone().requires_grad_(True)
    p10, idx = torch.topk(F.softmax(z, dim=-1), k=min(10, C))
    sem = torch.mean(torch.sum(p10 * dist.double()[torch.arange(B, device=DEV).unsqueeze(1), idx], dim=-1), dim=0)
    ref = F.cross_entropy(z, torch.argmax(target.long(), dim=1)).mean() * C + (lam * sem.mean()) * C
    ref.backward()
    oh = torch.zeros_like(target)
    oh.scatter_(1, logits.argmax(1, keepdim=True), 1)
    ref_score = (oh * target).sum() / B
    out = torch.empty(2, device=DEV)
    dl = torch.full((B, C), float("nan"), device=DEV)
    ws = torch.empty(ops._lib.lib().vl_gqa_loss_ws_bytes(B), dtype=torch.uint8, device=DEV)
    ops.gqa_loss(logits, target, dist, lam, out, dl, ws)
    assert abs(out[0].item() - ref.item()) <= 2e-6 * abs(ref.item()), (out[0].item(), ref.item())
    assert out[1].item() == ref_score.item()
    torch.testing.assert_close(dl.double(), z.grad, rtol=2e-5, atol=2e-5 * z.grad.abs().max().item())


@pytest.mark.parametrize("K,budget", [(1024, 5), (1024, 40), (2048, 16), (14336, 88), (14336, 96), (14336, 250), (448, 9)])
def test_dw_streamk_form_equals_the_tile_per_workgroup_form(K, budget):
    """vl_dw_grouped_streamk (a fixed workgroup budget; tiles cut by a share boundary are completed from partial tiles):
    against fp64; equal to one workgroup per tile up to fp32 re-association; bit-reproducible; the same bits from row-major
    and K-major operands (same segments, same order); SFT mask, accumulation and the bias column sums ride along; the
    hand-off flags are zero again after every launch."""
    H, I = 768, 3072
    dqkv, dt1, du, dt2 = (_rand(K, n, seed=160 + i, scale=0.5).to(BF16) for i, n in enumerate((3 * H, H, I, H)))
    x, ctx, x1, hh = (_rand(K, n, seed=170 + i).to(BF16) for i, n in enumerate((H, H, H, I)))
    mask = (torch.rand(I, H, generator=torch.Generator().manual_seed(19)) < 0.6).float().to(DEV)
    names = ("q", "k", "v", "o", "w1", "w2")
    shapes = dict(q=(H, H), k=(H, H), v=(H, H), o=(H, H), w1=(I, H), w2=(H, I))

    def fresh():
        outs = {n: torch.full(shapes[n], float("nan"), device=DEV) for n in names}
        cs = {n: torch.full(((shapes[n][1] + 255) // 256, shapes[n][0]), float("nan"), device=DEV) for n in ("q", "k", "v", "w1")}
        probs = [(dqkv[:, :H], None, x, None, outs["q"], None, H, H, cs["q"]), (dqkv[:, H:2 * H], None, x, None, outs["k"], None, H, H, cs["k"]),
                 (dqkv[:, 2 * H:], None, x, None, outs["v"], None, H, H, cs["v"]), (dt1, None, ctx, None, outs["o"], None, H, H, None),
                 (du, None, x1, None, outs["w1"], mask, I, H, cs["w1"]), (dt2, None, hh, None, outs["w2"], None, H, I, None)]
        return outs, cs, probs
    ws = ops.dw_streamk_ws(budget, DEV)
    flags = ws[:4096].view(torch.int32)
    o_sk, c_sk, p_sk = fresh()
    ops.dw_grouped_mixed(p_sk, K, 3, budget=budget, ws=ws)
    torch.cuda.synchronize()
    assert int(flags.abs().sum()) == 0
    o_t, c_t, p_t = fresh()
    ops.dw_grouped_mixed(p_t, K, 3)
    refs = dict(q=dqkv[:, :H].double().t() @ x.double(), k=dqkv[:, H:2 * H].double().t() @ x.double(),
                v=dqkv[:, 2 * H:].double().t() @ x.double(), o=dt1.double().t() @ ctx.double(),
                w1=(du.double().t() @ x1.double()) * mask.double(), w2=dt2.double().t() @ hh.double())
    tol = 3e-4 * math.sqrt(max(K, 256) / 256)
    for n in names:
        torch.testing.assert_close(o_sk[n].double(), refs[n], rtol=2e-5, atol=tol, msg=lambda m, n=n: "%s: %s" % (n, m))
        # fp32 re-association only: a handful of ulps of the largest partial sum
        torch.testing.assert_close(o_sk[n], o_t[n], rtol=0, atol=4e-6 * float(refs[n].abs().max()) + 1e-6)
    for n in c_sk:
        torch.testing.assert_close(c_sk[n].sum(0), c_t[n].sum(0), rtol=0, atol=2e-5 * math.sqrt(K / 64) * 8)
    # bit-reproducible, and the same bits from the K-major images (identical segments and summation order)
    o2, c2, p2 = fresh()
    ops.dw_grouped_mixed(p2, K, 3, budget=budget, ws=ws)
    for n in names:
        assert torch.equal(o_sk[n], o2[n]), n
    for n in c_sk:
        assert torch.equal(c_sk[n], c2[n]), n
    mats = [dqkv, dt1, du, dt2, x, ctx, x1, hh]
    imgs = [torch.empty(ops._lib.lib().vl_blocked_elems(K, m.shape[1]), dtype=BF16, device=DEV) for m in mats]
    ops.transpose_blocked([(m, d, None) for m, d in zip(mats, imgs)], K)
    Tqkv, Tt1, Tu, Tt2, Tx, Tctx, Tx1, Th = imgs
    o3, c3, _ = fresh()
    pk = [(Tqkv, 3 * H, Tx, H, o3["q"], None, H, H, c3["q"]), (Tqkv[64 * H:], 3 * H, Tx, H, o3["k"], None, H, H, c3["k"]),
          (Tqkv[2 * 64 * H:], 3 * H, Tx, H, o3["v"], None, H, H, c3["v"]), (Tt1, H, Tctx, H, o3["o"], None, H, H, None),
          (Tu, I, Tx1, H, o3["w1"], mask, I, H, c3["w1"]), (Tt2, H, Th, I, o3["w2"], None, H, I, None)]
    ops.dw_grouped_mixed(pk, K, 0, budget=budget, ws=ws)
    for n in names:
        assert torch.equal(o_sk[n], o3[n]), n
    # accumulation
    ops.dw_grouped_mixed(p_sk, K, 3, accumulate=True, budget=budget, ws=ws)
    for n in names:
        assert torch.equal(o_sk[n], o2[n] + o2[n]), n
    torch.cuda.synchronize()
    assert int(flags.abs().sum()) == 0


def test_gqa_loss_with_non_finite_logits_yields_nan_without_leaving_the_row():
    """A diverged step (a NaN logit, or a row of -inf) must reach the loss as NaN like the reference's eager arithmetic
    (task_utils.py:413-428) -- never as an out-of-range index: the arg-max sentinels used to be dereferenced unguarded."""
    B, C = 6, 1842
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(B, C, generator=g).to(DEV)
    logits[1, 77] = float("nan")
    logits[3, :] = float("-inf")
    logits[4, :] = float("nan")
    target = torch.zeros(B, C, device=DEV)
    target[torch.arange(B), torch.arange(B) * 3] = 1.0
    dist = torch.rand(B, C, generator=g).to(DEV)
    out = torch.zeros(2, device=DEV)
    dl = torch.zeros(B, C, device=DEV)
    ws = torch.empty(ops._lib.lib().vl_gqa_loss_ws_bytes(B), dtype=torch.uint8, device=DEV)
    ops.gqa_loss(logits, target, dist, 10.0, out, dl, ws)
    torch.cuda.synchronize()
    assert math.isnan(out[0].item())
    assert torch.isfinite(dl[[0, 2, 5]]).all()          # the healthy rows keep finite gradients
    assert torch.isnan(dl[1]).all() and torch.isnan(dl[3]).all() and torch.isnan(dl[4]).all()
    # and the healthy part is what a batch without the bad rows gives (the row terms are independent)
    good = [0, 2, 5]
    out2, dl2 = torch.zeros(2, device=DEV), torch.zeros(3, C, device=DEV)
    ops.gqa_loss(logits[good].contiguous(), target[good].contiguous(), dist[good].contiguous(), 10.0, out2, dl2, ws)
    torch.testing.assert_close(dl[good] * (B / 3.0), dl2, rtol=1e-6, atol=1e-7)


def test_adamw_per_segment_step_counts():
    """seg_step: a segment that first received a gradient later than the others is bias-corrected with ITS step count
    (pytorch_transformers.AdamW keeps state['step'] per parameter; oracle/adamw_oracle.py restates that)."""
    n = 4096
    p0, g = _rand(n, seed=61), _rand(n, seed=62)
    seg_end = torch.tensor([1024, 4096], dtype=torch.int64, device=DEV)
    seg_lr = torch.tensor([4e-5, 1e-4], device=DEV)
    seg_wd = torch.tensor([1e-4, 0.0], device=DEV)
    b1, b2, eps = 0.9, 0.999, 1e-6
    m0, v0 = _rand(n, seed=63) * 0.1, _rand(n, seed=64).abs() * 0.01
    p, m, v = p0.clone(), m0.clone(), v0.clone()
    steps = torch.tensor([7, 2], dtype=torch.int64, device=DEV)
    ops.adamw(p, g.clone(), m, v, seg_end, seg_lr, seg_wd, b1, b2, eps, 7, True, 1.0, None, 1.0, False, seg_step=steps)
    for (lo, hi, t, lr, wd) in ((0, 1024, 7, 4e-5, 1e-4), (1024, 4096, 2, 1e-4, 0.0)):
        q, mm, vv = p0[lo:hi].clone(), m0[lo:hi].clone(), v0[lo:hi].clone()
        # the same kernel with a uniform step on that slice alone is the reference for the slice
        ops.adamw(q, g[lo:hi].clone(), mm, vv, torch.tensor([hi - lo], dtype=torch.int64, device=DEV),
                  torch.tensor([lr], device=DEV), torch.tensor([wd], device=DEV), b1, b2, eps, t, True, 1.0, None, 1.0, False)
        assert torch.equal(p[lo:hi], q) and torch.equal(m[lo:hi], mm) and torch.equal(v[lo:hi], vv)
    # and a uniform seg_step array changes nothing against the scalar step
    pa, ma, va = p0.clone(), m0.clone(), v0.clone()
    pb, mb, vb = p0.clone(), m0.clone(), v0.clone()
    ops.adamw(pa, g.clone(), ma, va, seg_end, seg_lr, seg_wd, b1, b2, eps, 5, True, 1.0, None, 1.0, False)
    ops.adamw(pb, g.clone(), mb, vb, seg_end, seg_lr, seg_wd, b1, b2, eps, 5, True, 1.0, None, 1.0, False,
              seg_step=torch.tensor([5, 5], dtype=torch.int64, device=DEV))
    assert torch.equal(pa, pb)


def test_qkv_attention_entry_points_equal_the_two_step_sequence():
    """vl_qkv_attention_fwd / _bwd == projection GEMM (split epilogue) + attention, and attention backward +
    dX GEMM, bit for bit; and fp32-grade against fp64 math."""
    from clg_vqa_amd import _lib
    monkey_small = ops.SMALL_GEMM
    ops.SMALL_GEMM = False  # the entry points take no workspace: compare with the same (big-tile) kernel choice
    try:
        _qkv_attention_entry_points()
    finally:
        ops.SMALL_GEMM = monkey_small


def _qkv_attention_entry_points():
    from clg_vqa_amd import _lib
    B, S, nh, dh = 2, 56, 12, 64
    H, M = nh * dh, B * S
    x, w = _rand(M, H, seed=101), _rand(3 * H, H, seed=102, scale=0.05)
    bias = _rand(3 * H, seed=103, scale=0.1)
    xh, xl = _split(x)
    wh, wl = _split(w)
    addmask = torch.zeros(M, device=DEV)
    addmask[S - 3:S] = -10000.0
    L = _lib.lib()
    mk = lambda *s: torch.empty(*s, dtype=BF16, device=DEV)  # noqa: E731
    q_hi, q_lo, c_hi, c_lo, lse = mk(M, 3 * H), mk(M, 3 * H), mk(M, H), mk(M, H), torch.empty(B * nh * S, device=DEV)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(L.vl_qkv_attention_fwd(xh.data_ptr(), xl.data_ptr(), wh.data_ptr(), wl.data_ptr(), bias.data_ptr(),
                                      addmask.data_ptr(), q_hi.data_ptr(), q_lo.data_ptr(), c_hi.data_ptr(), c_lo.data_ptr(),
                                      lse.data_ptr(), B, S, nh, dh, S, 0.0, 1, st), "vl_qkv_attention_fwd")
    r_hi, r_lo, rc_hi, rc_lo, rlse = mk(M, 3 * H), mk(M, 3 * H), mk(M, H), mk(M, H), torch.empty_like(lse)
    ops.gemm_nt(xh, xl, wh, wl, M, 3 * H, H, 3, EPI_SPLIT, bias=bias, out_hi=r_hi, out_lo=r_lo)
    ops.attn2_fwd(r_hi, r_lo, addmask, rc_hi, rc_lo, rlse, B, S, nh, dh, 0.0, 1)
    assert torch.equal(q_hi, r_hi) and torch.equal(c_hi, rc_hi) and torch.equal(c_lo, rc_lo) and torch.equal(lse, rlse)
    qkv = x.double() @ w.double().t() + bias.double()
    ref, _ = _attn_ref(qkv, addmask.double().view(B, S), B, S, nh)
    assert ((c_hi.double() + c_lo.double()) - ref).abs().max().item() < 5e-5 * max(1.0, ref.abs().max().item())
    # backward: dctx -> dqkv -> dx = dqkv W + resid
    d16 = _rand(M, H, seed=104).to(BF16)
    wt = wh.t().contiguous()  # [H, 3H]
    resid = _rand(M, H, seed=105)
    dqkv, dx = mk(M, 3 * H), torch.empty(M, H, device=DEV)
    _lib.check(L.vl_qkv_attention_bwd(q_hi.data_ptr(), addmask.data_ptr(), d16.data_ptr(), lse.data_ptr(), wt.data_ptr(),
                                      resid.data_ptr(), dqkv.data_ptr(), dx.data_ptr(), B, S, nh, dh, S, 0.0, 1, st),
               "vl_qkv_attention_bwd")
    rdq, rdx = mk(M, 3 * H), torch.empty(M, H, device=DEV)
    ops.attn2_bwd(q_hi, addmask, d16, lse, rdq, B, S, nh, dh, 0.0, 1)
    ops.gemm_nt(rdq, None, wt, None, M, H, 3 * H, 1, EPI_F32, resid=resid, out32=rdx)
    assert torch.equal(dqkv, rdq) and torch.equal(dx, rdx)


@pytest.mark.parametrize("M,N", [(256, 768), (256, 1842), (8, 130), (3, 1536)])
def test_head_activation_kernels(M, N):
    """vl_act_fwd / vl_act_bwd: activation + dropout + operand split (forward), mask * activation' + cast + zero pad
    (backward) against torch, and the same dropout mask in both directions."""
    z = _rand(M, N, seed=60)
    dy = _rand(M, N, seed=61)
    ld = (N + 63) // 64 * 64
    acts = ((ops.ACT_RELU, torch.relu), (ops.ACT_TANH, torch.tanh), (ops.ACT_GELU, _gelu), (ops.ACT_NONE, lambda t: t))
    for act, fn in acts:
        for p in (0.0, 0.25):
            seed = 1234 + act
            ones = torch.ones(M, N, device=DEV)
            mask = torch.empty(M, N, device=DEV)
            ops.act_fwd(ones, M, N, ops.ACT_NONE, p, seed, out32=mask)  # keep-scale of every element at this seed
            if p == 0.0:
                assert torch.equal(mask, ones)
            else:
                assert set(mask.unique().tolist()) <= {0.0, 1.0 / (1.0 - p)} or \
                    ((mask == 0) | ((mask - 1.0 / (1.0 - p)).abs() < 1e-6)).all()
                if M * N > 10000:
                    assert abs((mask == 0).float().mean().item() - p) < 0.01
            out32 = torch.full((M, N), float("nan"), device=DEV)
            hi = torch.full((M, ld), float("nan"), dtype=BF16, device=DEV)
            lo = torch.full((M, ld), float("nan"), dtype=BF16, device=DEV)
            ops.act_fwd(z, M, N, act, p, seed, out32=out32, out_hi=hi, out_lo=lo)
            zz = z.clone().requires_grad_(True)
            ref = fn(zz) * mask
            torch.testing.assert_close(out32, ref.detach(), rtol=1e-5, atol=2e-6)
            eh, el = _split(out32)
            assert torch.equal(hi[:, :N], eh) and torch.equal(lo[:, :N], el)
            assert (hi[:, N:].float() == 0).all() and (lo[:, N:].float() == 0).all()
            ref.backward(dy)
            dz32 = torch.full((M, N), float("nan"), device=DEV)
            dz16 = torch.full((M, ld), float("nan"), dtype=BF16, device=DEV)
            ops.act_bwd(dy, None if act == ops.ACT_NONE else z, M, N, act, p, seed, dz32=dz32, dz16=dz16)
            torch.testing.assert_close(dz32, zz.grad, rtol=1e-5, atol=2e-6)
            assert torch.equal(dz16[:, :N], dz32.to(BF16)) and (dz16[:, N:].float() == 0).all()


@pytest.mark.parametrize("M,N,K,tile", [(1024, 3072, 768, 0), (896, 3072, 768, 5), (512, 768, 256, 3), (640, 1024, 128, 2)])
def test_gemm_epilogue_writes_the_kmajor_image_and_column_sums(M, N, K, tile):
    """VL_GX_IMG / VL_GX_COLSUM: the 16-bit epilogues store out_hi additionally as the K-major image of the weight-gradient
    GEMM (bit-equal to vl_transpose_blocked of the row-major out_hi) and, for the GELU' epilogue, the column-sum partials
    whose total is the bias gradient."""
    from clg_vqa_amd import _lib
    L = _lib.lib()
    x, w = _rand(M, K, seed=70), _rand(N, K, seed=71, scale=0.05)
    bias = _rand(N, seed=72)
    a, al = _split(x)
    b, bl = _split(w)
    img_ref = torch.empty(L.vl_blocked_elems(M, N), dtype=BF16, device=DEV)
    # forward: erf-GELU + split
    hi, lo, aux = (torch.empty(M, N, dtype=BF16, device=DEV) for _ in range(3))
    img = torch.full((L.vl_blocked_elems(M, N),), float("nan"), dtype=BF16, device=DEV)
    ops.gemm_nt(a, al, b, bl, M, N, K, 3, EPI_GELU_SPLIT, bias=bias, out_hi=hi, out_lo=lo, aux16=aux, tile=tile, image=img)
    ops.transpose_blocked([(hi, img_ref, None)], M)
    assert torch.equal(img.view(torch.int16), img_ref.view(torch.int16))
    hi0 = torch.empty_like(hi)
    ops.gemm_nt(a, al, b, bl, M, N, K, 3, EPI_GELU_SPLIT, bias=bias, out_hi=hi0, out_lo=lo, aux16=aux, tile=tile)
    assert torch.equal(hi, hi0)  # the row-major outputs do not change
    # backward: GELU' multiply + bf16, image + column sums
    du = torch.empty(M, N, dtype=BF16, device=DEV)
    cs = torch.full((8 * ((M + 255) // 256), N), float("nan"), device=DEV)
    img.fill_(float("nan"))
    rows = ops.gemm_nt(a, None, b, None, M, N, K, 1, EPI_DGELU_BF16, out_hi=du, aux16=aux, tile=tile, image=img, colsum=cs)
    assert 0 < rows <= cs.shape[0]
    ops.transpose_blocked([(du, img_ref, None)], M)
    assert torch.equal(img.view(torch.int16), img_ref.view(torch.int16))
    ref = du.double().sum(0)
    got = cs[:rows].double().sum(0)
    assert (got - ref).abs().max().item() <= 1e-5 * du.double().abs().sum(0).max().item() + 1e-6
    out = torch.empty(N, device=DEV)
    ops.colreduce_multi([(cs[:rows], N, (out,))])
    torch.testing.assert_close(out.double(), ref, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("K", [64, 448, 1024, 14336])
def test_dw_grouped_rowmajor_matches_fp64_and_the_kmajor_path(K):
    """The weight-gradient products straight from the ROW-MAJOR activations (transposing LDS reads): against fp64, against
    the K-major path (same products, same k order inside a tile: bit-equal), with the packed [Q|K|V] gradient as three
    column sub-ranges, an SFT mask, accumulation, ragged tile edges, and the bias column sums out of the same pass."""
    H, I = 768, 3072
    dqkv, dt1, du, dt2 = (_rand(K, n, seed=60 + i, scale=0.5).to(BF16) for i, n in enumerate((3 * H, H, I, H)))
    x, ctx, x1, hh = (_rand(K, n, seed=70 + i).to(BF16) for i, n in enumerate((H, H, H, I)))
    mask = (torch.rand(I, H, generator=torch.Generator().manual_seed(9)) < 0.6).float().to(DEV)
    names = ("q", "k", "v", "o", "w1", "w2")
    shapes = dict(q=(H, H), k=(H, H), v=(H, H), o=(H, H), w1=(I, H), w2=(H, I))
    outs = {n: torch.full(shapes[n], float("nan"), device=DEV) for n in names}
    cs = {n: torch.full(((shapes[n][1] + 255) // 256, shapes[n][0]), float("nan"), device=DEV) for n in ("q", "k", "v", "w1")}
    probs = [(dqkv[:, :H], x, outs["q"], None, H, H, cs["q"]), (dqkv[:, H:2 * H], x, outs["k"], None, H, H, cs["k"]),
             (dqkv[:, 2 * H:], x, outs["v"], None, H, H, cs["v"]), (dt1, ctx, outs["o"], None, H, H, None),
             (du, x1, outs["w1"], mask, I, H, cs["w1"]), (dt2, hh, outs["w2"], None, H, I, None)]
    ops.dw_grouped_rowmajor(probs, K)
    refs = dict(q=dqkv[:, :H].double().t() @ x.double(), k=dqkv[:, H:2 * H].double().t() @ x.double(),
                v=dqkv[:, 2 * H:].double().t() @ x.double(), o=dt1.double().t() @ ctx.double(),
                w1=(du.double().t() @ x1.double()) * mask.double(), w2=dt2.double().t() @ hh.double())
    tol = 3e-4 * math.sqrt(max(K, 256) / 256)
    for n in names:
        torch.testing.assert_close(outs[n].double(), refs[n], rtol=2e-5, atol=tol, msg=lambda m, n=n: "%s: %s" % (n, m))
    bias_ref = dict(q=dqkv[:, :H], k=dqkv[:, H:2 * H], v=dqkv[:, 2 * H:], w1=du)
    for n, c in cs.items():
        torch.testing.assert_close(c.double().sum(0), bias_ref[n].double().sum(0), rtol=1e-5, atol=1e-3 * math.sqrt(K / 64))
    # the K-major path adds the same products in the same order
    mats = [dqkv, dt1, du, dt2, x, ctx, x1, hh]
    imgs = [torch.empty(ops._lib.lib().vl_blocked_elems(K, m.shape[1]), dtype=BF16, device=DEV) for m in mats]
    ops.transpose_blocked([(m, d, None) for m, d in zip(mats, imgs)], K)
    Tqkv, Tt1, Tu, Tt2, Tx, Tctx, Tx1, Th = imgs
    o2 = {n: torch.full(shapes[n], float("nan"), device=DEV) for n in names}
    ops.dw_grouped([(Tqkv, 0, 3 * H, Tx, H, o2["q"], None, H, H), (Tqkv, H, 3 * H, Tx, H, o2["k"], None, H, H),
                    (Tqkv, 2 * H, 3 * H, Tx, H, o2["v"], None, H, H), (Tt1, 0, H, Tctx, H, o2["o"], None, H, H),
                    (Tu, 0, I, Tx1, H, o2["w1"], mask, I, H), (Tt2, 0, H, Th, I, o2["w2"], None, H, I)], K)
    for n in names:
        assert torch.equal(outs[n], o2[n]), n
    # mixed layouts (one side row-major, the other its K-major image): the same sums again, bias partials included
    for mode in (1, 2):
        o3 = {n: torch.full(shapes[n], float("nan"), device=DEV) for n in names}
        c3 = {n: torch.full_like(c, float("nan")) for n, c in cs.items()}
        A = {n: (v[0], v[1]) for n, v in dict(q=(dqkv[:, :H], Tqkv), k=(dqkv[:, H:2 * H], Tqkv[64 * H:]),
                                             v=(dqkv[:, 2 * H:], Tqkv[2 * 64 * H:]), o=(dt1, Tt1), w1=(du, Tu), w2=(dt2, Tt2)).items()}
        Bm = dict(q=(x, Tx), k=(x, Tx), v=(x, Tx), o=(ctx, Tctx), w1=(x1, Tx1), w2=(hh, Th))
        acols = dict(q=3 * H, k=3 * H, v=3 * H, o=H, w1=I, w2=H)
        bcols = dict(q=H, k=H, v=H, o=H, w1=H, w2=I)
        pm = []
        for n in names:
            a_t, a_ld = (A[n][0], None) if mode & 1 else (A[n][1], acols[n])
            b_t, b_ld = (Bm[n][0], None) if mode & 2 else (Bm[n][1], bcols[n])
            pm.append((a_t, a_ld, b_t, b_ld, o3[n], mask if n == "w1" else None, shapes[n][0], shapes[n][1], c3.get(n)))
        ops.dw_grouped_mixed(pm, K, mode)
        for n in names:
            assert torch.equal(outs[n], o3[n]), (mode, n)
        for n in cs:
            assert torch.equal(cs[n], c3[n]), (mode, n)
    first = {n: o.clone() for n, o in outs.items()}
    ops.dw_grouped_rowmajor(probs, K, accumulate=True)
    for n in names:
        assert torch.equal(outs[n], first[n] + first[n]), n
    odd_a, odd_b = _rand(K, 320, seed=80).to(BF16), _rand(K, 200, seed=81).to(BF16)  # ragged 256-tiles, N % 8 == 0
    o3 = torch.full((320, 200), float("nan"), device=DEV)
    c3 = torch.full((1, 320), float("nan"), device=DEV)
    ops.dw_grouped_rowmajor([(odd_a, odd_b, o3, None, 320, 200, c3)], K)
    torch.testing.assert_close(o3.double(), odd_a.double().t() @ odd_b.double(), rtol=2e-5, atol=tol)
    torch.testing.assert_close(c3[0].double(), odd_a.double().sum(0), rtol=1e-5, atol=1e-3 * math.sqrt(K / 64))

import torch, sys
sys.path.insert(0, ".")
from clg_vqa_amd import ops
from clg_vqa_amd.ops import BF16, EPI_GELU_SPLIT, EPI_DGELU_BF16
DEV="cuda"
def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)
def _split(x):
    hi = torch.empty_like(x, dtype=BF16); lo = torch.empty_like(x, dtype=BF16)
    ops.split_f32(x.contiguous(), hi, lo); return hi, lo
M,N,K=448,512,256
x, w, bias = _rand(M, K, seed=70), _rand(N, K, seed=71, scale=0.1), _rand(N, seed=72)
xh, xl = _split(x); wh, wl = _split(w)
outs={}
for width in (2,5):
    ops.GEMM_TILE = width
    u16 = torch.full((M, N), float("nan"), dtype=BF16, device=DEV)
    hh, hl, dh = torch.full_like(u16, float("nan")), torch.full_like(u16, float("nan")), torch.full_like(u16, float("nan"))
    ops.gemm_nt(xh, xl, wh, wl, M, N, K, 3, EPI_GELU_SPLIT, bias=bias, out_hi=hh, out_lo=hl, aux16=u16)
    aux = _rand(M, N, seed=73).to(BF16)
    ops.gemm_nt(xh, None, wh, None, M, N, K, 1, EPI_DGELU_BF16, out_hi=dh, aux16=aux)
    outs[width]=(u16,hh,hl,dh)
for name,a,b in zip(("u16","hh","hl","dh"),outs[2],outs[5]):
    d=(a.view(torch.int16)!=b.view(torch.int16))
    print(name, int(d.sum()), "rows", d.any(1).nonzero().flatten()[:10].tolist(), "cols", d.any(0).nonzero().flatten()[:10].tolist())
    if d.any():
        i=d.nonzero()[0]; print("  first", i.tolist(), a[i[0],i[1]].item(), b[i[0],i[1]].item())

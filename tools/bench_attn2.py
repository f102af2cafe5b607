"""The text cross-attention of a 64x64-level transformer block (M = 196 608 rows, C = 320, 77 keys): ONE fused launch (fused.hip) against the
three launches it replaces (LayerNorm-folded q projection, attn_text_kernel, out projection + residual + statistics)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from univst_amd import _native as nat


def t(f, it=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


M, C, heads, T, rpb = 196608, 320, 8, 77, 65536
g = torch.Generator().manual_seed(0)
x = torch.randn(M, C, generator=g).half().cuda()
xf = x.float()
stats = torch.stack([xf.view(M, 2, 160).sum(-1), (xf * xf).view(M, 2, 160).sum(-1)], -1).contiguous()
del xf
wq = (torch.randn(C, C, generator=g) / math.sqrt(C) * math.log2(math.e) / math.sqrt(40)).half().cuda()
wo = (torch.randn(C, C, generator=g) / math.sqrt(C)).half().cuda()
bo = (0.1 * torch.randn(C, generator=g)).half().cuda()
kv = torch.randn(3 * T, 2 * C, generator=g).half().cuda()
wsum, lnb = wq.float().sum(1).contiguous(), torch.zeros(C, device="cuda")
wqf, wof = nat.frag_pack(wq), nat.frag_pack(wo)
out = torch.empty(M, C, device="cuda", dtype=torch.float16)
st2 = torch.empty(M, 2, 2, device="cuda")
q = torch.empty(M, C, device="cuda", dtype=torch.float16)
idx = torch.tensor([[0], [1], [2]], dtype=torch.int32).cuda()
kvb = kv.view(3, T, 2 * C)
fused = lambda: nat.attn2_fused(x, wqf, kv, wof, bo, rpb, heads, ln=(stats, wsum, lnb), q_prescaled=True, stats_out=st2, out=out)
ms_f = t(fused)
ms_q = t(lambda: nat.linear_ln(x, wq, out=q, ln=(stats, wsum, lnb)))
o = nat.attention(q.view(3, rpb, C), kvb[..., :C], kvb[..., C:], idx, heads, ldkv=2 * C, Nkv=T, C_=C, q_prescaled=True)
ms_a = t(lambda: nat.attention(q.view(3, rpb, C), kvb[..., :C], kvb[..., C:], idx, heads, ldkv=2 * C, Nkv=T, C_=C, q_prescaled=True))
ms_o = t(lambda: nat.linear_ln(o.view(M, C), wo, bias=bo, residual=x, out=out, stats_out=st2))
fl = 4.0 * M * C * C + 4.0 * M * T * C
print(f"fused attn2: {ms_f:.3f} ms ({fl / ms_f / 1e9:.0f} TF, {2.0 * 2 * M * C / ms_f / 1e6:.0f} GB/s of X + Y) | three launches: q {ms_q:.3f} + attn {ms_a:.3f} + out {ms_o:.3f} = {ms_q + ms_a + ms_o:.3f} ms")

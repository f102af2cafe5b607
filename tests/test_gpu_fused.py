"""GPU parity of the round-5 fused operators (univst_amd/csrc/fused.hip), through the C ABI, against plain torch fp32 restatements of the
ops they replace (attention.py:321-327 = norm2 -> attn2 (diffusers Attention over 77 text keys) -> + hidden_states) on the same fp16 inputs.

Tolerances: fp16 storage, fp32 accumulation; the fused kernel rounds Q, P, O to fp16 where the unfused graph stores them and Y once before
the residual add (the reference's own order): max error <= 3e-3 of the output scale, relative RMS <= 1e-3.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    from univst_amd import _native
    _native.load()
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return _native


def _frag_ref(w):
    """univst_frag_pack, written out: [N/16][K/32][64 lanes][8] with lane (l15, g) = W[nf*16 + l15][ks*32 + g*8 .. +8]"""
    N, K = w.shape
    return w.view(N // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(N, K)


def test_frag_pack_matches_its_documented_order(nat):
    w = torch.arange(320 * 320, dtype=torch.float32).remainder(2039).half().cuda().view(320, 320)
    assert torch.equal(nat.frag_pack(w), _frag_ref(w))
    with pytest.raises(RuntimeError, match="multiple of 16"):
        nat.frag_pack(torch.zeros(24, 32).half().cuda())


def _attn2_ref(xn, wq, kv, wo, bo, res, rows_per_branch, heads):
    """fp32: q = xn Wq^T -> per branch softmax(q k^T / sqrt(d)) v -> Wo^T + bias + residual (q, o rounded to fp16 as the graph stores them)"""
    M, C = xn.shape
    d = C // heads
    B = -(-M // rows_per_branch)
    T = kv.shape[0] // B
    q = (xn @ wq.float().t()).half().float()
    out = torch.empty(M, C, device=xn.device)
    for b in range(B):
        r0, r1 = b * rows_per_branch, min((b + 1) * rows_per_branch, M)
        k = kv[b * T:(b + 1) * T, :C].float().view(T, heads, d).transpose(0, 1)
        v = kv[b * T:(b + 1) * T, C:].float().view(T, heads, d).transpose(0, 1)
        qb = q[r0:r1].view(-1, heads, d).transpose(0, 1)
        o = F.scaled_dot_product_attention(qb[None], k[None], v[None])[0].transpose(0, 1).reshape(r1 - r0, C)
        out[r0:r1] = o.half().float()
    return out @ wo.float().t() + bo.float() + res.float()


@pytest.mark.parametrize("M,rpb,T,ln,pre,row_mean", [(3 * 4096, 4096, 77, True, True, 0.7),       # three branches of one 64x64 frame, the graph's configuration
                                                     (2 * 1024 + 37, 1024, 77, True, False, 0.7),    # ragged M: the last block is partly empty
                                                     (3 * 4096, 4096, 77, False, True, 0.0),         # no LayerNorm fold (ln_fold = 0): normalised rows in, residual apart
                                                     (2 * 640, 640, 80, True, True, 30.0),           # all 80 key slots in use; row mean = 30 std
                                                     (192, 64, 5, True, False, 0.7)])                # three one-block branches, a handful of keys
def test_attn2_fused_matches_torch(nat, M, rpb, T, ln, pre, row_mean):
    C, heads = 320, 8
    d = C // heads
    g = torch.Generator().manual_seed(M + T)
    B = -(-M // rpb)
    x0 = torch.randn(M, C, generator=g).half().cuda()
    wp = (torch.randn(C, C, generator=g) / math.sqrt(C)).half().cuda()
    bp = (row_mean + 0.3 * torch.randn(C, generator=g)).half().cuda()
    stats = torch.full((M, C // 160, 2), float("nan"), device="cuda", dtype=torch.float32)
    x = nat.linear_ln(x0, wp, bias=bp, stats_out=stats) if M >= 38400 else None
    if x is None:       # small M: the producer linear takes a tile without the statistics epilogue — make them here, as documented
        x = nat.linear(x0, wp, bias=bp)
        xf = x.float()
        stats = torch.stack([xf.view(M, C // 160, 160).sum(-1), (xf * xf).view(M, C // 160, 160).sum(-1)], -1).contiguous()
    gamma = (1.0 + 0.3 * torch.randn(C, generator=g)).half().cuda()
    beta = (0.2 * torch.randn(C, generator=g)).half().cuda()
    wq = (torch.randn(C, C, generator=g) / math.sqrt(C)).half().cuda()
    wo = (torch.randn(C, C, generator=g) / math.sqrt(C)).half().cuda()
    bo = (0.1 * torch.randn(C, generator=g)).half().cuda()
    kv = torch.randn(B * T, 2 * C, generator=g).half().cuda()
    xn = F.layer_norm(x.float(), (C,), gamma.float(), beta.float(), 1e-5)
    sc = math.log2(math.e) / math.sqrt(d) if pre else 1.0
    wqs = (wq.float() * sc).half()                                 # what finalize() stores as "#qs"
    stats_out = torch.full((M, C // 160, 2), float("nan"), device="cuda", dtype=torch.float32)
    if ln:
        wl = (wqs.float() * gamma.float()[None]).half()
        wsum = wl.float().sum(1).contiguous()
        lnb = (wqs.float() @ beta.float()).contiguous()
        got = nat.attn2_fused(x, nat.frag_pack(wl), kv, nat.frag_pack(wo), bo, rpb, heads, ln=(stats, wsum, lnb), q_prescaled=pre, stats_out=stats_out)
        want = _attn2_ref(xn, wq, kv, wo, bo, x, rpb, heads)
    else:
        xnh = xn.half()
        got = nat.attn2_fused(xnh, nat.frag_pack(wqs), kv, nat.frag_pack(wo), bo, rpb, heads, residual=x, q_prescaled=pre, stats_out=stats_out)
        want = _attn2_ref(xnh.float(), wq, kv, wo, bo, x, rpb, heads)
    gf = got.float()
    scale = want.abs().max().item()
    mx = (gf - want).abs().max().item() / scale
    rms = ((gf - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()
    assert mx < 3e-3 and rms < 1e-3, (mx, rms)
    want_st = torch.stack([gf.view(M, C // 160, 160).sum(-1), (gf * gf).view(M, C // 160, 160).sum(-1)], -1)
    assert torch.allclose(stats_out, want_st, rtol=2e-5, atol=2e-3), (stats_out - want_st).abs().max().item()


def test_attn2_fused_equals_the_three_launch_path(nat):
    """the same block through univst_linear_ln (q) + univst_attention (77 keys) + univst_linear (out + residual): both paths round Q and O to
    fp16 at the same places, so they agree to the last rounding of Y (<= 2 fp16 ulp of the output scale)"""
    M, C, heads, T, rpb = 2 * 4096, 320, 8, 77, 4096
    d = C // heads
    g = torch.Generator().manual_seed(5)
    x = torch.randn(M, C, generator=g).half().cuda()
    wq = (torch.randn(C, C, generator=g) / math.sqrt(C) * math.log2(math.e) / math.sqrt(d)).half().cuda()
    wo = (torch.randn(C, C, generator=g) / math.sqrt(C)).half().cuda()
    bo = (0.1 * torch.randn(C, generator=g)).half().cuda()
    kv = torch.randn(2 * T, 2 * C, generator=g).half().cuda()
    q = nat.linear(x, wq)
    idx = torch.tensor([[0], [1]], dtype=torch.int32).cuda()
    kvb = kv.view(2, T, 2 * C)
    o = nat.attention(q.view(2, rpb, C), kvb[..., :C], kvb[..., C:], idx, heads, ldkv=2 * C, Nkv=T, C_=C, q_prescaled=True).view(M, C)
    y3 = nat.linear(o, wo, bias=bo, residual=x)
    yf = nat.attn2_fused(x, nat.frag_pack(wq), kv, nat.frag_pack(wo), bo, rpb, heads, q_prescaled=True)
    err = (yf.float() - y3.float()).abs().max().item() / y3.float().abs().max().item()
    assert err < 1.5e-3, err


def test_attn2_fused_rejects_other_shapes(nat):
    x = torch.zeros(128, 640).half().cuda()
    w = torch.zeros(640, 640).half().cuda()
    with pytest.raises(RuntimeError, match="not a shape this kernel serves"):
        nat.attn2_fused(x, w, torch.zeros(77, 1280).half().cuda(), w, None, 128, 8)
    x = torch.zeros(96, 320).half().cuda()
    w = torch.zeros(320, 320).half().cuda()
    with pytest.raises(RuntimeError, match="not a shape this kernel serves"):
        nat.attn2_fused(x, w, torch.zeros(2 * 77, 640).half().cuda(), w, None, 48, 8)       # a block would straddle two branches


# ---------------------------------------------------------------------------------------------------------------------------------
# the transformer block's per-frame GroupNorm folded into proj_in (attention.py:121-123)

@pytest.mark.parametrize("frames,N,mean", [(12, 4096, 0.5), (12, 4096, 25.0), (64, 768, 0.5)])        # 256-row tiles inside a frame; 768 rows per frame: 192-row tiles
def test_groupnorm_folded_into_linear(nat, frames, N, mean):
    """univst_groupnorm_fold_linear + univst_linear_sets against torch fp32 GroupNorm(32, eps 1e-6, per frame) -> linear on the same fp16 inputs; the
    second case has group means of 25 standard deviations (the folded bias is fp32: an fp16 one would lose 1 % of the output scale there)."""
    C = 320
    g = torch.Generator().manual_seed(frames + N)
    x = (torch.randn(frames * N, C, generator=g) * (1.0 + torch.rand(C, generator=g)) + mean * torch.randn(C, generator=g)).half().cuda()
    gamma = (1.0 + 0.3 * torch.randn(C, generator=g)).half().cuda()
    beta = (0.2 * torch.randn(C, generator=g)).half().cuda()
    w = (torch.randn(C, C, generator=g) / math.sqrt(C)).half().cuda()
    b = (0.1 * torch.randn(C, generator=g)).half().cuda()
    wsets, b32 = nat.groupnorm_fold_linear(x, gamma, beta, 32, 1e-6, N, w, b)
    stats = torch.full((frames * N, 2, 2), float("nan"), device="cuda")
    got = nat.linear_sets(x, wsets, b32, N, stats_out=stats).float()
    xn = F.group_norm(x.float().view(frames, N, C).transpose(1, 2), 32, gamma.float(), beta.float(), 1e-6).transpose(1, 2).reshape(frames * N, C)
    ref = xn @ w.float().t() + b.float()
    scale = ref.abs().max().item()
    mx = (got - ref).abs().max().item() / scale
    rms = ((got - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()
    assert mx < 3e-3 and rms < 1e-3, (mx, rms)
    two = nat.linear(nat.groupnorm_nhwc(x.view(frames, N, 1, C), gamma, beta, 32, 1e-6, N).view(frames * N, C), w, bias=b).float()
    assert (got - two).abs().max().item() / scale < 3e-3
    want_st = torch.stack([got.view(-1, 2, 160).sum(-1), (got * got).view(-1, 2, 160).sum(-1)], -1)
    assert torch.allclose(stats, want_st, rtol=2e-5, atol=2e-3)


def test_linear_sets_refuses_what_the_direct_path_does_not_take(nat):
    x = torch.zeros(1536, 320).half().cuda()
    with pytest.raises(RuntimeError, match="direct 256x320 path"):
        nat.linear_sets(x, torch.zeros(2, 320, 320).half().cuda(), torch.zeros(2, 320, device="cuda"), 768)


@pytest.mark.parametrize("M,rpb,row_mean", [(3 * 4096, 4096, 0.7), (2 * 1024 + 37, 1024, 0.7), (2 * 640, 640, 30.0)])
def test_attn12_fused_matches_the_two_step_path_and_torch(nat, M, rpb, row_mean):
    """univst_attn12_fused (attn1.to_out + residual in front of the fused text cross-attention; H2 stays in LDS, LayerNorm statistics taken inside) against
    torch fp32 and against univst_linear + univst_attn2_fused (which differ only in where the statistics come from)"""
    C, heads, T = 320, 8, 77
    d = C // heads
    g = torch.Generator().manual_seed(M)
    B = -(-M // rpb)
    ao = torch.randn(M, C, generator=g).half().cuda()
    hin = (torch.randn(M, C, generator=g) + row_mean).half().cuda()
    wp = (torch.randn(C, C, generator=g) / math.sqrt(C)).half().cuda()
    bp = (0.3 * torch.randn(C, generator=g)).half().cuda()
    gamma = (1.0 + 0.3 * torch.randn(C, generator=g)).half().cuda()
    beta = (0.2 * torch.randn(C, generator=g)).half().cuda()
    wq = (torch.randn(C, C, generator=g) / math.sqrt(C)).half().cuda()
    wo = (torch.randn(C, C, generator=g) / math.sqrt(C)).half().cuda()
    bo = (0.1 * torch.randn(C, generator=g)).half().cuda()
    kv = torch.randn(B * T, 2 * C, generator=g).half().cuda()
    wqs = (wq.float() * (math.log2(math.e) / math.sqrt(d))).half()
    wl = (wqs.float() * gamma.float()[None]).half()
    wsum, lnb = wl.float().sum(1).contiguous(), (wqs.float() @ beta.float()).contiguous()
    st = torch.full((M, 2, 2), float("nan"), device="cuda")
    got = nat.attn12_fused(ao, nat.frag_pack(wp), bp, hin, nat.frag_pack(wl), wsum, lnb, kv, nat.frag_pack(wo), bo, rpb, heads, q_prescaled=True, stats_out=st).float()
    h2 = (ao.float() @ wp.float().t() + bp.float() + hin.float()).half()
    xn = F.layer_norm(h2.float(), (C,), gamma.float(), beta.float(), 1e-5)
    want = _attn2_ref(xn, wq, kv, wo, bo, h2, rpb, heads)
    scale = want.abs().max().item()
    mx = (got - want).abs().max().item() / scale
    rms = ((got - want).pow(2).mean().sqrt() / want.pow(2).mean().sqrt()).item()
    assert mx < 3e-3 and rms < 1e-3, (mx, rms)
    h2n = nat.linear(ao, wp, bias=bp, residual=hin)
    h2f = h2n.float()
    stats = torch.stack([h2f.view(M, 2, 160).sum(-1), (h2f * h2f).view(M, 2, 160).sum(-1)], -1).contiguous()
    two = nat.attn2_fused(h2n, nat.frag_pack(wl), kv, nat.frag_pack(wo), bo, rpb, heads, ln=(stats, wsum, lnb), q_prescaled=True).float()
    assert (got - two).abs().max().item() / scale < 1.5e-3
    want_st = torch.stack([got.view(M, 2, 160).sum(-1), (got * got).view(M, 2, 160).sum(-1)], -1)
    assert torch.allclose(st, want_st, rtol=2e-5, atol=2e-3)

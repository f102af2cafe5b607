"""First vertical slice of the SD3 / SD3.5 rectified-flow path (BASELINE config 5), through the C ABI: the
reference-owned pieces — CrossFrameProcessor / AttentionShiftProcessor (backbones/video_diffusion_sd3/pnp_utils.py:9-271) against
golden G18 (the reference's own __call__ at head_dim 64, 3 branches x 16 frames, 24 image + 7 text tokens, inside and outside the
shift window) and against the G16/G18-pinned oracle at multi-tile token counts, the plugin's attention_adain, rf_inversion / rf_solver (inversion_tools/flow_inversion.py:123-264) against golden
G17 — plus the adaLN-modulate and per-head RMSNorm operators against the (parity-unpinned) restatement of diffusers'
JointTransformerBlock.  Tolerances: fp16 storage of q/k/v and of the projections: 4e-3 of the output scale (max), 1e-3 rms.
Second half: the MM-DiT backbone mirror (models/transformer_3D_model.py) against the parity-unpinned restatement of diffusers'
SD3Transformer2DModel."""
import os
import types

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sd3_ref  # noqa: E402


@pytest.fixture(scope="module")
def nat():
    from univst_amd import _native
    _native.load()
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    return _native


def errs(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    e = got - ref
    return e.abs().max().item() / ref.abs().max().item(), (e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()


class _Attn(torch.nn.Module):
    """what the processors read from diffusers' Attention: `.heads`, the parameters (through state_dict) and norm_q.eps."""

    def __init__(self, params, heads):
        super().__init__()
        self.heads = heads
        self._p = {k: v.clone() for k, v in params.items()}
        self.norm_q = types.SimpleNamespace(eps=1e-6)

    def state_dict(self, *a, **k):
        return self._p


def test_processors_match_reference_golden_g18(golden):
    from univst_amd.backbones.video_diffusion_sd3.pnp_utils import CrossFrameProcessor, AttentionShiftProcessor
    g = golden("g18_sd3_processors_hd64")
    # both sides start from the same fp16-rounded parameters and activations (the golden was produced in fp32 from fp32 values;
    # the oracle — pinned to it at 4e-7 — is re-evaluated on the rounded values so that only the kernels' arithmetic is compared)
    P = {k: v.half().float() for k, v in g["params"].items()}
    hid, enc = g["hidden"].half(), g["enc"].half()
    attn = _Attn(P, 2)
    img, txt = CrossFrameProcessor()(attn, hid.cuda(), enc.cuda())
    r_img, r_txt = sd3_ref.joint_attention(P, 2, hid.float(), enc.float())
    for got, ref, gold in ((img, r_img, g["cross_frame"]["img"]), (txt, r_txt, g["cross_frame"]["txt"])):
        mx, rms = errs(got, ref)
        assert mx < 4e-3 and rms < 1e-3, (mx, rms)
        assert errs(got, gold)[0] < 2e-2          # and it is the reference's own output up to the fp16 rounding of the inputs
    only = CrossFrameProcessor()(attn, hid.cuda())
    mx, rms = errs(only, sd3_ref.joint_attention(P, 2, hid.float(), None))
    assert mx < 4e-3 and rms < 1e-3, (mx, rms)
    assert errs(only, g["cross_frame_no_text"])[0] < 2e-2
    plain = img
    for idx in (0, 17, 30, 31):
        proc = AttentionShiftProcessor(0.0, 0.6)
        s_img, s_txt = proc(attn, hid.cuda(), enc.cuda(), idx=idx)
        r_img, r_txt = sd3_ref.joint_attention(P, 2, hid.float(), enc.float(), idx=idx, shift=True, eta1=0.0, eta2=0.6)
        mx, rms = errs(s_img, r_img)
        mxt, rmst = errs(s_txt, r_txt)
        assert mx < 4e-3 and rms < 1e-3 and mxt < 4e-3 and rmst < 1e-3, (idx, mx, rms, mxt, rmst)
        assert errs(s_img, g[f"shift_idx{idx}"]["img"])[0] < 2e-2 and errs(s_txt, g[f"shift_idx{idx}"]["txt"])[0] < 2e-2
        if idx <= 30:       # inside the window the stylised branch moves, the other two never do
            assert torch.equal(s_img[:32], plain[:32]) and not torch.equal(s_img[32:], plain[32:])
        else:
            assert torch.equal(s_img, plain)


def test_joint_attention_multi_tile_vs_oracle():
    """token counts that span several 64-key tiles and end inside one (200 image tokens, 77 text tokens: four tiles per image
    source with a 8-key tail, two text tiles with a 13-key tail), inside the shift window — native processors vs the oracle
    (pinned to the reference's __call__ by G16 / G18)."""
    from univst_amd.backbones.video_diffusion_sd3.pnp_utils import AttentionShiftProcessor
    g = torch.Generator().manual_seed(99)
    C, heads, N, Nt = 128, 2, 200, 77
    P = {}
    for nm in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"):
        P[nm + ".weight"] = (torch.randn(C, C, generator=g) / C ** 0.5).half().float()
        P[nm + ".bias"] = (0.1 * torch.randn(C, generator=g)).half().float()
    for nm in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
        P[nm + ".weight"] = (1.0 + 0.2 * torch.randn(64, generator=g)).half().float()
    hid = torch.randn(48, N, C, generator=g).half()
    hid[32:] = hid[32:] * 1.4 - 0.2
    enc = torch.randn(48, Nt, C, generator=g).half()
    img, txt = AttentionShiftProcessor(0.0, 0.6)(_Attn(P, heads), hid.cuda(), enc.cuda(), idx=12)
    r_img, r_txt = sd3_ref.joint_attention(P, heads, hid.float(), enc.float(), idx=12, shift=True, eta1=0.0, eta2=0.6)
    mx, rms = errs(img, r_img)
    mxt, rmst = errs(txt, r_txt)
    assert mx < 4e-3 and rms < 1e-3 and mxt < 4e-3 and rmst < 1e-3, (mx, rms, mxt, rmst)


def test_attention_adain_helper_matches_reference_golden(golden):
    from univst_amd.backbones.video_diffusion_sd3 import pnp_utils as sd3
    g = golden("g18_sd3_processors_hd64")["attention_adain"]
    cnt, sty = g["cnt"].half(), g["sty"].half()
    got = sd3.attention_adain(cnt.cuda(), sty.cuda())
    mx, rms = errs(got, sd3_ref.attention_adain(cnt.float(), sty.float()))
    assert mx < 3e-3 and rms < 1e-3, (mx, rms)
    assert errs(got, g["out"])[0] < 1e-2
    lat = sd3.latent_adain(torch.randn(16, 4, 6, 6).half().cuda(), (0.3 + 0.6 * torch.randn(16, 4, 6, 6)).half().cuda())
    assert lat.shape == (16, 4, 6, 6) and torch.isfinite(lat.float()).all()


def test_rectified_flow_inversions_match_reference_golden_g17(golden, tmp_path):
    from univst_amd.inversion_tools import flow_inversion as fi
    g = golden("g17_sd3_rf")
    sig, z0 = g["sigmas"], g["z0"]

    def vel(x, t1000, idx):          # the closed-form velocity field the golden generator used as the transformer
        tt = (t1000 / 1000.0).reshape(-1)[0]
        return torch.tanh(0.7 * x.flip(-1)) * (0.5 + tt) - 0.3 * x + 0.05 * idx

    class Bar:
        def __enter__(self): return self
        def __exit__(self, *a): return False
        def update(self): pass

    class Pipe:
        device = "cuda"
        scheduler = types.SimpleNamespace(sigmas=sig, set_timesteps=lambda n, device=None: None)
        def encode_prompt(self, prompt, prompt_2, prompt_3): return (torch.zeros(1, 4, 8), None, torch.zeros(1, 8), None)
        def progress_bar(self, total=None): return Bar()
        def transformer(self, hidden_states, timestep, encoder_hidden_states, pooled_projections, idx=0, ft_indices=None, ft_timesteps=None,
                        ft_path=None, return_dict=False):
            return (vel(hidden_states.float(), timestep.float(), idx),)
    torch.manual_seed(1701)          # rf_inversion draws its target noise with torch.randn_like at the same place as the reference
    zT = fi.rf_inversion(Pipe(), z0.clone(), gamma=0.5, num_inference_steps=10, inversion_path=str(tmp_path))
    mx, rms = errs(zT, g["rf_inversion"]["final"])
    assert mx < 5e-3 and rms < 2e-3, (mx, rms)          # ten fp16-stored steps
    assert (tmp_path / "ddim_latents_0.pt").exists() and (tmp_path / "ddim_latents_10.pt").exists()
    zS = fi.rf_solver(Pipe(), z0.clone(), num_inference_steps=10)
    mx, rms = errs(zS, g["rf_solver"]["final"])
    assert mx < 5e-3 and rms < 2e-3, (mx, rms)


def test_adaln_modulate_and_rms_norm_vs_block_restatement(nat):
    """the two normalisation operators an MM-DiT block needs around the joint attention, against the (parity-unpinned) restatement
    of diffusers' JointTransformerBlock pieces: AdaLayerNormZero's modulate and the per-head RMSNorm of q / k."""
    g = torch.Generator().manual_seed(7)
    B, N, C, heads = 6, 50, 128, 2
    x = (torch.randn(B, N, C, generator=g) * 1.7 + 0.4).half()
    temb = torch.randn(B, C, generator=g).half()
    w = (torch.randn(6 * C, C, generator=g) / C ** 0.5).half()
    b = (0.1 * torch.randn(6 * C, generator=g)).half()
    ref, *_ = sd3_ref.ada_layer_norm_zero(x.float(), temb.float(), w.float(), b.float())
    emb = torch.nn.functional.linear(torch.nn.functional.silu(temb.float()), w.float(), b.float())
    shift_msa, scale_msa = emb[:, :C].half(), emb[:, C:2 * C].half()
    got = nat.adaln_modulate(x.cuda(), scale_msa.cuda(), shift_msa.cuda())
    ref16 = torch.nn.functional.layer_norm(x.float(), (C,), None, None, 1e-6) * (1 + scale_msa.float()[:, None]) + shift_msa.float()[:, None]
    mx, rms = errs(got, ref16)
    assert mx < 2e-3 and rms < 5e-4, (mx, rms)
    assert errs(got, ref)[0] < 5e-3
    q = (torch.randn(B * N, heads * 64, generator=g) * 2).half()
    wq = (1.0 + 0.2 * torch.randn(64, generator=g)).half()
    want = sd3_ref._rms(q.float().view(B * N, heads, 64), wq.float(), 1e-6).reshape(B * N, heads * 64)
    got = nat.rmsnorm_heads_(q.clone().cuda(), heads, wq.cuda())
    mx, rms = errs(got, want)
    assert mx < 2e-3 and rms < 5e-4, (mx, rms)
    # the block restatement itself runs end to end (shape / finiteness; parity unpinned: diffusers is on neither box)
    P = {"norm1.linear.weight": w.float(), "norm1.linear.bias": b.float(), "norm1_context.linear.weight": w.float(), "norm1_context.linear.bias": b.float()}
    for pre in ("ff", "ff_context"):
        P[pre + ".net.0.proj.weight"] = torch.randn(4 * C, C, generator=g) / C ** 0.5
        P[pre + ".net.0.proj.bias"] = torch.zeros(4 * C)
        P[pre + ".net.2.weight"] = torch.randn(C, 4 * C, generator=g) / (4 * C) ** 0.5
        P[pre + ".net.2.bias"] = torch.zeros(C)
    for nm in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"):
        P["attn." + nm + ".weight"] = torch.randn(C, C, generator=g) / C ** 0.5
        P["attn." + nm + ".bias"] = torch.zeros(C)
    e2, h2 = sd3_ref.joint_transformer_block(P, heads, torch.randn(48, 9, C, generator=g), torch.randn(48, 5, C, generator=g), torch.randn(48, C, generator=g))
    assert e2.shape == (48, 5, C) and h2.shape == (48, 9, C) and torch.isfinite(e2).all() and torch.isfinite(h2).all()


# ------------------------------------------------------------------------------------------------------------------------------
# the MM-DiT backbone (host mirror of the reference's CustomSD3Transformer2DModel on the native kernels) against the restatement of
# diffusers' SD3Transformer2DModel in oracle/sd3_ref.py.  PARITY UNPINNED for everything but the processors (diffusers is on
# neither box): these tests show that the native composition computes what the restatement says, not that the restatement is
# diffusers.  fp16 storage between ~25 operators per block: 1e-2 of the output scale (max), 4e-3 rms.
def _tiny_sd3(layers=3, dual=(0,), qk_norm="rms_norm", heads=2, seed=11):
    from univst_amd.backbones.video_diffusion_sd3.models.transformer_3D_model import CustomSD3Transformer2DModel
    torch.manual_seed(seed)
    m = CustomSD3Transformer2DModel(sample_size=32, patch_size=2, in_channels=16, num_layers=layers, attention_head_dim=64,
                                    num_attention_heads=heads, joint_attention_dim=64, caption_projection_dim=64 * heads, pooled_projection_dim=32,
                                    out_channels=16, pos_embed_max_size=24, dual_attention_layers=dual, qk_norm=qk_norm)
    for n, p in m.named_parameters():                       # adaLN / RMS weights away from their trivial values
        if n.endswith("norm_q.weight") or n.endswith("norm_k.weight") or "norm_added" in n:
            p.data = 1.0 + 0.2 * torch.randn_like(p)
    m = m.half()
    P = {k: v.float() for k, v in m.state_dict().items()}   # the oracle sees the fp16-rounded weights
    return m.cuda(), P


def _sd3_inputs(B, hw, T, seed=5):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(B, 16, hw, hw, generator=g).half(), torch.randn(B, T, 64, generator=g).half(),
            torch.randn(B, 32, generator=g).half(), torch.tensor([437.0]))


def test_sd3_transformer_stock_processors_vs_restatement(nat, tmp_path):
    """diffusers' stock joint attention (no cross-frame keys), dual-attention first block, context_pre_only last block; the
    feature dump of transformer_3D_model.py:76-82."""
    m, P = _tiny_sd3()
    lat, enc, pooled, t = _sd3_inputs(6, 16, 7)
    feats = {}
    want = sd3_ref.sd3_transformer(P, m.config, lat.float(), enc.float(), pooled.float(), t, attn_kw=dict(clip_length=0), features=feats)
    got = m(hidden_states=lat.cuda(), timestep=t.cuda(), encoder_hidden_states=enc.cuda(), pooled_projections=pooled.cuda(), return_dict=False,
            idx=3, ft_indices=[1], ft_timesteps=[3], ft_path=str(tmp_path))[0]
    mx, rms = errs(got, want)
    assert mx < 1e-2 and rms < 4e-3, (mx, rms)
    f = torch.load(tmp_path / "inversion_feature_map_1_block_3_step.pt", weights_only=True)
    assert f.shape == (6, 8, 8, 128)
    mx, rms = errs(f.reshape(6, 64, 128), feats[1])
    assert mx < 1e-2 and rms < 4e-3, (mx, rms)
    # skip_layers (transformer_3D_model.py:58-75) and the object output
    got2 = m(hidden_states=lat.cuda(), timestep=t.cuda().expand(6), encoder_hidden_states=enc.cuda(), pooled_projections=pooled.cuda(), skip_layers=[1]).sample
    assert got2.shape == got.shape and not torch.allclose(got2, got)


@pytest.mark.parametrize("idx", [10, 45])
def test_sd3_transformer_univst_processors_vs_restatement(nat, idx):
    """the three-branch batch of the transfer loop (3 x 16 frames) with the reference's processors registered the way
    register_spatial_attention_pnp does (AttentionShiftProcessor on every attn / attn2), inside (idx 10) and outside (45) the window."""
    from univst_amd.backbones.video_diffusion_sd3 import pnp_utils
    m, P = _tiny_sd3(layers=2, dual=(0,))
    pnp_utils.register_spatial_attention_pnp(types.SimpleNamespace(transformer=m), eta1=0.0, eta2=0.6)
    assert all(isinstance(p, pnp_utils.AttentionShiftProcessor) for p in m.attn_processors.values()) and len(m.attn_processors) == 3
    lat, enc, pooled, t = _sd3_inputs(48, 8, 5, seed=9)
    want = sd3_ref.sd3_transformer(P, m.config, lat.float(), enc.float(), pooled.float(), t,
                                   attn_kw=dict(idx=idx, shift=True, eta1=0.0, eta2=0.6, clip_length=16))
    got = m(hidden_states=lat.cuda(), timestep=t.cuda().expand(48), encoder_hidden_states=enc.cuda(), pooled_projections=pooled.cuda(),
            return_dict=False, joint_attention_kwargs={"idx": idx})[0]
    mx, rms = errs(got, want)
    assert mx < 1e-2 and rms < 4e-3, (mx, rms)


def test_sd3_joint_attention_long_tokens_pipelined_kernel(nat):
    """1100 image + 77 text tokens per frame (ragged tails in both segments), head_dim 64, q / k RMSNorm: the image queries take the
    software-pipelined head_dim-64 kernel (prescaled q out of the norm, text tokens as the extra key segment, merged duplicate
    sources), the text queries the generic body — against the G16/G18-pinned oracle, inside the shift window."""
    g = torch.Generator().manual_seed(31)
    heads, dh, N, Nt, F_ = 2, 64, 1100, 77, 3
    C = heads * dh
    P = {}
    for nm in ("to_q", "to_k", "to_v", "add_q_proj", "add_k_proj", "add_v_proj", "to_out.0", "to_add_out"):
        P[nm + ".weight"] = (torch.randn(C, C, generator=g) / C ** 0.5).half().float()
        P[nm + ".bias"] = (0.1 * torch.randn(C, generator=g)).half().float()
    for nm in ("norm_q", "norm_k", "norm_added_q", "norm_added_k"):
        P[nm + ".weight"] = (1.0 + 0.2 * torch.randn(dh, generator=g)).half().float()
    hid = torch.randn(3 * F_, N, C, generator=g).half()
    hid[2 * F_:] = hid[2 * F_:] * 1.4 + 0.2
    enc = torch.randn(3 * F_, Nt, C, generator=g).half()
    for shift, idx in ((False, -1), (True, 12)):
        w_img, w_txt = sd3_ref.joint_attention(P, heads, hid.float(), enc.float(), idx=idx, shift=shift, eta1=0.0, eta2=0.6, clip_length=F_)
        g_img, g_txt = nat.sd3_joint_attention({k: v.half().cuda() for k, v in P.items()}, hid.cuda(), enc.cuda(), heads, clip_length=F_, shift=shift,
                                               idx=idx, eta1=0.0, eta2=0.6)
        for got, want in ((g_img, w_img), (g_txt, w_txt)):
            mx, rms = errs(got, want)
            assert mx < 4e-3 and rms < 1e-3, (shift, mx, rms)


def test_linear_gated_epilogue_and_unfused_block_path(nat):
    """Y = residual + gate[b] * gelu_tanh(X W^T + bias) on the small (128-row), split-K and 256 x 320 (ragged N) kernels against torch
    fp32; and a block whose processor does NOT advertise the fused gated residual (a third-party processor) gives the same output
    as the fused path."""
    g = torch.Generator().manual_seed(21)
    for B, N, K, Nout in ((3, 50, 64, 96), (2, 40, 2048, 64), (12, 4096, 64, 1536)):
        x = torch.randn(B * N, K, generator=g).half().cuda()
        w = (torch.randn(Nout, K, generator=g) / K ** 0.5).half().cuda()
        b = (0.2 * torch.randn(Nout, generator=g)).half().cuda()
        r = torch.randn(B * N, Nout, generator=g).half().cuda()
        emb = torch.randn(B, 3 * Nout, generator=g).half().cuda()
        gate = emb[:, Nout:2 * Nout]
        lin = x.float() @ w.float().T + b.float()
        for act, f in ((None, lambda v: v), (nat.ACT_GELU_TANH, lambda v: torch.nn.functional.gelu(v, approximate="tanh"))):
            want = r.float() + gate.float().repeat_interleave(N, dim=0) * f(lin)
            got = nat.linear_gated(x, w, b, residual=r, act=act, gate=gate, rows_per_gate=N)
            mx, rms = errs(got, want)
            assert mx < 3e-3 and rms < 6e-4, (B, N, K, Nout, act, mx, rms)
        mx, _ = errs(nat.linear_gated(x, w, b, act=nat.ACT_GELU_TANH), torch.nn.functional.gelu(lin, approximate="tanh"))
        assert mx < 3e-3, mx
    m, _ = _tiny_sd3(layers=2, dual=(0,))
    lat, enc, pooled, t = _sd3_inputs(6, 16, 7)
    args = dict(hidden_states=lat.cuda(), timestep=t.cuda(), encoder_hidden_states=enc.cuda(), pooled_projections=pooled.cuda(), return_dict=False)
    fused = m(**args)[0]

    class Plain:                       # no supports_fused_gated_residual: the block applies gate_residual itself
        def __init__(self, inner):
            self.inner = inner

        def __call__(self, attn, hidden_states, encoder_hidden_states=None, **kw):
            return self.inner(attn, hidden_states, encoder_hidden_states=encoder_hidden_states, **kw)
    m.set_attn_processor({n: Plain(p) for n, p in m.attn_processors.items()})
    plain = m(**args)[0]
    mx, rms = errs(plain, fused)
    assert mx < 4e-3 and rms < 1e-3, (mx, rms)


def test_sd3_elementwise_operators(nat):
    g = torch.Generator().manual_seed(3)
    x = (3 * torch.randn(5, 40, 64, generator=g)).half()
    for act, ref in ((nat.ACT_SILU, torch.nn.functional.silu), (nat.ACT_GELU_TANH, lambda v: torch.nn.functional.gelu(v, approximate="tanh"))):
        mx, _ = errs(nat.activation(x.cuda(), act), ref(x.float()))
        assert mx < 1e-3, (act, mx)
    emb = torch.randn(5, 6 * 64, generator=g).half().cuda()
    y = torch.randn(5, 40, 64, generator=g).half()
    got = nat.gate_residual(x.cuda(), emb[:, 128:192], y.cuda())
    mx, _ = errs(got, x.float() + emb[:, 128:192].float().cpu()[:, None] * y.float())
    assert mx < 1e-3, mx
    t = torch.tensor([0.0, 1.0, 437.0, 999.0])
    mx, _ = errs(nat.timestep_embedding(t.cuda(), 256), sd3_ref.timestep_embedding(t))
    assert mx < 2e-3, mx                                    # fp16 storage of values in [-1, 1]
    lat = torch.randn(3, 16, 8, 12, generator=g).half()
    rows = nat.sd3_patchify(lat.cuda(), 2)
    want = torch.nn.functional.unfold(lat.float(), kernel_size=2, stride=2).transpose(1, 2).reshape(-1, 64)       # (c, u, v) order = conv weight
    assert torch.equal(rows.float().cpu(), want)
    r2 = torch.randn(3 * 4 * 6, 64, generator=g).half()
    back = nat.sd3_unpatchify(r2.cuda(), 3, 16, 8, 12, 2)
    want = torch.einsum("nhwpqc->nchpwq", r2.float().reshape(3, 4, 6, 2, 2, 16)).reshape(3, 16, 8, 12)
    assert torch.equal(back.float().cpu(), want)


# ------------------------------------------------------------------------------------------------------------------------------
# the pipeline mirror (pipelines/custom_pipeline.py): its loops against golden G19 — the reference's own video_style_transfer /
# reconstruction over a closed-form velocity field — and, with a mask, against the G19-pinned oracle loop (fixed reading of the
# undefined name).  The "transformer" here is a test double (torch ops on the GPU); everything the pipeline itself computes (mask
# blend, latent AdaIN, eta-interpolated Euler step) is native.  fp16 latents over 50 steps: 6e-3 of the output scale, 2e-3 rms.
class _ToyTransformer:
    config = types.SimpleNamespace(in_channels=4, patch_size=2, sample_size=6, joint_attention_dim=8)
    device = torch.device("cuda")

    def __init__(self, frames):
        self.frames, self.calls = frames, []

    def __call__(self, hidden_states, timestep, encoder_hidden_states=None, pooled_projections=None, return_dict=False, joint_attention_kwargs=None):
        idx = (joint_attention_kwargs or {}).get("idx", 0)
        self.calls.append((tuple(hidden_states.shape), tuple(encoder_hidden_states.shape), tuple(pooled_projections.shape), idx))
        return (sd3_ref.toy_velocity(hidden_states.float(), timestep, idx, self.frames).half(),)


def _sd3_pipe(frames):
    from univst_amd.backbones.video_diffusion_sd3.pipelines.custom_pipeline import CustomStableDiffusion3Pipeline
    from univst_amd.schedulers import FlowMatchEulerDiscreteScheduler
    return CustomStableDiffusion3Pipeline(transformer=_ToyTransformer(frames), scheduler=FlowMatchEulerDiscreteScheduler())


def test_sd3_pipeline_loops_vs_g19(nat, tmp_path):
    from PIL import Image
    import numpy as np
    g = torch.load("tests/golden/g19_sd3_pipeline_loops.pt", weights_only=True)
    ti = sd3_ref.toy_loop_inputs()
    Fr = ti["content"][0].shape[0]
    pipe = _sd3_pipe(Fr)
    pe, pp = torch.zeros(1, 3, 8), torch.zeros(1, 8)
    from univst_amd.backbones.video_diffusion_sd3.pnp_utils import latent_adain
    start = latent_adain(ti["content"][50].half().cuda(), ti["style"][50].half().cuda())
    kw = dict(latents=start, img_latents=ti["content"][0], num_inference_steps=50, prompt_embeds=pe, pooled_prompt_embeds=pp, eta_base=0.85,
              eta_trend="constant", start_step=25, end_step=39, output_type="latent")
    out = pipe.video_style_transfer("", content_inv_latents=ti["content"], style_inv_latents=ti["style"], **kw).images
    mx, rms = errs(out, g["video_style_transfer"])
    assert mx < 6e-3 and rms < 2e-3, (mx, rms)
    calls = pipe.transformer.calls
    assert len(calls) == 50 and calls[7] == ((3 * Fr, 4, 6, 6), (3 * Fr, 3, 8), (3 * Fr, 8), 7)
    # the same through files (the reference's hand-off: ddim_latents_{k}.pt, load_ddim_latents_at_t) and with a mask folder
    cdir, sdir, mdir = tmp_path / "c", tmp_path / "s", tmp_path / "m"
    for d in (cdir, sdir, mdir):
        d.mkdir()
    for k in range(51):
        torch.save(ti["content"][k], cdir / f"ddim_latents_{k}.pt")
        torch.save(ti["style"][k], sdir / f"ddim_latents_{k}.pt")
    for f in range(Fr):
        Image.fromarray((ti["mask"][0, f].numpy() * 255).astype(np.uint8)).save(mdir / ("%05d.png" % f))
    out2 = pipe.video_style_transfer("", content_inv_path=str(cdir), style_inv_path=str(sdir), **kw).images
    assert torch.equal(out2, out)
    outm = pipe.video_style_transfer("", content_inv_path=str(cdir), style_inv_path=str(sdir), mask_path=str(mdir), **kw).images
    ts, sig = sd3_ref.flow_match_schedule(50)
    eta = sd3_ref.generate_eta_values(ts, 25, 39, 0.85, "constant")
    vf = lambda x, t, i: sd3_ref.toy_velocity(x, t, i, Fr)          # noqa: E731
    want = sd3_ref.sd3_transfer_loop(vf, start.float().cpu(), ti["content"][0], ti["content"], ti["style"], ts, sig, eta, mask=ti["mask"])
    mx, rms = errs(outm, want)
    assert mx < 6e-3 and rms < 2e-3, (mx, rms)
    assert errs(outm, g["video_style_transfer"])[0] > 2e-2          # the mask matters
    # reconstruction (:45-124): no idx is passed to the transformer
    pipe2 = _sd3_pipe(Fr)
    pipe2.encode_prompt = lambda **k: (pe, None, pp, None)
    rec = pipe2.reconstruction(ti["content"][0], ti["content"][50], 0.85, "constant", 25, 39, prompt="", DTYPE=torch.float32, num_inference_steps=50,
                               output_type="latent")
    mx, rms = errs(rec, g["reconstruction"])
    assert mx < 6e-3 and rms < 2e-3, (mx, rms)
    assert all(c[3] == 0 for c in pipe2.transformer.calls)


class _FakeVAE:
    """16-channel linear stand-in for the stock SD3 AutoencoderKL (third-party, never re-implemented): 8x8 average pooling and a
    fixed 3 -> 16 channel map; decode inverts it approximately.  Only the call sites are exercised."""
    config = types.SimpleNamespace(scaling_factor=1.5305, shift_factor=0.0609)

    def __init__(self):
        g = torch.Generator().manual_seed(2)
        self.w = torch.randn(16, 3, generator=g).cuda()

    def parameters(self):
        return iter([self.w.half()])

    def encode(self, px):
        z = torch.einsum("oc,nchw->nohw", self.w, torch.nn.functional.avg_pool2d(px.float(), 8)).to(px.dtype)
        return types.SimpleNamespace(latent_dist=types.SimpleNamespace(sample=lambda: z))

    def decode(self, z, return_dict=False):
        img = torch.einsum("co,nohw->nchw", torch.linalg.pinv(self.w), z.float())
        return (torch.nn.functional.interpolate(img, scale_factor=8, mode="nearest").to(z.dtype),)


def test_sd3_end_to_end_chain_through_files(nat, tmp_path):
    """the four steps of scripts/start_sd3.sh on the native MM-DiT with stand-in VAE / prompt embeddings, through the same files:
    content + style inversion (rf_solver, CrossFrameProcessor, feature dump) -> mask propagation on the dumped features ->
    video_style_transfer with AttentionShiftProcessor and the propagated masks.  Plumbing check (shapes, files, flags); parity of the
    pieces is covered above."""
    import numpy as np
    from PIL import Image
    from univst_amd.backbones.video_diffusion_sd3 import pnp_utils
    from univst_amd.backbones.video_diffusion_sd3.pipelines.custom_pipeline import CustomStableDiffusion3Pipeline
    from univst_amd.inversion_tools import flow_inversion as fi
    from univst_amd.schedulers import FlowMatchEulerDiscreteScheduler
    from univst_amd.src import mask_propagation as mp
    from univst_amd.src.util import load_ddim_latents_at_t
    Fr, HW = 16, 64
    m, _ = _tiny_sd3(layers=2, dual=(0,))
    m.set_attn_processor({n: pnp_utils.CrossFrameProcessor() for n in m.attn_processors})
    pipe = CustomStableDiffusion3Pipeline(transformer=m, scheduler=FlowMatchEulerDiscreteScheduler(), vae=_FakeVAE())
    g = torch.Generator().manual_seed(4)
    pe, pp = torch.randn(1, 5, 64, generator=g).half().cuda(), torch.randn(1, 32, generator=g).half().cuda()
    pipe.encode_prompt = lambda **k: (pe, None, pp, None)
    cdir, sdir = tmp_path / "content", tmp_path / "style.png"
    cdir.mkdir()
    rng = np.random.default_rng(0)
    base = rng.integers(0, 256, (HW, HW, 3), dtype=np.uint8)
    for f in range(Fr):
        Image.fromarray(np.roll(base, 2 * f, axis=1)).save(cdir / ("%05d.png" % f))
    Image.fromarray(rng.integers(0, 256, (HW, HW, 3), dtype=np.uint8)).save(sdir)
    out = {k: tmp_path / k for k in ("c_inv", "c_rec", "c_ft", "s_inv", "s_rec", "masks", "styl")}
    for d in out.values():
        d.mkdir()
    zc = fi.content_inversion_reconstruction(pipe, str(cdir), str(out["c_inv"]), str(out["c_rec"]), Fr, HW, HW, 50, torch.float16, ft_indices=[1],
                                             ft_timesteps=[5], ft_path=str(out["c_ft"]), is_rf_solver=True, reconstruct=False)
    zs = fi.style_inversion_reconstruction(pipe, str(sdir), str(out["s_inv"]), str(out["s_rec"]), Fr, HW, HW, 50, torch.float16, is_rf_solver=True,
                                           reconstruct=False)
    assert zc.shape == (Fr, 16, 8, 8) and torch.isfinite(zc).all() and torch.isfinite(zs).all()
    assert len(list(out["c_inv"].glob("ddim_latents_*.pt"))) == 51 and len(list(out["s_inv"].glob("ddim_latents_*.pt"))) == 51
    ft = out["c_ft"] / "inversion_feature_map_1_block_5_step.pt"
    dumped = torch.load(ft, weights_only=True)
    assert dumped.shape == (Fr, 4, 4, 128)
    # rf_solver's MIDPOINT evaluation takes no ft_* arguments (flow_inversion.py:242-249), so the file holds the hidden state of the
    # FIRST evaluation of step 5 — the transformer at (ddim_latents_5, t_curr) — not the midpoint's, which would overwrite it otherwise
    import tempfile
    z5 = load_ddim_latents_at_t(5, str(out["c_inv"])).cuda()
    pipe.scheduler.set_timesteps(50, device="cuda")
    sig = torch.flip(pipe.scheduler.sigmas, dims=[0])
    t_curr, t_next = float(sig[5]), float(sig[6])
    with tempfile.TemporaryDirectory() as td:
        for tt, name in ((t_curr, "first"), (t_curr + (t_next - t_curr) / 2, "mid")):
            zin = z5 if name == "first" else None
            if name == "mid":
                v = m(hidden_states=z5, timestep=torch.full((Fr,), 1000 * t_curr, device="cuda", dtype=torch.float16), encoder_hidden_states=pe,
                      pooled_projections=pp, idx=5, return_dict=False)[0]
                zin = (z5.float() + (t_next - t_curr) / 2 * v.float()).half()
            m(hidden_states=zin, timestep=torch.full((Fr,), 1000 * tt, device="cuda", dtype=torch.float16), encoder_hidden_states=pe, pooled_projections=pp,
              idx=5, ft_indices=[1], ft_timesteps=[5], ft_path=td, return_dict=False)
            got = torch.load(os.path.join(td, "inversion_feature_map_1_block_5_step.pt"), weights_only=True)
            d = (got.float() - dumped.float()).abs().max().item() / dumped.float().abs().max().item()
            if name == "first":
                assert d < 1e-3, ("the dumped feature is not the first evaluation's hidden state", d)
            else:
                assert d > 1e-2, ("the midpoint evaluation's features are indistinguishable: the check is vacuous", d)
    first = np.zeros((HW, HW), np.uint8)
    first[16:48, 8:40] = 1
    Image.fromarray(first).save(tmp_path / "first.png")
    margs = mp.build_parser().parse_args(["--feature_path", str(ft), "--backbone", "sd3", "--mask_path", str(tmp_path / "first.png"),
                                          "--output_path", str(out["masks"]), "--num_frames", str(Fr)])
    mp.video_mask_propogation(margs)
    mdir = out["masks"] / "sd3" / "first"
    assert len(list(mdir.glob("*.png"))) == Fr
    # step 4 (run_video_style_transfer_sd3.py:84-101)
    noises = load_ddim_latents_at_t(50, str(out["c_inv"])).cuda()
    snoise = load_ddim_latents_at_t(50, str(out["s_inv"])).cuda()
    start = pnp_utils.latent_adain(noises, snoise)
    pnp_utils.register_spatial_attention_pnp(pipe)
    res = pipe.video_style_transfer("", latents=start, img_latents=load_ddim_latents_at_t(0, str(out["c_inv"])), num_inference_steps=50,
                                    content_inv_path=str(out["c_inv"]), style_inv_path=str(out["s_inv"]), mask_path=str(mdir), eta_base=0.85,
                                    eta_trend="constant", start_step=25, end_step=39).images
    assert len(res) == Fr and res[0].size == (HW, HW)
    lat = pipe.video_style_transfer("", latents=start, img_latents=load_ddim_latents_at_t(0, str(out["c_inv"])), num_inference_steps=50,
                                    content_inv_path=str(out["c_inv"]), style_inv_path=str(out["s_inv"]), mask_path=None, eta_base=0.85,
                                    eta_trend="constant", start_step=25, end_step=39, output_type="latent").images
    assert lat.shape == (Fr, 16, 8, 8) and torch.isfinite(lat).all()


# ------------------------------------------------------------------------------------------------------------------------------
# frame shard of the SD3 path: two PROCESSES on this box's one GPU (as the SD-v1.5 shard is tested), the library's IPC communicator
# carrying K | V of the clip's first frame and of the previous frame inside univst_sd3_joint_attention
def _sd3_shard_rank(rank, world, port, q):
    import os
    import traceback
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        from univst_amd.backbones.video_diffusion_sd3 import pnp_utils
        from univst_amd.parallel import Sd3FrameShard
        m, _ = _tiny_sd3(layers=2, dual=(0,))
        pnp_utils.register_spatial_attention_pnp(types.SimpleNamespace(transformer=m), eta1=0.0, eta2=0.6)
        lat, enc, pooled, t = _sd3_inputs(48, 8, 5, seed=9)
        sh = Sd3FrameShard(rank, world, 16).attach(m, tokens=16)
        out = {}
        for idx in (10, 45, 11):                      # inside / outside the shift window; three consecutive exchanges (parity reuse)
            v = m(hidden_states=sh.slice_branches(lat).cuda(), timestep=t.cuda().expand(3 * sh.local), encoder_hidden_states=sh.slice_branches(enc).cuda(),
                  pooled_projections=sh.slice_branches(pooled).cuda(), return_dict=False, joint_attention_kwargs={"idx": idx})[0]
            out[idx] = torch.stack([sh.gather_frames(c) for c in v.chunk(3)]).cpu()      # [branch, F, C, h, w]: every rank ends with all frames
        torch.cuda.synchronize()
        from univst_amd import _native
        st = _native.load().univst_comm_status(sh.comm.ptr)                      # a bounded wait that gave up leaves wrong rows AND a code: name it
        assert st == 0, f"rank {rank}: the communicator gave up waiting (status {st})"
        q.put((rank, {k: v.float().numpy() for k, v in out.items()}, None))      # by value: a tensor would travel as an fd of a process that may be gone
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, None, traceback.format_exc()))


@pytest.mark.parametrize("world", [2, 8])
def test_sd3_frame_shard_two_processes_ipc(nat, world, monkeypatch):
    """world 2: eight frames per rank; world 8: two frames per rank (rank 1's previous frame is also the clip's second one, every rank
    but 0 takes both halo blocks from the communicator, rank 0 multicasts to seven peers).
    World 8 runs with the exchange on the forward's own stream (UNIVST_KV_OVERLAP=0): eight processes with TWO queues each oversubscribe the hardware
    queues of the one GPU they share, a spinning wait kernel then starves a peer's unmapped queue for seconds and a bounded wait gives up (seen in 2 of
    8 runs, status 104).  That is a property of ranks sharing a device — on a node every GPU has one process; world 2 keeps the forked stream."""
    import socket
    import torch.multiprocessing as mp
    if world > 2:
        monkeypatch.setenv("UNIVST_KV_OVERLAP", "0")
    from univst_amd.backbones.video_diffusion_sd3 import pnp_utils
    m, _ = _tiny_sd3(layers=2, dual=(0,))
    pnp_utils.register_spatial_attention_pnp(types.SimpleNamespace(transformer=m), eta1=0.0, eta2=0.6)
    lat, enc, pooled, t = _sd3_inputs(48, 8, 5, seed=9)
    ref = {}
    for idx in (10, 45, 11):
        v = m(hidden_states=lat.cuda(), timestep=t.cuda().expand(48), encoder_hidden_states=enc.cuda(), pooled_projections=pooled.cuda(),
              return_dict=False, joint_attention_kwargs={"idx": idx})[0]
        ref[idx] = torch.stack(list(v.chunk(3))).cpu()
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_sd3_shard_rank, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = {}
    for _ in range(world):
        rank, out, err = q.get(timeout=600)
        assert err is None, err
        res[rank] = out
    for pr in procs:
        pr.join(timeout=120)
    for r in range(world):
        for idx in ref:
            mx, rms = errs(torch.from_numpy(res[r][idx]), ref[idx])
            assert mx < 4e-3 and rms < 1e-3, (r, idx, mx, rms)


def _sd3_pipeline_args(Fr=16, hw=8, seed=21):
    """in-memory arguments of CustomStableDiffusion3Pipeline.video_style_transfer for the tiny MM-DiT: the 51 content / style inversion
    latents, start latents, image latents, prompt embeddings, load_mask-style masks (uint8 {0,1} [1, F, 8 hw, 8 hw])"""
    g = torch.Generator().manual_seed(seed)
    ci = [torch.randn(Fr, 16, hw, hw, generator=g).half() for _ in range(51)]
    sy = [torch.randn(Fr, 16, hw, hw, generator=g).half() * 0.8 + 0.1 for _ in range(51)]
    start = torch.randn(Fr, 16, hw, hw, generator=g).half()
    img = torch.randn(Fr, 16, hw, hw, generator=g).half()
    pe, pp = torch.randn(1, 5, 64, generator=g).half(), torch.randn(1, 32, generator=g).half()
    masks = torch.zeros(1, Fr, 8 * hw, 8 * hw, dtype=torch.uint8)
    for f in range(Fr):
        masks[0, f, 8 + f:40 + f, 16:48] = 1
    return ci, sy, start, img, pe, pp, masks


def _sd3_pipeline_call(pipe, args, **kw):
    ci, sy, start, img, pe, pp, masks = args
    return pipe.video_style_transfer("", latents=start.cuda(), img_latents=img.cuda(), num_inference_steps=50, content_inv_latents=ci, style_inv_latents=sy,
                                     masks=masks, prompt_embeds=pe.cuda(), pooled_prompt_embeds=pp.cuda(), eta_base=0.85, eta_trend="constant", start_step=25,
                                     end_step=39, output_type="latent", **kw).images


def _sd3_pipeline_rank(rank, world, port, q):
    """one process of `torchrun --nproc-per-node 2 src/sd3/run_video_style_transfer_sd3.py` from the pipeline on: launcher environment,
    parallel.init_distributed, then the PIPELINE METHOD with the full clip as arguments on every rank"""
    import os
    import traceback
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      UNIVST_DIST_BACKEND="gloo")
    try:
        import torch.distributed as dist
        from univst_amd.parallel import init_distributed, Sd3FrameShard
        from univst_amd.backbones.video_diffusion_sd3 import pnp_utils
        from univst_amd.backbones.video_diffusion_sd3.pipelines.custom_pipeline import CustomStableDiffusion3Pipeline
        from univst_amd.schedulers import FlowMatchEulerDiscreteScheduler
        assert init_distributed() == (rank, world)
        m, _ = _tiny_sd3(layers=2, dual=(0,))
        pipe = CustomStableDiffusion3Pipeline(transformer=m, scheduler=FlowMatchEulerDiscreteScheduler())
        pnp_utils.register_spatial_attention_pnp(pipe)
        out = _sd3_pipeline_call(pipe, _sd3_pipeline_args())
        assert isinstance(m.transformer_blocks[0].attn._uv_frame_shard, Sd3FrameShard)
        torch.cuda.synchronize()
        q.put((rank, out.float().cpu().numpy(), None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, None, traceback.format_exc()))


def test_sd3_pipeline_method_frame_sharded_two_processes(nat):
    """the SD3 twin of test_pipeline_method_frame_sharded_two_processes: two processes with the launcher's environment call
    ``CustomStableDiffusion3Pipeline.video_style_transfer`` (tiny MM-DiT, 16 frames, 50 steps, masks, eta window) with the FULL clip;
    the method shards the frames itself (Sd3FrameShard over the IPC communicator), every rank gets all frames back, and they equal the
    unsharded call of this process."""
    import socket
    import torch.multiprocessing as mp
    from univst_amd.backbones.video_diffusion_sd3 import pnp_utils
    from univst_amd.backbones.video_diffusion_sd3.pipelines.custom_pipeline import CustomStableDiffusion3Pipeline
    from univst_amd.schedulers import FlowMatchEulerDiscreteScheduler
    world = 2
    m, _ = _tiny_sd3(layers=2, dual=(0,))
    pipe = CustomStableDiffusion3Pipeline(transformer=m, scheduler=FlowMatchEulerDiscreteScheduler())
    pnp_utils.register_spatial_attention_pnp(pipe)
    want = _sd3_pipeline_call(pipe, _sd3_pipeline_args()).float().cpu()
    assert torch.isfinite(want).all()
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_sd3_pipeline_rank, args=(r, world, port, q)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = {}
    for _ in range(world):
        rank, out, err = q.get(timeout=900)
        assert err is None, err
        res[rank] = torch.from_numpy(out)
    for pr in procs:
        pr.join(timeout=120)
    for r in range(world):
        assert res[r].shape == want.shape
        mx, rms = errs(res[r], want)
        assert mx < 2e-2 and rms < 5e-3, (r, mx, rms)


def test_bench_sd3_two_ranks_end_to_end():
    """`python bench.py --workload sd3_transfer --gpus 2`: bench.py spawns its two ranks, both share this box's GPU (gloo process group for
    the rendezvous), the frame-sharded MM-DiT step runs through the IPC communicator and rank 0 prints ONE JSON line with n_gpus = 2
    (full SD3.5-medium weights, small clip: timing of two ranks on one GPU means nothing)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "sd3_transfer", "--gpus", "2", "--backend", "gloo", "--frames", "4",
                        "--latent", "16", "--steps", "2", "--warmup", "1", "--no-profile"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["parallelism"].startswith("frames2") and d["scaling"] == "strong"


# ------------------------------------------------------------------------------------------------------------------------------
# BASELINE config 5 at its own size: SD3.5-medium widths (24 blocks x 1536, 24 heads of 64, 13 dual-attention
# blocks, context_pre_only last block), 1024 x 1024 frames = 4096 image tokens + 333 text tokens per frame, the reference's processors
# registered.  Frames reduced to 3 branches x 3 so the fp32 oracle fits on the device next to the model (2.2 B parameters twice): the
# geometry per frame — ragged 256 x 320 tiles of the 1536 / 4608 / 6144-wide linears, attn_pp64_kernel over 3 x 4096 + 333 keys,
# merged duplicate sources at f = 0, 1 — is config 5's.
@pytest.mark.parametrize("rank,world", [(1, 8), (0, 8), (7, 8)])
def test_sd3_emulated_communicator_rank(nat, rank, world):
    """the sharded joint attention of one rank ALONE through the communicator's emulated mode (bench.py --workload sd3_transfer --emulate-rank r/8
    --emulate-wire G --comm-emulated): pack, post on the forked stream, two-phase joint attention on ranks > 0, flag wait, unpack, barrier — finite
    output, no bounded wait gave up, wire time accounted (one exchange + one barrier per attention call)."""
    from univst_amd import _native
    from univst_amd.backbones.video_diffusion_sd3 import pnp_utils
    from univst_amd.parallel import Sd3FrameShard, EmulatedIpcComm
    m, _ = _tiny_sd3(layers=2, dual=(0,))
    pnp_utils.register_spatial_attention_pnp(types.SimpleNamespace(transformer=m), eta1=0.0, eta2=0.6)
    lat, enc, pooled, t = _sd3_inputs(48, 8, 5, seed=9)
    comm = EmulatedIpcComm(rank, world, 1 << 17, wire_gbps=50.0, latency_us=2.0)
    sh = Sd3FrameShard(rank, world, 16, comm=comm).attach(m, tokens=16)
    comm.wire_us()
    for idx in (10, 45):
        v = m(hidden_states=sh.slice_branches(lat).cuda(), timestep=t.cuda().expand(3 * sh.local), encoder_hidden_states=sh.slice_branches(enc).cuda(),
              pooled_projections=sh.slice_branches(pooled).cuda(), return_dict=False, joint_attention_kwargs={"idx": idx})[0]
        torch.cuda.synchronize()
        assert torch.isfinite(v.float()).all()
    assert _native.load().univst_comm_status(comm.ptr) == 0
    # 2 forwards x 3 attention calls (2 layers, one dual): a barrier each, and on ranks > 0 two incoming packs each (rank 0 receives nothing)
    assert comm.wire_us() >= 2 * 3 * 2.0 * (3 if rank > 0 else 1)
    comm.close()


def _sd35_medium_for_tests():
    from univst_amd.backbones.video_diffusion_sd3 import pnp_utils
    from univst_amd.backbones.video_diffusion_sd3.models.transformer_3D_model import sd35_medium
    torch.manual_seed(1905)
    with torch.device("cuda"):
        m = sd35_medium()
    with torch.no_grad():
        for n, p in m.named_parameters():
            if n.endswith("norm_q.weight") or n.endswith("norm_k.weight") or "norm_added" in n:
                p.copy_(1.0 + 0.2 * torch.randn_like(p))
    m = m.half().requires_grad_(False)
    pnp_utils.register_spatial_attention_pnp(types.SimpleNamespace(transformer=m), eta1=0.0, eta2=0.6)
    return m


def _sd35_config5_inputs(Fc=16, hl=128):
    g = torch.Generator().manual_seed(7)
    lat = torch.randn(3 * Fc, 16, hl, hl, generator=g).half()
    lat[2 * Fc:] = lat[2 * Fc:] * 1.2 + 0.1
    enc = torch.randn(1, 77 + 256, 4096, generator=g).half().expand(3 * Fc, -1, -1).contiguous()
    pooled = torch.randn(1, 2048, generator=g).half().expand(3 * Fc, -1).contiguous()
    return lat, enc, pooled, torch.tensor([437.0])


def _sd35_config5_rank(rank, world, port, q):
    import os
    import traceback
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    try:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        from univst_amd.parallel import Sd3FrameShard
        m = _sd35_medium_for_tests()
        lat, enc, pooled, t = _sd35_config5_inputs()
        sh = Sd3FrameShard(rank, world, 16).attach(m, tokens=64 * 64)
        v = m(hidden_states=sh.slice_branches(lat).cuda(), timestep=t.cuda().expand(3 * sh.local), encoder_hidden_states=sh.slice_branches(enc).cuda(),
              pooled_projections=sh.slice_branches(pooled).cuda(), return_dict=False, joint_attention_kwargs={"idx": 12})[0]
        torch.cuda.synchronize()
        q.put((rank, v.float().cpu().numpy(), None))          # this rank's frames of the three branches [3 * local, 16, h, w]
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        q.put((rank, None, traceback.format_exc()))


def test_sd35_medium_config5_size_16_frames_shard_equals_unsharded(nat):
    """BASELINE config 5 at ITS OWN size — SD3.5-medium, 1024 px, 16 frames per branch, three branches, inside the shift window — once unsharded and once
    frame-sharded over two PROCESSES through the library's IPC communicator (eight frames per rank; round 6: the exchange posted on the forked stream
    before the text projections, the joint attention in two phases on rank 1): finite, and shard == unsharded up to fp16 summation order.  (The
    3-frame case of test_sd35_medium_three_branch_forward_at_1024px_vs_oracle pins the numbers to the fp32 oracle; the oracle at 16 frames would be
    12621 x 4429 fp32 scores x 24 heads x 48 frames.)"""
    import socket
    import torch.multiprocessing as mp
    m = _sd35_medium_for_tests()
    lat, enc, pooled, t = _sd35_config5_inputs()
    want = m(hidden_states=lat.cuda(), timestep=t.cuda().expand(48), encoder_hidden_states=enc.cuda(), pooled_projections=pooled.cuda(),
             return_dict=False, joint_attention_kwargs={"idx": 12})[0].float().cpu()
    assert torch.isfinite(want).all()
    del m
    torch.cuda.empty_cache()
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_sd35_config5_rank, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = {}
    for _ in range(2):
        rank, out, err = q.get(timeout=900)
        assert err is None, err
        res[rank] = torch.from_numpy(out)
    for pr in procs:
        pr.join(timeout=120)
    w5 = want.view(3, 16, *want.shape[1:])
    for r in range(2):
        got = res[r].view(3, 8, *want.shape[1:])
        mx, rms = errs(got, w5[:, 8 * r:8 * r + 8])
        assert torch.isfinite(got).all() and mx < 6e-3 and rms < 1.5e-3, (r, mx, rms)


@pytest.mark.parametrize("idx", [12, 40])
def test_sd35_medium_three_branch_forward_at_1024px_vs_oracle(nat, idx):
    """one three-branch MM-DiT forward of the transfer loop inside (idx 12) / outside (idx 40) the shift window vs
    oracle/sd3_ref.sd3_transformer evaluated in fp32 ON THE DEVICE with the same fp16-valued weights (reference:
    backbones/video_diffusion_sd3/pnp_utils.py:135-271, models/transformer_3D_model.py:12-113).  Tolerance: 24 blocks x ~25 fp16-stored
    operators: max 1e-2 of max|ref|, relative RMS 5e-3 (measured 1.9e-3 / 1.8e-3; the 2-3-block tests above use 1e-2 / 4e-3)."""
    import json
    import os
    from univst_amd.backbones.video_diffusion_sd3 import pnp_utils
    from univst_amd.backbones.video_diffusion_sd3.models.transformer_3D_model import sd35_medium
    torch.manual_seed(1905)
    with torch.device("cuda"):
        m = sd35_medium()
    with torch.no_grad():
        for n, p in m.named_parameters():                   # RMS / adaLN parameters away from their trivial values, as in _tiny_sd3
            if n.endswith("norm_q.weight") or n.endswith("norm_k.weight") or "norm_added" in n:
                p.copy_(1.0 + 0.2 * torch.randn_like(p))
    m = m.half().requires_grad_(False)
    pnp_utils.register_spatial_attention_pnp(types.SimpleNamespace(transformer=m), eta1=0.0, eta2=0.6)
    Fc, hl, T = 3, 128, 77 + 256
    for proc in m.attn_processors.values():
        proc.clip_length = Fc                               # (the reference hard-codes 16 frames per clip, pnp_utils.py:26)
    g = torch.Generator(device="cuda").manual_seed(7)
    lat = torch.randn(3 * Fc, 16, hl, hl, generator=g, device="cuda").half()
    lat[2 * Fc:] = lat[2 * Fc:] * 1.2 + 0.1                  # the stylised branch away from the content's statistics
    enc = torch.randn(1, T, 4096, generator=g, device="cuda").half().expand(3 * Fc, -1, -1).contiguous()
    pooled = torch.randn(1, 2048, generator=g, device="cuda").half().expand(3 * Fc, -1).contiguous()
    t = torch.tensor([437.0], device="cuda")
    got = m(hidden_states=lat, timestep=t.expand(3 * Fc), encoder_hidden_states=enc, pooled_projections=pooled, return_dict=False,
            joint_attention_kwargs={"idx": idx})[0].float()
    torch.cuda.synchronize()
    P = {k: v.float() for k, v in m.state_dict().items()}
    sd3_ref.SDPA_MAX_BATCH = 1                               # 24 x 4429 x 12621 fp32 scores = 5.4 GB per frame
    try:
        with torch.no_grad():
            want = sd3_ref.sd3_transformer(P, m.config, lat.float(), enc.float(), pooled.float(), t,
                                           attn_kw=dict(idx=idx, shift=True, eta1=0.0, eta2=0.6, clip_length=Fc))
    finally:
        sd3_ref.SDPA_MAX_BATCH = None
    mx, rms = errs(got, want)
    print(f"SD3.5-medium 3x{Fc} frames at 1024 px, idx {idx}: max {mx:.2e} rms {rms:.2e}")
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_sd3_config5_size.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    data = json.load(open(path)) if os.path.exists(path) else {}
    data[f"sd35_medium_3x{Fc}x1024px_idx{idx}"] = {"max_rel_to_max": mx, "rel_rms": rms}
    json.dump(data, open(path, "w"), indent=1, sort_keys=True)
    assert mx < 1e-2 and rms < 5e-3, (mx, rms)

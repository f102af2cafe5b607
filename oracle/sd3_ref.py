"""Oracle groundwork for SURVEY §8f-4 (SD3 / SD3.5 rectified-flow backbone): the REFERENCE-OWNED pieces of that path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The HIP path it checks: csrc/sd3.hip behind
univst_amd/backbones/video_diffusion_sd3/ and univst_amd/inversion_tools/flow_inversion.py.  Pinned to reference code by goldens
G15-G18 (each function cites what it follows):

  * attention_adain / latent_adain of the SD3 plugin          backbones/video_diffusion_sd3/pnp_utils.py:287-316
  * the cross-frame key/value gather ['first', -1, 0]          pnp_utils.py:27,53-78
  * CrossFrameProcessor / AttentionShiftProcessor              pnp_utils.py:9-131 / :134-271
  * rf_inversion / rf_solver                                   inversion_tools/flow_inversion.py:123-264

The reference is BROKEN at HEAD on this path: AttentionShiftProcessor reads an attribute `self.thresh2` that is
never set (pnp_utils.py:186).  The documented FIXED READING used here and by the golden generator is thresh2 == eta2 (the only
value that makes beta run from 0.9 at eta1*50 to 0.1 at eta2*50, like the SD-v1.5 closure, pnp_utils.py:49-50 of that plugin);
the generator sets that attribute on the processor instance before calling the reference's own __call__ — no reference code is
edited or copied.  The MM-DiT backbone itself (diffusers SD3Transformer2DModel: patch embedding, adaLN, the joint transformer
blocks around these processors) and the FlowMatchEuler sigma schedule are third-party and absent: the restatements at the end of
this file (`joint_transformer_block`, `sd3_block`, `sd3_transformer`) are PARITY UNPINNED and say so.
"""
import math
from typing import Callable, Dict, List, Optional

import torch
import torch.nn.functional as F


def attention_adain(cnt: torch.Tensor, sty: torch.Tensor) -> torch.Tensor:
    """pnp_utils.py:287-300 on [B, heads, N, d]: style mean / (unbiased) std over the token axis, applied to
    F.instance_norm(cnt) — which on a 4-D tensor treats dim 1 (heads) as channels and normalises over (N, d) JOINTLY with the
    biased variance and eps 1e-5 (the same layout quirk as the SD-v1.5 plugin)."""
    sty_mean = sty.mean(dim=[-2], keepdim=True)
    sty_std = sty.std(dim=[-2], keepdim=True)
    mu = cnt.mean(dim=(2, 3), keepdim=True)
    var = cnt.var(dim=(2, 3), keepdim=True, unbiased=False)
    return ((cnt - mu) / torch.sqrt(var + 1e-5) * sty_std + sty_mean).to(cnt.dtype)


def latent_adain(cnt: torch.Tensor, sty: torch.Tensor) -> torch.Tensor:
    """pnp_utils.py:303-316 on [B, C, H, W] (frames are the batch): per (frame, channel) statistics over (H, W)."""
    sty_mean = sty.mean(dim=[2, 3], keepdim=True)
    sty_std = sty.std(dim=[2, 3], keepdim=True)
    mu = cnt.mean(dim=(2, 3), keepdim=True)
    var = cnt.var(dim=(2, 3), keepdim=True, unbiased=False)
    return ((cnt - mu) / torch.sqrt(var + 1e-5) * sty_std + sty_mean).to(cnt.dtype)


SDPA_MAX_BATCH = None          # tests at config-5 size set this (frames per scaled_dot_product_attention call)


def cross_frame_gather(x: torch.Tensor, clip_length: int = 16) -> torch.Tensor:
    """pnp_utils.py:53-78: x [(b f), heads, N, d] -> [(b f), heads, 3N, d], the tokens of frames ['first', f-1 (clipped), f] of the
    same clip concatenated along the token axis."""
    bf, hh, n, d = x.shape
    xb = x.view(bf // clip_length, clip_length, hh, n, d)
    first = torch.zeros(clip_length, dtype=torch.long)
    prev = (torch.arange(clip_length) - 1).clip(0, clip_length - 1)
    cur = torch.arange(clip_length)
    out = torch.cat([xb[:, first], xb[:, prev], xb[:, cur]], dim=-2)
    return out.reshape(bf, hh, 3 * n, d)


def _rms(x: torch.Tensor, w: Optional[torch.Tensor], eps: float) -> torch.Tensor:
    """diffusers RMSNorm over the head dim (SD3.5 qk_norm="rms_norm"): x * rsqrt(mean(x^2) + eps) * weight."""
    if w is None:
        return x
    return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + eps) * w


def joint_attention(P: Dict[str, torch.Tensor], heads: int, hidden: torch.Tensor, enc: Optional[torch.Tensor], idx: int = -1,
                    shift: bool = False, eta1: float = 0.0, eta2: float = 0.6, clip_length: int = 16, rms_eps: float = 1e-6,
                    context_pre_only: bool = False):
    """CrossFrameProcessor (shift=False, pnp_utils.py:17-131) / AttentionShiftProcessor (shift=True, :143-271, fixed reading
    thresh2 == eta2).  P holds the attention module's parameters: to_{q,k,v}.{weight,bias}, norm_{q,k}.weight, add_{q,k,v}_proj.*,
    norm_added_{q,k}.weight, to_out.0.*, to_add_out.*.  hidden [(3 f) or (b f), N, C]; enc [(b f), Nt, C] or None."""
    lin = lambda x, n: F.linear(x, P[n + ".weight"], P.get(n + ".bias"))
    B = hidden.shape[0]
    q, k, v = lin(hidden, "to_q"), lin(hidden, "to_k"), lin(hidden, "to_v")
    d = k.shape[-1] // heads
    sp = lambda t: t.view(B, -1, heads, d).transpose(1, 2)
    q, k, v = sp(q), sp(k), sp(v)
    q = _rms(q, P.get("norm_q.weight"), rms_eps)
    k = _rms(k, P.get("norm_k.weight"), rms_eps)
    if shift:
        c = B // 3
        if idx >= eta1 * 50 and idx <= eta2 * 50:          # pnp_utils.py:183-194 (alpha 0.8, gamma 2.0)
            alpha, gamma = 0.8, 2.0
            beta = (0.9 - 0.1) / (eta1 * 50 - eta2 * 50) * (idx - eta2 * 50) + 0.1
            q, k, v = q.clone(), k.clone(), v.clone()
            q[2 * c:3 * c] = alpha * q[:c] + (1 - alpha) * q[2 * c:3 * c]
            k[2 * c:3 * c] = beta * attention_adain(k[2 * c:3 * c], k[c:2 * c]) + (1 - beta) * k[c:2 * c]
            v[2 * c:3 * c] = beta * attention_adain(v[2 * c:3 * c], v[c:2 * c]) + (1 - beta) * v[c:2 * c]
            q[2 * c:3 * c] = gamma * q[2 * c:3 * c]
    if clip_length:                                        # 0: diffusers' stock JointAttnProcessor2_0 (no cross-frame keys)
        k = cross_frame_gather(k, clip_length)
        v = cross_frame_gather(v, clip_length)
    if enc is not None:
        eq, ek, ev = sp(lin(enc, "add_q_proj")), sp(lin(enc, "add_k_proj")), sp(lin(enc, "add_v_proj"))
        eq = _rms(eq, P.get("norm_added_q.weight"), rms_eps)
        ek = _rms(ek, P.get("norm_added_k.weight"), rms_eps)
        q, k, v = torch.cat([q, eq], dim=2), torch.cat([k, ek], dim=2), torch.cat([v, ev], dim=2)
    if SDPA_MAX_BATCH and B > SDPA_MAX_BATCH:              # (memory only: the fp32 math path materialises [B, heads, Nq, Nkv] scores)
        o = torch.cat([F.scaled_dot_product_attention(q[i:i + SDPA_MAX_BATCH], k[i:i + SDPA_MAX_BATCH], v[i:i + SDPA_MAX_BATCH])
                       for i in range(0, B, SDPA_MAX_BATCH)])
    else:
        o = F.scaled_dot_product_attention(q, k, v)
    o = o.transpose(1, 2).reshape(B, -1, heads * d)
    n_img = hidden.shape[1]
    if enc is not None:
        o, eo = o[:, :n_img], o[:, n_img:]
        if not context_pre_only:
            eo = lin(eo, "to_add_out")
        return lin(o, "to_out.0"), eo
    return lin(o, "to_out.0")


def rf_inversion(velocity_fn: Callable, latents: torch.Tensor, sigmas: torch.Tensor, target_noise: torch.Tensor, gamma: float = 0.5) -> List[torch.Tensor]:
    """flow_inversion.py:123-188: controlled forward ODE towards `target_noise`; sigmas as the scheduler gives them (decreasing,
    last 0) — the loop runs them flipped, t from 0 to 1.  velocity_fn(x, t_scaled_by_1000, idx) -> v.  Returns the trajectory."""
    ts = torch.flip(sigmas, dims=[0])
    traj = [latents.clone()]
    for idx, (t_curr, t_prev) in enumerate(zip(ts[:-1], ts[1:])):
        pred = velocity_fn(latents, t_curr * 1000, idx)
        target_v = (target_noise - latents) / (1.0 - t_curr)
        v = gamma * target_v + (1 - gamma) * pred
        latents = latents + (t_prev - t_curr) * v
        traj.append(latents.clone())
    return traj


def rf_solver(velocity_fn: Callable, latents: torch.Tensor, sigmas: torch.Tensor) -> List[torch.Tensor]:
    """flow_inversion.py:191-264: second-order (midpoint-derivative) inversion: x += dt v + dt^2/2 * (v_mid - v) / (dt/2)."""
    ts = torch.flip(sigmas, dims=[0])
    traj = [latents.clone()]
    for idx, (t_curr, t_prev) in enumerate(zip(ts[:-1], ts[1:])):
        pred = velocity_fn(latents, 1000 * t_curr, idx)
        mid = latents + (t_prev - t_curr) / 2 * pred
        pred_mid = velocity_fn(mid, 1000 * (t_curr + (t_prev - t_curr) / 2), idx)
        first_order = (pred_mid - pred) / ((t_prev - t_curr) / 2)
        latents = latents + (t_prev - t_curr) * pred + 0.5 * (t_prev - t_curr) ** 2 * first_order
        traj.append(latents.clone())
    return traj


def flow_match_sigmas(n: int, shift: float = 3.0, num_train_timesteps: int = 1000) -> torch.Tensor:
    """diffusers 0.35.1 FlowMatchEulerDiscreteScheduler.set_timesteps(n) with the SD3 config (shift 3.0, no dynamic shifting):
    sigma = shift*s / (1 + (shift-1)*s) on s = linspace(1, 1/T, n), then a trailing 0.  Third-party, parity unpinned."""
    s = torch.linspace(1.0, 1.0 / num_train_timesteps, n, dtype=torch.float32)
    sig = shift * s / (1 + (shift - 1) * s)
    return torch.cat([sig, torch.zeros(1)])


# ------------------------------------------------------------------------------------------------------------------------------
# THIRD-PARTY, PARITY UNPINNED: diffusers 0.35.1 `JointTransformerBlock` (models/attention.py) — the MM-DiT block that calls the
# processors above.  diffusers is absent from the reference tree and from both boxes, so this is restated from its published
# forward (non-dual-attention block, context_pre_only = False) and pinned by nothing but this reading; it exists so that the native
# adaLN-modulate / RMS-norm / joint-attention operators have a block-level composition to be checked against
# (tests/test_gpu_sd3.py), not as a claim about the SD3.5 backbone.
def ada_layer_norm_zero(x: torch.Tensor, temb: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-6):
    """AdaLayerNormZero: emb = Linear(SiLU(temb)) -> (shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp);
    returns LayerNorm(x; no affine) * (1 + scale_msa) + shift_msa and the other four."""
    emb = F.linear(F.silu(temb), w, b)
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = emb.chunk(6, dim=1)
    xn = F.layer_norm(x, (x.shape[-1],), None, None, eps) * (1 + scale_msa[:, None]) + shift_msa[:, None]
    return xn, gate_msa, shift_mlp, scale_mlp, gate_mlp


def feed_forward_gelu_tanh(x: torch.Tensor, P: Dict[str, torch.Tensor], pre: str) -> torch.Tensor:
    """FeedForward(dim, activation_fn="gelu-approximate"): Linear(dim, 4 dim) -> GELU(tanh) -> Linear(4 dim, dim)."""
    h = F.gelu(F.linear(x, P[pre + ".net.0.proj.weight"], P[pre + ".net.0.proj.bias"]), approximate="tanh")
    return F.linear(h, P[pre + ".net.2.weight"], P[pre + ".net.2.bias"])


def joint_transformer_block(P: Dict[str, torch.Tensor], heads: int, hidden: torch.Tensor, enc: torch.Tensor, temb: torch.Tensor,
                            idx: int = -1, shift: bool = False, eta1: float = 0.0, eta2: float = 0.6, clip_length: int = 16):
    """hidden [(b f), N, C], enc [(b f), Nt, C], temb [(b f), C] -> (enc', hidden').  P: the block's state dict
    (norm1.linear.*, norm1_context.linear.*, attn.*, ff.*, ff_context.*)."""
    attnP = {k[len("attn."):]: v for k, v in P.items() if k.startswith("attn.")}
    nh, gate_msa, shift_mlp, scale_mlp, gate_mlp = ada_layer_norm_zero(hidden, temb, P["norm1.linear.weight"], P["norm1.linear.bias"])
    ne, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = ada_layer_norm_zero(enc, temb, P["norm1_context.linear.weight"],
                                                                              P["norm1_context.linear.bias"])
    a_img, a_txt = joint_attention(attnP, heads, nh, ne, idx=idx, shift=shift, eta1=eta1, eta2=eta2, clip_length=clip_length)
    hidden = hidden + gate_msa[:, None] * a_img
    n2 = F.layer_norm(hidden, (hidden.shape[-1],), None, None, 1e-6) * (1 + scale_mlp[:, None]) + shift_mlp[:, None]
    hidden = hidden + gate_mlp[:, None] * feed_forward_gelu_tanh(n2, P, "ff")
    enc = enc + c_gate_msa[:, None] * a_txt
    n2c = F.layer_norm(enc, (enc.shape[-1],), None, None, 1e-6) * (1 + c_scale_mlp[:, None]) + c_shift_mlp[:, None]
    enc = enc + c_gate_mlp[:, None] * feed_forward_gelu_tanh(n2c, P, "ff_context")
    return enc, hidden


# ------------------------------------------------------------------------------------------------------------------------------
# THIRD-PARTY, PARITY UNPINNED: the rest of diffusers 0.35.1 `SD3Transformer2DModel` (models/transformers/transformer_sd3.py) as the
# reference's CustomSD3Transformer2DModel.forward drives it (transformer_3D_model.py:44-103): PatchEmbed with a centre-cropped
# positional table, CombinedTimestepTextProjEmbeddings, the context embedder, the block stack incl. the dual-attention blocks of
# SD3.5-medium (AdaLayerNormZeroX, attn2) and the context_pre_only last block (AdaLayerNormContinuous), norm_out, proj_out and the
# unpatchify.  Restated from the published definitions; nothing in the reference tree or on either box can pin it.  P: the model's
# state dict under diffusers' parameter names, fp32.
def timestep_embedding(t: torch.Tensor, dim: int = 256, flip_sin_to_cos: bool = True, downscale_freq_shift: float = 0.0,
                       max_period: float = 10000.0) -> torch.Tensor:
    """embeddings.get_timestep_embedding (scale 1)."""
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(half, dtype=torch.float32, device=t.device) / (half - downscale_freq_shift)
    emb = t.float()[:, None] * torch.exp(exponent)[None]
    emb = torch.cat([emb.sin(), emb.cos()], dim=-1)
    return torch.cat([emb[:, half:], emb[:, :half]], dim=-1) if flip_sin_to_cos else emb


def _lin(x, P, pre):
    return F.linear(x, P[pre + ".weight"], P.get(pre + ".bias"))


def _ln(x):
    return F.layer_norm(x, (x.shape[-1],), None, None, 1e-6)


def sd3_block(P: Dict[str, torch.Tensor], pre: str, heads: int, hidden, enc, temb, dual: bool, context_pre_only: bool,
              attn_kw: Optional[dict] = None):
    """JointTransformerBlock.forward incl. use_dual_attention / context_pre_only.  attn_kw: arguments of `joint_attention` above
    (idx, shift, eta1, eta2, clip_length; clip_length 0 = diffusers' stock processor, no cross-frame keys)."""
    kw = dict(attn_kw or {})
    sub = lambda name: {k[len(pre + name + "."):]: v for k, v in P.items() if k.startswith(pre + name + ".")}   # noqa: E731
    emb = _lin(F.silu(temb), P, pre + "norm1.linear")
    if dual:
        sh_a, sc_a, g_a, sh_m, sc_m, g_m, sh_a2, sc_a2, g_a2 = emb.chunk(9, dim=1)
    else:
        sh_a, sc_a, g_a, sh_m, sc_m, g_m = emb.chunk(6, dim=1)
    nx = _ln(hidden)
    nh = nx * (1 + sc_a[:, None]) + sh_a[:, None]
    cemb = _lin(F.silu(temb), P, pre + "norm1_context.linear")
    if context_pre_only:
        c_sc, c_sh = cemb.chunk(2, dim=1)                           # AdaLayerNormContinuous: scale first
        ne = _ln(enc) * (1 + c_sc[:, None]) + c_sh[:, None]
    else:
        c_sh_a, c_sc_a, c_g_a, c_sh_m, c_sc_m, c_g_m = cemb.chunk(6, dim=1)
        ne = _ln(enc) * (1 + c_sc_a[:, None]) + c_sh_a[:, None]
    a_img, a_txt = joint_attention(sub("attn"), heads, nh, ne, context_pre_only=context_pre_only, **kw)
    hidden = hidden + g_a[:, None] * a_img
    if dual:
        nh2 = nx * (1 + sc_a2[:, None]) + sh_a2[:, None]
        hidden = hidden + g_a2[:, None] * joint_attention(sub("attn2"), heads, nh2, None, **kw)
    n2 = _ln(hidden) * (1 + sc_m[:, None]) + sh_m[:, None]
    hidden = hidden + g_m[:, None] * feed_forward_gelu_tanh(n2, P, pre + "ff")
    if context_pre_only:
        return None, hidden
    enc = enc + c_g_a[:, None] * a_txt
    n2c = _ln(enc) * (1 + c_sc_m[:, None]) + c_sh_m[:, None]
    enc = enc + c_g_m[:, None] * feed_forward_gelu_tanh(n2c, P, pre + "ff_context")
    return enc, hidden


def sd3_transformer(P: Dict[str, torch.Tensor], cfg, latents, enc_in, pooled, timestep, attn_kw: Optional[dict] = None,
                    features: Optional[dict] = None):
    """CustomSD3Transformer2DModel.forward (transformer_3D_model.py:44-103).  cfg: patch_size, num_layers, num_attention_heads,
    out_channels, pos_embed_max_size, dual_attention_layers.  features (optional dict): hidden states after each block."""
    B, Cc, H, W = latents.shape
    p = cfg.patch_size
    hp, wp = H // p, W // p
    x = F.conv2d(latents, P["pos_embed.proj.weight"], P["pos_embed.proj.bias"], stride=p).flatten(2).transpose(1, 2)
    m = cfg.pos_embed_max_size
    top, left = (m - hp) // 2, (m - wp) // 2
    pos = P["pos_embed.pos_embed"].reshape(1, m, m, -1)[:, top:top + hp, left:left + wp].reshape(1, hp * wp, -1)
    x = x + pos
    t_emb = _lin(F.silu(_lin(timestep_embedding(timestep.reshape(-1).expand(B)), P, "time_text_embed.timestep_embedder.linear_1")), P,
                 "time_text_embed.timestep_embedder.linear_2")
    p_emb = _lin(F.silu(_lin(pooled, P, "time_text_embed.text_embedder.linear_1")), P, "time_text_embed.text_embedder.linear_2")
    temb = t_emb + p_emb
    enc = _lin(enc_in, P, "context_embedder")
    for i in range(cfg.num_layers):
        enc, x = sd3_block(P, f"transformer_blocks.{i}.", cfg.num_attention_heads, x, enc, temb, dual=i in cfg.dual_attention_layers,
                           context_pre_only=i == cfg.num_layers - 1, attn_kw=attn_kw)
        if features is not None:
            features[i] = x
    sc, sh = _lin(F.silu(temb), P, "norm_out.linear").chunk(2, dim=1)
    x = _ln(x) * (1 + sc[:, None]) + sh[:, None]
    x = _lin(x, P, "proj_out").reshape(B, hp, wp, p, p, cfg.out_channels)
    return torch.einsum("nhwpqc->nchpwq", x).reshape(B, cfg.out_channels, hp * p, wp * p)


# ------------------------------------------------------------------------------------------------------------------------------
# The SD3 pipeline's loops (backbones/video_diffusion_sd3/pipelines/custom_pipeline.py) — REFERENCE-OWNED, pinned by golden G19 (the
# reference's own `reconstruction` and `video_style_transfer` run over a closed-form velocity field and the restated
# FlowMatchEuler tables).  Fixed reading of the second defect at HEAD: the undefined name `ddim_inv_latents_at_t` (:303) is the
# content inversion latent of the step (what the SD-v1.5 loop blends at the same place); in the no-mask loop that G19 pins it is
# multiplied by 0.0.
def generate_eta_values(timesteps, start_step: int, end_step: int, eta: float, eta_trend: str) -> List[float]:
    """custom_pipeline.py:18-43"""
    assert 0 <= start_step < end_step <= len(timesteps)
    out = [0.0] * len(timesteps)
    total = timesteps[start_step] - timesteps[end_step - 1]
    for i in range(start_step, end_step):
        if eta_trend == "constant":
            out[i] = eta
        elif eta_trend == "linear_increase":
            out[i] = eta * (timesteps[start_step] - timesteps[i]) / total
        elif eta_trend == "linear_decrease":
            out[i] = eta * (timesteps[i] - timesteps[end_step - 1]) / total
        else:
            raise NotImplementedError(eta_trend)
    return out


def flow_match_schedule(n: int, shift: float = 3.0, T: int = 1000):
    """diffusers 0.35.1 FlowMatchEulerDiscreteScheduler (SD3 config): __init__ shifts sigma = t/T once (sigma_max 1, sigma_min =
    shifted 1/T); set_timesteps(n) takes linspace(sigma_max, sigma_min, n) and applies the shift AGAIN; trailing 0.  Third-party,
    parity unpinned.  (`flow_match_sigmas` above is the simplified table golden G17 was generated with; it differs from this one only
    in the position of the last node.)  -> (timesteps [n], sigmas [n+1]) fp32"""
    import numpy as np
    s0 = np.linspace(1, T, T, dtype=np.float32)[::-1].copy() / T                  # __init__: fp32 table, shifted once
    s0 = shift * s0 / (1 + (shift - 1) * s0)
    smax, smin = float(s0[0]), float(s0[-1])
    s = (np.linspace(smax * T, smin * T, n) / T).astype(np.float32)               # set_timesteps: float64 linspace, fp32 from here on
    sig = torch.from_numpy((shift * s / (1 + (shift - 1) * s)).astype(np.float32))
    return sig * T, torch.cat([sig, torch.zeros(1)])


def toy_velocity(x: torch.Tensor, t1000: torch.Tensor, idx: int, frames: int) -> torch.Tensor:
    """closed-form stand-in for the transformer on the three-branch batch [content | style | stylised]: the stylised branch's
    output also reads the other two branches, so a wrong branch order or a wrong chunk shows."""
    tt = float(t1000.reshape(-1)[0]) / 1000.0
    v = torch.tanh(0.7 * x.flip(-1)) * (0.5 + tt) - 0.3 * x + 0.01 * idx
    if x.shape[0] == 3 * frames:
        v = torch.cat([v[:2 * frames], v[2 * frames:] + 0.2 * x[:frames] - 0.1 * x[frames:2 * frames]])
    return v


def toy_loop_inputs(seed: int = 1901, frames: int = 4, ch: int = 4, hw: int = 6, n: int = 50):
    g = torch.Generator().manual_seed(seed)
    content = [torch.randn(frames, ch, hw, hw, generator=g) * (0.3 + 0.7 * k / n) for k in range(n + 1)]       # k = 0: the clean latents
    style = [0.2 + 1.3 * torch.randn(frames, ch, hw, hw, generator=g) * (0.3 + 0.7 * k / n) for k in range(n + 1)]
    mask = (torch.rand(1, frames, 8 * hw, 8 * hw, generator=g) > 0.6).to(torch.uint8)
    return dict(content=content, style=style, mask=mask)


def sd3_reconstruction_loop(velocity_fn: Callable, img_latents, inversed_latents, timesteps, sigmas, eta_values, T: int = 1000):
    """custom_pipeline.py:88-118 (guidance_scale 1.0): fp32 latents, v' = v + eta (-(target - x)/t - v), Euler step.  The
    transformer is called WITHOUT joint_attention_kwargs here (:96-102): velocity_fn sees step index 0 throughout."""
    x, target = inversed_latents.float(), img_latents.float()
    for i, t in enumerate(timesteps):
        v = velocity_fn(x, t.expand(x.shape[0]), 0).float()
        tv = -(target - x) / (t / T)
        v = v + eta_values[i] * (tv - v)
        x = x + (sigmas[i + 1] - sigmas[i]) * v
    return x


def sd3_transfer_loop(velocity_fn: Callable, latents, img_latents, content_inv, style_inv, timesteps, sigmas, eta_values, mask=None,
                      T: int = 1000):
    """custom_pipeline.py:284-335.  latents [F, C, h, w]; content_inv / style_inv: lists indexed by the file label k (50 - i is read
    at step i); mask [1, F, H, W] {0,1} or None; velocity_fn(x [3F, ...], t [3F], i) -> v."""
    n = len(timesteps)
    x, target = latents, img_latents.clone()
    rm = None
    if mask is not None:
        rm = F.interpolate(mask.to(x.dtype), size=x.shape[-2:], mode="bilinear", align_corners=False).permute(1, 0, 2, 3).contiguous()
    for i, t in enumerate(timesteps):
        c_t, s_t = content_inv[50 - i].to(x.dtype), style_inv[50 - i].to(x.dtype)
        if rm is not None and i <= 0.9 * n:
            x = (1 - rm) * x + rm * c_t
        if i >= 0.8 * n and i <= 0.9 * n:
            m = rm if rm is not None else 0.0
            x = (1.0 - m) * latent_adain(x, s_t) + m * c_t          # fixed reading of `ddim_inv_latents_at_t`
        v = velocity_fn(torch.cat([c_t, s_t, x]), t.expand(3 * x.shape[0]), i)[2 * x.shape[0]:]
        tv = -(target - x) / (t / T)
        v = v + eta_values[i] * (tv - v)
        x = (x.float() + (sigmas[i + 1] - sigmas[i]) * v).to(v.dtype)
    return x

"""Oracle: functional restatement of the pseudo-3D SD UNet forward with PnP hooks.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Weights come in as a plain
``dict[str, Tensor]`` keyed by the reference's state-dict names.

Reference files restated (relative to the reference checkout):
  backbones/video_diffusion_sd/models/unet_3d_condition.py:306-443   (forward)
  backbones/video_diffusion_sd/models/unet_3d_blocks.py:129-645      (block wiring)
  backbones/video_diffusion_sd/models/attention.py:104-430           (transformer, sparse-causal attn)
  backbones/video_diffusion_sd/models/resnet.py:12-394               (PseudoConv3d, up/down, ResBlock)
  backbones/video_diffusion_sd/pnp_utils.py:7-139                    (PnP closure, AdaINs)
Third-party (diffusers 0.35.1, restated, parity unpinned by the reference):
  Attention/AttnProcessor2_0, FeedForward(GEGLU), Timesteps, TimestepEmbedding.
"""
import math
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F

SD15_CONFIG = dict(
    in_channels=4,
    out_channels=4,
    block_out_channels=(320, 640, 1280, 1280),
    layers_per_block=2,
    cross_attention_dim=768,
    attention_head_dim=8,      # used as HEAD COUNT (unet_3d_blocks.py:269-271)
    norm_num_groups=32,
    norm_eps=1e-5,
    flip_sin_to_cos=True,
    freq_shift=0,
    down_block_types=("CrossAttnDownBlockPseudo3D",) * 3 + ("DownBlockPseudo3D",),
    up_block_types=("UpBlockPseudo3D",) + ("CrossAttnUpBlockPseudo3D",) * 3,
)

TINY_CONFIG = dict(SD15_CONFIG, block_out_channels=(32, 64, 64, 64), layers_per_block=2,
                   cross_attention_dim=32, attention_head_dim=2, norm_num_groups=8)
# SD-v2.x shaped tiny config: Linear proj_in/out, per-level head counts (head_dim 16/32), wider text dim
TINY_SD2_CONFIG = dict(TINY_CONFIG, use_linear_projection=True, attention_head_dim=(2, 2, 4, 4), cross_attention_dim=64)

# SD-v2.1 (run_content_inversion_sd.py:78 of the reference keeps it as the commented default): same channel widths, Linear
# proj_in / proj_out, head_dim 64 => 5 / 10 / 20 / 20 heads (config.json's attention_head_dim is the head COUNT per level), 1024-wide
# OpenCLIP text states
SD21_CONFIG = dict(SD15_CONFIG, use_linear_projection=True, attention_head_dim=(5, 10, 20, 20), cross_attention_dim=1024)

# pnp_utils.py:104-111 — the 8 injected attn1 layers: {up_block: [attention indices]}
PNP_LAYERS = {1: [1, 2], 2: [0, 1, 2], 3: [0, 1, 2]}


# --------------------------------------------------------------------------- third-party restatements
def timestep_sinusoid(t: torch.Tensor, dim: int, flip_sin_to_cos=True, freq_shift=0) -> torch.Tensor:
    """diffusers.models.embeddings.get_timestep_embedding (scale=1, max_period=1e4); fp32."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device)
    exponent = exponent / (half - freq_shift)
    emb = t[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


def _lin(sd, name, x, bias=True):
    b = sd.get(name + ".bias") if bias else None
    return F.linear(x, sd[name + ".weight"], b)


# rows of the (b f) axis evaluated per SDPA call (None = all at once).  The result does not depend on it (rows are
# independent); the BASELINE-size device tests set it so an fp32 "math" SDPA never materialises [48,8,4096,12288] scores.
SDPA_MAX_BATCH = None


def sdpa(q, k, v, heads):
    """AttnProcessor2_0 core: split heads, softmax(QK^T/sqrt(d)) V, merge heads."""
    b, nq, c = q.shape
    d = c // heads
    q = q.view(b, nq, heads, d).transpose(1, 2)
    k = k.view(b, -1, heads, d).transpose(1, 2)
    v = v.view(b, -1, heads, d).transpose(1, 2)
    step = b if not SDPA_MAX_BATCH else int(SDPA_MAX_BATCH)
    o = torch.cat([F.scaled_dot_product_attention(q[i:i + step], k[i:i + step], v[i:i + step]) for i in range(0, b, step)])
    return o.transpose(1, 2).reshape(b, nq, c)


def plain_attention(sd, p, x, ctx, heads):
    """diffusers Attention.forward (attn2 / attn_temporal): attention.py:208-232 call sites."""
    ctx = x if ctx is None else ctx
    q = _lin(sd, p + ".to_q", x, bias=False)
    k = _lin(sd, p + ".to_k", ctx, bias=False)
    v = _lin(sd, p + ".to_v", ctx, bias=False)
    return _lin(sd, p + ".to_out.0", sdpa(q, k, v, heads))


def feed_forward_geglu(sd, p, x):
    """diffusers FeedForward(activation_fn='geglu'): Linear(C->8C), x*gelu(gate) (erf), Linear(4C->C)."""
    h = _lin(sd, p + ".net.0.proj", x)
    a, gate = h.chunk(2, dim=-1)
    return _lin(sd, p + ".net.2", a * F.gelu(gate))


# --------------------------------------------------------------------------- PnP (reference-authored)
def attention_adain(cnt, sty):
    """pnp_utils.py:114-125.  mean/UNBIASED std over tokens (dim 1) of the style; F.instance_norm on
    [c, N, C] treats N as channels => normalises over the LAST axis per (frame, token), biased var,
    eps 1e-5 (layout quirk kept on purpose)."""
    sty_mean = sty.mean(dim=1, keepdim=True)
    sty_std = sty.std(dim=1, keepdim=True)
    mu = cnt.mean(dim=-1, keepdim=True)
    var = cnt.var(dim=-1, unbiased=False, keepdim=True)
    return (cnt - mu) / torch.sqrt(var + 1e-5) * sty_std + sty_mean


def latent_adain(cnt, sty):
    """pnp_utils.py:128-139.  style stats over dims (0,3,4) => per (channel, frame), unbiased std;
    content instance_norm on 5-D => per channel over (F,H,W) jointly, biased var, eps 1e-5."""
    sty_mean = sty.mean(dim=[0, 3, 4], keepdim=True)
    sty_std = sty.std(dim=[0, 3, 4], keepdim=True)
    mu = cnt.mean(dim=[2, 3, 4], keepdim=True)
    var = cnt.var(dim=[2, 3, 4], unbiased=False, keepdim=True)
    return (cnt - mu) / torch.sqrt(var + 1e-5) * sty_std + sty_mean


def pnp_beta(idx, eta1=0.0, eta2=0.5):
    """pnp_utils.py:50"""
    return (0.9 - 0.1) / (eta1 * 50 - eta2 * 50) * (idx - eta2 * 50) + 0.1


def pnp_shift(q, k, v, idx, eta1=0.0, eta2=0.5, alpha=0.65, gamma=3.0):
    """pnp_utils.py:44-57 on [3c, N, C] tensors (returns new tensors)."""
    c = q.shape[0] // 3
    if not (idx >= eta1 and idx <= eta2 * 50):
        return q, k, v
    beta = pnp_beta(idx, eta1, eta2)
    q, k, v = q.clone(), k.clone(), v.clone()
    q[2 * c:3 * c] = alpha * q[:c] + (1 - alpha) * q[2 * c:3 * c]
    k[2 * c:3 * c] = beta * attention_adain(k[2 * c:3 * c], k[c:2 * c]) + (1 - beta) * k[c:2 * c]
    v[2 * c:3 * c] = beta * attention_adain(v[2 * c:3 * c], v[c:2 * c]) + (1 - beta) * v[c:2 * c]
    q[2 * c:3 * c] = gamma * q[2 * c:3 * c]
    return q, k, v


def sparse_causal_gather(x, clip_length, index):
    """attention.py:384-413 / pnp_utils.py:59-84: x [(b f), N, C] -> [(b f), len(index)*N, C]."""
    bf, n, c = x.shape
    x = x.view(bf // clip_length, clip_length, n, c)
    parts = []
    for ind in index:
        if isinstance(ind, str):
            if ind == "first":
                fi = [0] * clip_length
            elif ind == "last":
                fi = [clip_length - 1] * clip_length
            elif ind in ("mid", "middle"):
                fi = [int(clip_length - 1) // 2] * clip_length
            else:
                raise ValueError(ind)
            fi = torch.tensor(fi)
        else:
            fi = (torch.arange(clip_length) + ind).clip(0, clip_length - 1)
        parts.append(x[:, fi])
    x = torch.cat(parts, dim=2)
    return x.reshape(bf, -1, c)


def attn1_forward(sd, p, x, clip_length, heads, pnp: Optional[dict]):
    """attention.py:349-430 (stock, index [-1,0,'first']) or pnp_utils.py:20-100 (PnP, [-1,'first'])."""
    q = _lin(sd, p + ".to_q", x, bias=False)
    k = _lin(sd, p + ".to_k", x, bias=False)
    v = _lin(sd, p + ".to_v", x, bias=False)
    if pnp is not None:
        q, k, v = pnp_shift(q, k, v, pnp["idx"], pnp.get("eta1", 0.0), pnp.get("eta2", 0.5))
        index = [-1, "first"]
    else:
        index = [-1, 0, "first"]
    if clip_length is not None:
        k = sparse_causal_gather(k, clip_length, index)
        v = sparse_causal_gather(v, clip_length, index)
    return _lin(sd, p + ".to_out.0", sdpa(q, k, v, heads))


# --------------------------------------------------------------------------- conv / norm / resblock
def pseudo_conv3d(sd, p, x, stride=1, padding=1, exact_temporal=True):
    """resnet.py:57-80.  x [b,c,f,h,w]; 2-D conv per frame then conv1d over f (dirac => identity)."""
    b, c, f, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    y = F.conv2d(y, sd[p + ".weight"], sd.get(p + ".bias"), stride=stride, padding=padding)
    _, co, ho, wo = y.shape
    y = y.view(b, f, co, ho, wo).permute(0, 2, 1, 3, 4)
    tw = sd.get(p + ".conv_temporal.weight")
    if tw is None or not exact_temporal:
        return y.contiguous()
    z = y.permute(0, 3, 4, 1, 2).reshape(b * ho * wo, co, f)
    z = F.conv1d(z, tw, sd[p + ".conv_temporal.bias"], padding=tw.shape[-1] // 2)
    return z.view(b, ho, wo, co, f).permute(0, 3, 4, 1, 2).contiguous()


def resnet_block(sd, p, x, temb, groups, eps, exact_temporal=True):
    """resnet.py:335-394 (time_embedding_norm='default', output_scale_factor=1)."""
    h = F.group_norm(x, groups, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], eps)   # 5-D => stats span frames
    h = F.silu(h)
    h = pseudo_conv3d(sd, p + ".conv1", h, exact_temporal=exact_temporal)
    t = _lin(sd, p + ".time_emb_proj", F.silu(temb))            # [b, Cout]
    h = h + t[:, :, None, None, None]
    h = F.group_norm(h, groups, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], eps)
    h = F.silu(h)
    h = pseudo_conv3d(sd, p + ".conv2", h, exact_temporal=exact_temporal)
    if (p + ".conv_shortcut.weight") in sd:
        x = pseudo_conv3d(sd, p + ".conv_shortcut", x, padding=0, exact_temporal=exact_temporal)
    return x + h


def transformer_block(sd, p, x, ctx, clip_length, heads, pnp, exact_temporal=True):
    """attention.py:280-346 (temporal_attention_position='after_feedforward')."""
    C = x.shape[-1]
    ln = lambda n, t: F.layer_norm(t, (C,), sd[p + f".{n}.weight"], sd[p + f".{n}.bias"], 1e-5)
    x = x + attn1_forward(sd, p + ".attn1", ln("norm1", x), clip_length, heads, pnp)
    x = plain_attention(sd, p + ".attn2", ln("norm2", x), ctx, heads) + x
    x = feed_forward_geglu(sd, p + ".ff", ln("norm3", x)) + x
    if clip_length is not None:
        if exact_temporal:
            bf, d, c = x.shape
            b = bf // clip_length
            y = x.view(b, clip_length, d, c).permute(0, 2, 1, 3).reshape(b * d, clip_length, c)
            y = plain_attention(sd, p + ".attn_temporal", ln("norm_temporal", y), None, heads) + y
            x = y.view(b, d, clip_length, c).permute(0, 2, 1, 3).reshape(bf, d, c)
        else:  # zero-initialised to_out weight => output == bias (attention.py:233)
            x = x + sd[p + ".attn_temporal.to_out.0.bias"]
    return x


def transformer_model(sd, p, x, ctx, heads, groups, pnp, exact_temporal=True):
    """attention.py:104-153 (use_linear_projection=False).  x [b,c,f,h,w], ctx [b,77,D]."""
    b, c, f, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    ctx = ctx.repeat_interleave(f, 0)
    res = y
    y = F.group_norm(y, groups, sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-6)    # per frame, eps 1e-6
    lin = sd[p + ".proj_in.weight"].dim() == 2        # use_linear_projection (SD-v2.x): attention.py:125-127,142-144
    if lin:
        y = F.linear(y.permute(0, 2, 3, 1).reshape(b * f, h * w, c), sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"])
    else:
        y = F.conv2d(y, sd[p + ".proj_in.weight"], sd[p + ".proj_in.bias"])
        y = y.permute(0, 2, 3, 1).reshape(b * f, h * w, c)
    y = transformer_block(sd, p + ".transformer_blocks.0", y, ctx, f, heads, pnp, exact_temporal)
    if lin:
        y = F.linear(y, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"]).view(b * f, h, w, c).permute(0, 3, 1, 2) + res
    else:
        y = y.view(b * f, h, w, c).permute(0, 3, 1, 2)
        y = F.conv2d(y, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"]) + res
    return y.view(b, f, c, h, w).permute(0, 2, 1, 3, 4).contiguous()


def upsample(sd, p, x, exact_temporal=True):
    """resnet.py:123-175: nearest x2 per frame then 3x3 conv."""
    b, c, f, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    y = F.interpolate(y, scale_factor=2.0, mode="nearest")
    y = y.view(b, f, c, 2 * h, 2 * w).permute(0, 2, 1, 3, 4)
    return pseudo_conv3d(sd, p + ".conv", y, exact_temporal=exact_temporal)


# --------------------------------------------------------------------------- full UNet
def unet_forward(sd: Dict[str, torch.Tensor], cfg: dict, sample: torch.Tensor, timestep,
                 encoder_hidden_states: torch.Tensor, pnp_idx: Optional[int] = None,
                 eta1: float = 0.0, eta2: float = 0.5, ft_indices: Optional[List[int]] = None,
                 exact_temporal: bool = True):
    """unet_3d_condition.py:306-443.  Returns (eps [B,4,F,H,W], {up_block_index: feature [F,H,W,C]}).

    ``pnp_idx`` is the step index the PnP closure sees (register_time); ``None`` = PnP not registered.
    ``ft_indices``: up-block outputs to return as the feature dump (branch 0, permuted to [F,H,W,C],
    unet_3d_condition.py:430-436).
    """
    boc = cfg["block_out_channels"]
    hd = cfg["attention_head_dim"]
    hl = (hd,) * 4 if isinstance(hd, int) else tuple(hd)          # head COUNT per down level (unet_3d_condition.py:118-119)
    groups, eps = cfg["norm_num_groups"], cfg["norm_eps"]
    lpb = cfg["layers_per_block"]
    B = sample.shape[0]
    t = torch.as_tensor(timestep, device=sample.device).reshape(-1).expand(B)
    t_emb = timestep_sinusoid(t, boc[0], cfg["flip_sin_to_cos"], cfg["freq_shift"]).to(sample.dtype)
    emb = _lin(sd, "time_embedding.linear_2", F.silu(_lin(sd, "time_embedding.linear_1", t_emb)))
    ctx = encoder_hidden_states

    def pnp_for(up_i, attn_j):
        if pnp_idx is None or attn_j not in PNP_LAYERS.get(up_i, []):
            return None
        return dict(idx=pnp_idx, eta1=eta1, eta2=eta2)

    x = pseudo_conv3d(sd, "conv_in", sample, exact_temporal=exact_temporal)
    skips = [x]
    for i, bt in enumerate(cfg["down_block_types"]):
        p = f"down_blocks.{i}"
        for j in range(lpb):
            x = resnet_block(sd, f"{p}.resnets.{j}", x, emb, groups, eps, exact_temporal)
            if bt.startswith("CrossAttn"):
                x = transformer_model(sd, f"{p}.attentions.{j}", x, ctx, hl[i], groups, None, exact_temporal)
            skips.append(x)
        if i != len(boc) - 1:
            x = pseudo_conv3d(sd, f"{p}.downsamplers.0.conv", x, stride=2, exact_temporal=exact_temporal)
            skips.append(x)
    x = resnet_block(sd, "mid_block.resnets.0", x, emb, groups, eps, exact_temporal)
    x = transformer_model(sd, "mid_block.attentions.0", x, ctx, hl[-1], groups, None, exact_temporal)
    x = resnet_block(sd, "mid_block.resnets.1", x, emb, groups, eps, exact_temporal)
    feats = {}
    for i, bt in enumerate(cfg["up_block_types"]):
        p = f"up_blocks.{i}"
        for j in range(lpb + 1):
            x = torch.cat([x, skips.pop()], dim=1)
            x = resnet_block(sd, f"{p}.resnets.{j}", x, emb, groups, eps, exact_temporal)
            if bt.startswith("CrossAttn"):
                x = transformer_model(sd, f"{p}.attentions.{j}", x, ctx, hl[len(boc) - 1 - i], groups, pnp_for(i, j),
                                      exact_temporal)
        if i != len(boc) - 1:
            x = upsample(sd, f"{p}.upsamplers.0", x, exact_temporal)
        if ft_indices is not None and i in ft_indices:
            feats[i] = x[0].permute(1, 2, 3, 0).contiguous()
    x = F.group_norm(x, groups, sd["conv_norm_out.weight"], sd["conv_norm_out.bias"], eps)
    x = F.silu(x)
    x = pseudo_conv3d(sd, "conv_out", x, exact_temporal=exact_temporal)
    return x, feats


# --------------------------------------------------------------------------- shapes + synthetic weights
def state_dict_shapes(cfg: dict) -> Dict[str, tuple]:
    """Every key of UNetPseudo3DConditionModel.state_dict() for ``cfg`` with its shape
    (ctor: unet_3d_condition.py:49-230; 216 of 902 keys are *_temporal* for SD-v1.5)."""
    boc = cfg["block_out_channels"]
    ted = boc[0] * 4
    D = cfg["cross_attention_dim"]
    lpb = cfg["layers_per_block"]
    s: Dict[str, tuple] = {}

    def conv(p, ci, co, k):
        s[p + ".weight"] = (co, ci, k, k)
        s[p + ".bias"] = (co,)
        if k > 1:
            s[p + ".conv_temporal.weight"] = (co, co, 3)
            s[p + ".conv_temporal.bias"] = (co,)

    def lin(p, ci, co, bias=True):
        s[p + ".weight"] = (co, ci)
        if bias:
            s[p + ".bias"] = (co,)

    def norm(p, c):
        s[p + ".weight"] = (c,)
        s[p + ".bias"] = (c,)

    def res(p, ci, co):
        norm(p + ".norm1", ci)
        conv(p + ".conv1", ci, co, 3)
        lin(p + ".time_emb_proj", ted, co)
        norm(p + ".norm2", co)
        conv(p + ".conv2", co, co, 3)
        if ci != co:
            conv(p + ".conv_shortcut", ci, co, 1)

    def attn(p, c, kv):
        lin(p + ".to_q", c, c, False)
        lin(p + ".to_k", kv, c, False)
        lin(p + ".to_v", kv, c, False)
        lin(p + ".to_out.0", c, c)

    def tr(p, c):
        norm(p + ".norm", c)
        if cfg.get("use_linear_projection"):
            lin(p + ".proj_in", c, c)
            lin(p + ".proj_out", c, c)
        else:
            conv(p + ".proj_in", c, c, 1)
            conv(p + ".proj_out", c, c, 1)
        b = p + ".transformer_blocks.0"
        attn(b + ".attn1", c, c); norm(b + ".norm1", c)
        attn(b + ".attn2", c, D); norm(b + ".norm2", c)
        attn(b + ".attn_temporal", c, c); norm(b + ".norm_temporal", c)
        lin(b + ".ff.net.0.proj", c, 8 * c); lin(b + ".ff.net.2", 4 * c, c); norm(b + ".norm3", c)

    conv("conv_in", cfg["in_channels"], boc[0], 3)
    lin("time_embedding.linear_1", boc[0], ted)
    lin("time_embedding.linear_2", ted, ted)
    out = boc[0]
    for i, bt in enumerate(cfg["down_block_types"]):
        inp, out = out, boc[i]
        for j in range(lpb):
            res(f"down_blocks.{i}.resnets.{j}", inp if j == 0 else out, out)
            if bt.startswith("CrossAttn"):
                tr(f"down_blocks.{i}.attentions.{j}", out)
        if i != len(boc) - 1:
            conv(f"down_blocks.{i}.downsamplers.0.conv", out, out, 3)
    res("mid_block.resnets.0", boc[-1], boc[-1])
    tr("mid_block.attentions.0", boc[-1])
    res("mid_block.resnets.1", boc[-1], boc[-1])
    rev = list(reversed(boc))
    out = rev[0]
    for i, bt in enumerate(cfg["up_block_types"]):
        prev, out = out, rev[i]
        inp = rev[min(i + 1, len(boc) - 1)]
        for j in range(lpb + 1):
            skip = inp if j == lpb else out
            rin = prev if j == 0 else out
            res(f"up_blocks.{i}.resnets.{j}", rin + skip, out)
            if bt.startswith("CrossAttn"):
                tr(f"up_blocks.{i}.attentions.{j}", out)
        if i != len(boc) - 1:
            conv(f"up_blocks.{i}.upsamplers.0.conv", out, out, 3)
    norm("conv_norm_out", boc[0])
    conv("conv_out", boc[0], cfg["out_channels"], 3)
    return s


def synth_state_dict(cfg: dict, seed: int = 33, dtype=torch.float32, trivial_temporal: bool = True,
                     device="cpu") -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights keyed by state-dict name (the documented recipe shared by the golden
    generator, the tests and bench.py; no SD checkpoint exists on either box).

    * matrices/convs: N(0, g/fan_in) with g chosen so activations stay O(1) in fp16;
    * norm weights 1 + 0.1 N(0,1), biases 0.05 N(0,1);
    * ``*_temporal*``: exactly the reference init when ``trivial_temporal`` — conv_temporal dirac / zero
      bias (resnet.py:53-55), attn_temporal.to_out.0.weight zero (attention.py:233), its bias
      U(+-1/sqrt(C)); otherwise small random values (exercises the general path of the oracle).
    Each tensor has its own generator seeded by (seed, crc32(name)) so any subset can be regenerated.
    """
    import zlib
    out = {}
    for name, shape in state_dict_shapes(cfg).items():
        g = torch.Generator(device="cpu").manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))
        rn = lambda *sh: torch.randn(*sh, generator=g, dtype=torch.float32)
        is_norm = (".norm" in name or name.startswith("conv_norm_out")) and len(shape) == 1
        if "conv_temporal" in name and trivial_temporal:
            t = torch.zeros(shape)
            if name.endswith("weight"):
                torch.nn.init.dirac_(t)
        elif "attn_temporal.to_out.0.weight" in name and trivial_temporal:
            t = torch.zeros(shape)
        elif "attn_temporal.to_out.0.bias" in name:
            t = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(shape[0])
        elif is_norm:
            t = 1.0 + 0.1 * rn(*shape) if name.endswith("weight") else 0.05 * rn(*shape)
        elif name.endswith(".bias"):
            t = 0.02 * rn(*shape)
        else:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            gain = 1.0
            if "conv_temporal" in name:
                gain = 0.3
            t = rn(*shape) * (gain / math.sqrt(fan_in))
        out[name] = t.to(dtype).to(device)
    return out

"""Synthetic SD-shaped weights / inputs generated directly on the GPU (bench.py, smoke): no SD checkpoint,
VAE or CLIP exists on the target boxes, so the benchmark uses random-init weights of the SD-v1.5 architecture
and N(0,1) latents.  The recipe mirrors oracle.unet_ref.synth_state_dict in distribution
(fan-in scaled matrices, unit norms, reference-initialised *_temporal* layers), not bit-for-bit."""
import math

import torch

SD15_UNET_CONFIG = dict(sample_size=64, in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280),
                        layers_per_block=2, cross_attention_dim=768, attention_head_dim=8, norm_num_groups=32,
                        norm_eps=1e-5)


SD21_UNET_CONFIG = dict(SD15_UNET_CONFIG, use_linear_projection=True, attention_head_dim=(5, 10, 20, 20), cross_attention_dim=1024)


def build_unet(config=None, device="cuda", dtype=torch.float16, seed=33):
    """UNetPseudo3DConditionModel with synthetic weights, materialised straight on ``device``."""
    from .backbones.video_diffusion_sd.models.unet_3d_condition import UNetPseudo3DConditionModel
    cfg = dict(SD15_UNET_CONFIG if config is None else config)
    with torch.device("meta"):
        unet = UNetPseudo3DConditionModel(**cfg)
    unet = unet.to_empty(device=device).to(dtype)
    synth_init_(unet, seed)
    return unet.requires_grad_(False)


@torch.no_grad()
def synth_init_(unet, seed=33):
    g = torch.Generator(device=unet.device).manual_seed(seed)
    for name, p in unet.state_dict().items():
        rn = lambda: torch.randn(p.shape, generator=g, device=p.device, dtype=torch.float32)
        is_norm = (".norm" in name or name.startswith("conv_norm_out")) and p.dim() == 1
        if "conv_temporal.weight" in name:
            t = torch.zeros(p.shape, device=p.device)
            c = min(p.shape[0], p.shape[1])
            t[torch.arange(c), torch.arange(c), p.shape[2] // 2] = 1.0          # nn.init.dirac_
        elif "conv_temporal.bias" in name or "attn_temporal.to_out.0.weight" in name:
            t = torch.zeros(p.shape, device=p.device)
        elif "attn_temporal.to_out.0.bias" in name:
            t = (torch.rand(p.shape, generator=g, device=p.device) * 2 - 1) / math.sqrt(p.shape[0])
        elif is_norm:
            t = 1.0 + 0.1 * rn() if name.endswith("weight") else 0.05 * rn()
        elif name.endswith(".bias"):
            t = 0.02 * rn()
        else:
            fan_in = 1
            for d in p.shape[1:]:
                fan_in *= d
            t = rn() / math.sqrt(fan_in)
        p.copy_(t.to(p.dtype))
    unet._native_dirty = True


def synth_transfer_inputs(F=16, h=64, w=64, D=768, n=50, device="cuda", seed=1234, with_mask=False, mask_hw=512):
    """content/style inversion trajectories (n+1 latents each), text embedding [3,77,D], optional masks."""
    g = torch.Generator(device=device).manual_seed(seed)
    rn = lambda *s: torch.randn(*s, generator=g, device=device, dtype=torch.float32).to(torch.float16)
    content = [rn(1, 4, F, h, w) for _ in range(n + 1)]
    style = [(rn(1, 4, 1, h, w).float().expand(1, 4, F, h, w) + 1e-3 * rn(1, 4, F, h, w).float()).to(torch.float16).contiguous()
             for _ in range(n + 1)]
    text = rn(1, 77, D).expand(3, -1, -1).contiguous()
    mask = None
    if with_mask:
        yy, xx = torch.meshgrid(torch.arange(mask_hw, device=device), torch.arange(mask_hw, device=device), indexing="ij")
        ms = []
        for f in range(F):
            cx = mask_hw / 2 - 4 * F / 2 + 4 * f
            ms.append(((xx - cx) ** 2 + (yy - mask_hw / 2) ** 2 <= (mask_hw / 4) ** 2).to(torch.uint8))
        mask = torch.stack(ms)[None]
    return content, style, text, mask


# ---- synthetic inputs of the side workloads of bench.py (mask propagation, sliding-window warp).  Product-side generators on
# purpose: the GPU legs of bench.py take nothing out of the test-infrastructure package.
def synth_maskprop_inputs(F=16, h=64, w=64, C=640, H=512, W=512, seed=11, device="cuda"):
    """spatially coherent features [F,h,w,C] fp16 (smooth background + an object signature moving with a disc + noise) and an
    anti-aliased multi-valued 'L' first-frame mask [H,W] uint8 (numpy), the shapes of the reference's feature dump / mask PNG."""
    import numpy as np
    g = torch.Generator(device="cpu").manual_seed(seed)
    low = torch.randn(1, C, 4, 4, generator=g)
    bg = torch.nn.functional.interpolate(low, size=(h, w), mode="bilinear", align_corners=False)[0]
    sig = torch.randn(C, generator=g)
    yy, xx = np.mgrid[0:h, 0:w]
    feats = []
    for f in range(F):
        cx = w / 2.0 - 0.5 * F / 2 + 0.5 * f
        m = torch.from_numpy((((xx - cx) ** 2 + (yy - h / 2.0) ** 2) <= (h / 4.0) ** 2).astype(np.float32))
        feats.append((bg + 1.5 * sig[:, None, None] * m[None] + 0.15 * torch.randn(C, h, w, generator=g)).permute(1, 2, 0))
    Y, X = np.mgrid[0:H, 0:W]
    d = np.sqrt((X - (W / 2.0 - 4.0 * W / 512.0 * 8)) ** 2 + (Y - H / 2.0) ** 2)
    first = np.clip((H / 4.0 - d) * 64.0 + 128.0, 0, 255).astype(np.uint8)
    return torch.stack(feats).to(torch.float16).to(device), first


def synth_flow(H, W, dx, dy, seed, noise=0.25, device="cuda"):
    """analytic translation field (+ seeded noise) standing in for a RAFT output: fp32 [H,W,2]."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    f = torch.empty(H, W, 2)
    f[..., 0] = dx
    f[..., 1] = dy
    return (f + noise * torch.randn(H, W, 2, generator=g)).to(device)


SVD_VAE_CONFIG = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                      norm_num_groups=32, scaling_factor=0.18215)


def vae_state_dict(config=None, device="cuda", dtype=torch.float16, seed=7):
    """Random-init state dict with the parameter names and shapes of diffusers' AutoencoderKLTemporalDecoder (the SVD VAE; no checkpoint exists on
    the target boxes): fan-in scaled convolutions / linears, near-unit norms, mix factors around 0 (sigmoid = 0.5: both branches of every
    SpatioTemporalResBlock matter).  Key list restated from the published class (univst_amd/csrc/vae.hip header)."""
    cfg = dict(SVD_VAE_CONFIG if config is None else config)
    boc, L, lat = cfg["block_out_channels"], cfg["layers_per_block"], cfg["latent_channels"]
    g = torch.Generator(device=device).manual_seed(seed)
    sd = {}

    def rn(*shape):
        return torch.randn(*shape, generator=g, device=device, dtype=torch.float32)

    def conv(p, co, ci, *k):
        fan = ci
        for d in k:
            fan *= d
        sd[p + ".weight"] = (rn(co, ci, *k) / math.sqrt(fan)).to(dtype)
        sd[p + ".bias"] = (0.02 * rn(co)).to(dtype)

    def lin(p, co, ci):
        sd[p + ".weight"] = (rn(co, ci) / math.sqrt(ci)).to(dtype)
        sd[p + ".bias"] = (0.02 * rn(co)).to(dtype)

    def norm(p, c):
        sd[p + ".weight"] = (1.0 + 0.1 * rn(c)).to(dtype)
        sd[p + ".bias"] = (0.05 * rn(c)).to(dtype)

    def resnet(p, ci, co):
        norm(p + ".norm1", ci); conv(p + ".conv1", co, ci, 3, 3); norm(p + ".norm2", co); conv(p + ".conv2", co, co, 3, 3)
        if ci != co:
            conv(p + ".conv_shortcut", co, ci, 1, 1)

    def st(p, ci, co):
        resnet(p + ".spatial_res_block", ci, co)
        t = p + ".temporal_res_block"
        norm(t + ".norm1", co); conv(t + ".conv1", co, co, 3, 1, 1); norm(t + ".norm2", co); conv(t + ".conv2", co, co, 3, 1, 1)
        sd[p + ".time_mixer.mix_factor"] = (0.5 * rn(1)).to(dtype)

    def attn(p, c):
        norm(p + ".group_norm", c)
        for n in ("to_q", "to_k", "to_v", "to_out.0"):
            lin(p + "." + n, c, c)

    conv("encoder.conv_in", boc[0], cfg["in_channels"], 3, 3)
    ci = boc[0]
    for b in range(4):
        for l in range(L):
            resnet(f"encoder.down_blocks.{b}.resnets.{l}", ci, boc[b]); ci = boc[b]
        if b < 3:
            conv(f"encoder.down_blocks.{b}.downsamplers.0.conv", boc[b], boc[b], 3, 3)
    resnet("encoder.mid_block.resnets.0", boc[3], boc[3]); attn("encoder.mid_block.attentions.0", boc[3]); resnet("encoder.mid_block.resnets.1", boc[3], boc[3])
    norm("encoder.conv_norm_out", boc[3]); conv("encoder.conv_out", 2 * lat, boc[3], 3, 3); conv("quant_conv", 2 * lat, 2 * lat, 1, 1)
    conv("decoder.conv_in", boc[3], lat, 3, 3)
    for l in range(L):
        st(f"decoder.mid_block.resnets.{l}", boc[3], boc[3])
    attn("decoder.mid_block.attentions.0", boc[3])
    ci = boc[3]
    for b in range(4):
        co = boc[3 - b]
        for l in range(L + 1):
            st(f"decoder.up_blocks.{b}.resnets.{l}", ci, co); ci = co
        if b < 3:
            conv(f"decoder.up_blocks.{b}.upsamplers.0.conv", co, co, 3, 3)
    norm("decoder.conv_norm_out", boc[0]); conv("decoder.conv_out", cfg["out_channels"], boc[0], 3, 3)
    conv("decoder.time_conv_out", cfg["out_channels"], cfg["out_channels"], 3, 1, 1)
    return sd

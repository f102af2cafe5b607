"""Host-side mirror of backbones/video_diffusion_sd/pnp_utils.py (reference :7-139): same four names, same
arguments.  ``register_*`` only record state on the attention modules — the injected attention itself
(AdaIN-guided Q/K/V shift + [-1,'first'] sparse-causal gather + SDPA) runs inside the native UNet graph
(csrc/pnp.hip, csrc/attention.hip).  ``attention_adain`` / ``latent_adain`` call the HIP kernels directly."""
import torch

from ... import _native

# pnp_utils.py:9,104 — {up_block: [attention indices]}
PNP_LAYERS = {1: [1, 2], 2: [0, 1, 2], 3: [0, 1, 2]}


def register_time(model, t):
    """pnp_utils.py:7-15"""
    for res, blocks in PNP_LAYERS.items():
        for block in blocks:
            tb = model.unet.up_blocks[res].attentions[block].transformer_blocks[0]
            setattr(tb.attn1, "idx", t)
            setattr(tb.attn2, "idx", t)


def register_spatial_attention_pnp(model, eta1=0.0, eta2=0.5):
    """pnp_utils.py:18-111: mark the 8 decoder attn1 layers as PnP layers (window eta1 <= idx <= eta2*50)."""
    for res, blocks in PNP_LAYERS.items():
        for block in blocks:
            m = model.unet.up_blocks[res].attentions[block].transformer_blocks[0].attn1
            setattr(m, "eta1", eta1)
            setattr(m, "eta2", eta2)
            setattr(m, "_univst_native_pnp", True)


def _as_f16_cuda(t):
    """fp16, contiguous, on the GPU.  A CPU tensor (the reference helpers accept any device) is MOVED to the GPU — the kernels have
    no CPU path, and without a GPU this raises — and the callers hand the result back on the input's device."""
    if not t.is_cuda:
        if not torch.cuda.is_available():
            raise RuntimeError("univst_amd AdaIN kernels run on the GPU only (no CPU path) and no GPU is visible")
        t = t.cuda()
    return t.to(torch.float16).contiguous()


def attention_adain(cnt_feat, sty_feat, ad=True):
    """pnp_utils.py:114-125 on [c, N, C] features (stand-alone form of the fused shift kernel with
    beta = 1, alpha = 0, gamma = 1: K2 <- AdaIN(K2, K1))."""
    c, N, C = cnt_feat.shape
    cnt, sty = _as_f16_cuda(cnt_feat), _as_f16_cuda(sty_feat)
    # the fused kernel works in place on a [3c*N, 3C] QKV buffer; only K of the style (rows c..2c) and of the stylised
    # branch (rows 2c..3c) are read, so one zeroed buffer with those two slices filled is all it needs
    buf = torch.zeros(3 * c, N, 3 * C, dtype=torch.float16, device=cnt.device)
    buf[c:2 * c, :, C:2 * C] = sty
    buf[2 * c:, :, C:2 * C] = cnt
    _native.attention_adain_shift_(buf.view(3 * c * N, 3 * C), c, N, C, 0.0, 1.0, 1.0)
    return buf[2 * c:, :, C:2 * C].to(device=cnt_feat.device, dtype=cnt_feat.dtype)


def latent_adain(cnt_feat, sty_feat, ad=True):
    """pnp_utils.py:128-139 on [1, C, F, h, w] latents."""
    if cnt_feat.shape[0] != 1:
        raise NotImplementedError("latent_adain: batch 1 only (as used by the pipeline)")
    return _native.latent_adain(_as_f16_cuda(cnt_feat), _as_f16_cuda(sty_feat)).to(device=cnt_feat.device, dtype=cnt_feat.dtype)

"""Mirror of backbones/video_diffusion_sd3/pnp_utils.py of the reference — the SD3 / SD3.5 plugin's attention processors and AdaIN
helpers — on the native kernels (first vertical slice of SURVEY §8f-4; csrc/sd3.hip).  Same class / function names and call
signatures: the processors are handed diffusers' ``Attention`` module (they read its parameters, never call its layers) and return
what the reference's ``__call__`` returns.  The MM-DiT that would call them (diffusers' SD3Transformer2DModel behind
``models/transformer_3D_model.py``) is third-party and not part of this build: there is no SD3 backbone here yet.

Known defect of the reference kept visible: ``AttentionShiftProcessor`` reads ``self.thresh2`` (pnp_utils.py:186), which nothing
sets — the reference path raises AttributeError at HEAD.  This mirror implements the documented fixed reading ``thresh2 == eta2``
(the only value for which beta runs 0.9 -> 0.1 across the window like the SD-v1.5 closure); oracle/sd3_ref.py and golden G16 pin it.
"""
import torch

from ... import _native


def _params(attn):
    """the attention module's parameters (to_q.weight, ..., to_add_out.bias) as the native call wants them: fp16 device tensors with the
    q | k | v projections fused (``_native.Sd3AttnParams``), cached on the module and rebuilt when a parameter is re-allocated, edited in
    place through the tensor (version counter) or re-registered."""
    ps = list(attn.parameters())
    key = tuple((p.data_ptr(), p._version) for p in ps)
    cached = attn.__dict__.get("_uv_native_params")
    if cached is None or cached[0] != key:
        dev = ps[0].device if ps and ps[0].device.type == "cuda" else torch.device("cuda")
        cached = (key, _native.Sd3AttnParams(attn.state_dict(), dev))
        attn.__dict__["_uv_native_params"] = cached
    return cached[1]


def _run(attn, hidden_states, encoder_hidden_states, shift, idx, eta1, eta2, clip_length=16, fuse=None):
    dt, dev = hidden_states.dtype, hidden_states.device
    if dev.type != "cuda":
        raise RuntimeError("univst_amd SD3 processors run on the GPU only (no CPU path): move the tensors to cuda")
    hid = hidden_states.to(torch.float16).contiguous()
    enc = None if encoder_hidden_states is None else encoder_hidden_states.to(torch.float16).contiguous()
    eps = getattr(getattr(attn, "norm_q", None), "eps", None) or 1e-6
    comm = None
    shard = getattr(attn, "_uv_frame_shard", None)            # univst_amd.parallel.Sd3FrameShard.attach: this rank's frames of every branch
    if shard is not None and shard.world > 1:
        if clip_length == 0:
            raise RuntimeError("a frame-sharded MM-DiT needs the cross-frame processors (CrossFrameProcessor / AttentionShiftProcessor)")
        clip_length, comm = shard.local, shard.comm.ptr
    out = _native.sd3_joint_attention(_params(attn), hid, enc, attn.heads, clip_length=clip_length, shift=shift, idx=idx, eta1=eta1, eta2=eta2,
                                      rms_eps=float(eps), fuse=fuse, comm=comm)
    if enc is None:
        return out.to(dt)
    return out[0].to(dt), out[1].to(dt)


class CrossFrameProcessor:
    """pnp_utils.py:9-131: joint attention whose image keys / values are those of frames ['first', f-1, f] of the clip.
    ``fused_gated_residual`` (addition, keyword only): the native MM-DiT block hands its gated residual to processors that
    advertise ``supports_fused_gated_residual``; the result is then hidden + gate * attention output (one epilogue, no extra pass)."""
    supports_fused_gated_residual = True
    clip_length = 16        # pnp_utils.py:26 hard-codes 16 frames per clip inside __call__; an attribute here so that other clip lengths can be set

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, idx=-1, *args, fused_gated_residual=None, **kwargs):
        if attention_mask is not None:
            raise NotImplementedError("attention_mask is not used on the UniVST path")
        return _run(attn, hidden_states, encoder_hidden_states, False, idx, 0.0, 0.0, clip_length=self.clip_length, fuse=fused_gated_residual)


class AttentionShiftProcessor:
    """pnp_utils.py:134-271: the same with the AdaIN-guided shift of the stylised branch inside eta1*50 <= idx <= eta2*50."""

    supports_fused_gated_residual = True
    clip_length = 16        # pnp_utils.py:147 (same hard-coded 16)

    def __init__(self, eta1, eta2):
        self.eta1, self.eta2 = eta1, eta2
        self.thresh2 = eta2          # the attribute the reference forgets to set (see the module docstring)

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, idx=-1, *args, fused_gated_residual=None, **kwargs):
        if attention_mask is not None:
            raise NotImplementedError("attention_mask is not used on the UniVST path")
        return _run(attn, hidden_states, encoder_hidden_states, True, idx, self.eta1, self.thresh2, clip_length=self.clip_length, fuse=fused_gated_residual)


def register_spatial_attention_pnp(model, eta1=0.0, eta2=0.6):
    """pnp_utils.py:276-286."""
    new_procs = {}
    for name, processor in model.transformer.attn_processors.items():
        new_procs[name] = AttentionShiftProcessor(eta1, eta2) if "attn" in name else processor
    model.transformer.set_attn_processor(new_procs)


def attention_adain(cnt_feat, sty_feat, ad=True):
    """pnp_utils.py:289-302 on [B, heads, N, d]: F.instance_norm over (N, d) jointly per (batch, head), re-coloured with the style's
    per-channel mean / unbiased std over N.  The shift kernel with alpha = 0, beta = 1, gamma = 1 (pure AdaIN of K) on a scratch
    q | k | v buffer whose style / stylised K slices are filled."""
    B, heads, N, d = cnt_feat.shape
    dev = cnt_feat.device
    C = heads * d
    buf = torch.zeros(3 * B, N, 3 * C, dtype=torch.float16, device="cuda")
    buf[B:2 * B, :, C:2 * C] = sty_feat.permute(0, 2, 1, 3).reshape(B, N, C).to("cuda")
    buf[2 * B:, :, C:2 * C] = cnt_feat.permute(0, 2, 1, 3).reshape(B, N, C).to("cuda")
    _native.sd3_adain_shift_(buf.view(3 * B * N, 3 * C), B, N, C, heads, 0.0, 1.0, 1.0)
    return buf[2 * B:, :, C:2 * C].reshape(B, N, heads, d).permute(0, 2, 1, 3).to(device=dev, dtype=cnt_feat.dtype)


def latent_adain(cnt_feat, sty_feat, ad=True):
    """pnp_utils.py:305-316 on [B, C, H, W] (frames are the batch): per (frame, channel) statistics over (H, W) — the SD-v1.5
    latent_adain kernel applied frame by frame ([1, C, 1, H, W] views)."""
    dev, dt = cnt_feat.device, cnt_feat.dtype
    c = cnt_feat.to(device="cuda", dtype=torch.float16)
    s = sty_feat.to(device="cuda", dtype=torch.float16)
    out = torch.empty_like(c)
    for f in range(c.shape[0]):
        out[f] = _native.latent_adain(c[f][None, :, None].contiguous(), s[f][None, :, None].contiguous())[0, :, 0]
    return out.to(device=dev, dtype=dt)

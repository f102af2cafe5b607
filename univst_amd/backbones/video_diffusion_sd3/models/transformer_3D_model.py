"""Mirror of backbones/video_diffusion_sd3/models/transformer_3D_model.py of the reference: ``CustomSD3Transformer2DModel`` — the
MM-DiT of SD3 / SD3.5 with the reference's extra forward arguments (``idx``, ``ft_indices``, ``ft_timesteps``, ``ft_path``,
transformer_3D_model.py:12-113) — with every tensor operation on the native HIP kernels (include/univst.h, csrc/sd3.hip + the GEMM /
attention kernels of the SD-v1.5 path).

The reference subclasses diffusers' ``SD3Transformer2DModel`` and only re-states its forward; the layers themselves (PatchEmbed,
CombinedTimestepTextProjEmbeddings, JointTransformerBlock, AdaLayerNormZero / ZeroX / Continuous, FeedForward, Attention) are
diffusers 0.35.1 code, which is absent from the reference tree and from both boxes.  They are RE-STATED here from their published
definitions — THIRD-PARTY, PARITY UNPINNED (oracle/sd3_ref.py ``sd3_transformer`` is the same reading in torch fp32 and is what the
GPU tests compare against; only the attention processors it calls are pinned to reference code, goldens G15-G18).

The module tree carries diffusers' parameter names (``pos_embed.proj.weight``, ``time_text_embed.timestep_embedder.linear_1.weight``,
``transformer_blocks.3.attn.add_q_proj.weight``, ``transformer_blocks.3.ff.net.0.proj.weight``, ``norm_out.linear.weight`` ...), so
``load_state_dict`` takes the ``transformer/`` checkpoint of ``stabilityai/stable-diffusion-3.5-medium`` as is.  The ``nn.Linear`` /
``nn.Conv2d`` children are parameter containers only: nothing here calls a torch layer, and a CPU tensor is an error (no CPU path).
"""
import os
from types import SimpleNamespace

import torch
from torch import nn

from .... import _native


class Transformer2DModelOutput:
    def __init__(self, sample):
        self.sample = sample


def _w(layer):
    """(weight, bias) of a parameter container, checked: the kernels take fp16 device pointers."""
    w, b = layer.weight, getattr(layer, "bias", None)
    if not (w.is_cuda and w.dtype == torch.float16):
        raise RuntimeError("univst_amd SD3 transformer runs in fp16 on the GPU only: call .half().cuda() on the model first")
    return w.detach(), (None if b is None else b.detach())


def _linear(x2d, layer, residual=None, out=None):
    w, b = _w(layer)
    return _native.linear(x2d, w.reshape(w.shape[0], -1), b, residual=residual, out=out)


class RMSNorm(nn.Module):
    def __init__(self, dim, eps):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))


class Attention(nn.Module):
    """parameter container with diffusers' Attention attribute names; ``forward`` hands itself to its processor exactly as
    diffusers does (the processors of pnp_utils.py read ``state_dict()``, ``heads``, ``norm_q.eps``, ``context_pre_only``)."""

    def __init__(self, dim, heads, dim_head, joint, context_pre_only, qk_norm, processor):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.inner_dim, self.context_pre_only = heads, inner, context_pre_only
        self.to_q, self.to_k, self.to_v = nn.Linear(dim, inner), nn.Linear(dim, inner), nn.Linear(dim, inner)
        self.to_out = nn.ModuleList([nn.Linear(inner, dim), nn.Identity()])
        if qk_norm == "rms_norm":
            self.norm_q, self.norm_k = RMSNorm(dim_head, 1e-6), RMSNorm(dim_head, 1e-6)
        if joint:
            self.add_q_proj, self.add_k_proj, self.add_v_proj = nn.Linear(dim, inner), nn.Linear(dim, inner), nn.Linear(dim, inner)
            if not context_pre_only:
                self.to_add_out = nn.Linear(inner, dim)
            if qk_norm == "rms_norm":
                self.norm_added_q, self.norm_added_k = RMSNorm(dim_head, 1e-6), RMSNorm(dim_head, 1e-6)
        self.processor = processor

    def set_processor(self, processor):
        self.processor = processor

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **kwargs):
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states, attention_mask=attention_mask, **kwargs)


class JointAttnProcessor2_0:
    """diffusers' stock joint attention (no cross-frame keys): what the blocks run until a UniVST processor is registered."""
    supports_fused_gated_residual = True

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, *args, fused_gated_residual=None, **kwargs):
        from ..pnp_utils import _run
        if attention_mask is not None:
            raise NotImplementedError("attention_mask is not used on the UniVST path")
        return _run(attn, hidden_states, encoder_hidden_states, False, -1, 0.0, 0.0, clip_length=0, fuse=fused_gated_residual)


class _AdaNorm(nn.Module):
    """AdaLayerNormZero (6 chunks) / ZeroX (9) / Continuous (2): only the conditioning linear has parameters."""

    def __init__(self, dim, chunks):
        super().__init__()
        self.linear = nn.Linear(dim, chunks * dim)


class _GELU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner)


class FeedForward(nn.Module):
    """FeedForward(dim, activation_fn="gelu-approximate"): net.0.proj -> GELU(tanh) -> net.2"""

    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([_GELU(dim, 4 * dim), nn.Identity(), nn.Linear(4 * dim, dim)])

    def forward(self, x, residual=None, gate=None):
        """residual / gate (optional): returns residual + gate[:, None] * ff(x) from the second linear's epilogue; GELU(tanh) sits in the
        first linear's epilogue — the 4x-wide intermediate is written once and read once."""
        B, N, D = x.shape
        w1, b1 = _w(self.net[0].proj)
        w2, b2 = _w(self.net[2])
        h = _native.linear_gated(x.reshape(B * N, D), w1, b1, act=_native.ACT_GELU_TANH)
        res = None if residual is None else residual.reshape(B * N, D)
        return _native.linear_gated(h, w2, b2, residual=res, gate=gate, rows_per_gate=N).view(B, N, D)


class JointTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, context_pre_only, qk_norm, use_dual_attention):
        super().__init__()
        self.dim, self.context_pre_only, self.use_dual_attention = dim, context_pre_only, use_dual_attention
        self.norm1 = _AdaNorm(dim, 9 if use_dual_attention else 6)
        self.norm1_context = _AdaNorm(dim, 2 if context_pre_only else 6)
        self.attn = Attention(dim, heads, dim_head, True, context_pre_only, qk_norm, JointAttnProcessor2_0())
        self.attn2 = Attention(dim, heads, dim_head, False, None, qk_norm, JointAttnProcessor2_0()) if use_dual_attention else None
        self.ff = FeedForward(dim)
        self.ff_context = None if context_pre_only else FeedForward(dim)

    def forward(self, hidden_states, encoder_hidden_states, temb, joint_attention_kwargs=None):
        """``temb`` is SiLU(conditioning) already: every adaLN layer starts with the same SiLU, done once per forward."""
        kw = joint_attention_kwargs or {}
        D = self.dim
        emb = _linear(temb, self.norm1.linear)                      # shift_msa scale_msa gate_msa shift_mlp scale_mlp gate_mlp [shift2 scale2 gate2]
        ch = lambda t, i: t[:, i * D:(i + 1) * D]                   # noqa: E731
        if self.use_dual_attention:
            nh, nh2 = _native.adaln_modulate(hidden_states, ch(emb, 1), ch(emb, 0), 1e-6, scale2=ch(emb, 7), shift2=ch(emb, 6))
        else:
            nh = _native.adaln_modulate(hidden_states, ch(emb, 1), ch(emb, 0), 1e-6)
        cemb = _linear(temb, self.norm1_context.linear)
        if self.context_pre_only:                                   # AdaLayerNormContinuous: (scale, shift)
            ne = _native.adaln_modulate(encoder_hidden_states, ch(cemb, 0), ch(cemb, 1), 1e-6)
        else:
            ne = _native.adaln_modulate(encoder_hidden_states, ch(cemb, 1), ch(cemb, 0), 1e-6)
        fused = getattr(self.attn.processor, "supports_fused_gated_residual", False)
        if fused:                                                   # hidden + gate_msa * attn (and the text stream's) in the out-projections' epilogue
            fg = dict(res_img=hidden_states, gate_img=ch(emb, 2))
            if not self.context_pre_only:
                fg.update(res_txt=encoder_hidden_states, gate_txt=ch(cemb, 2))
            hidden_states, enc = self.attn(hidden_states=nh, encoder_hidden_states=ne, fused_gated_residual=fg, **kw)
        else:
            a_img, a_txt = self.attn(hidden_states=nh, encoder_hidden_states=ne, **kw)
            hidden_states = _native.gate_residual(hidden_states, ch(emb, 2), a_img)
            enc = None if self.context_pre_only else _native.gate_residual(encoder_hidden_states, ch(cemb, 2), a_txt)
        if self.use_dual_attention:
            if getattr(self.attn2.processor, "supports_fused_gated_residual", False):
                hidden_states = self.attn2(hidden_states=nh2, fused_gated_residual=dict(res_img=hidden_states, gate_img=ch(emb, 8)), **kw)
            else:
                hidden_states = _native.gate_residual(hidden_states, ch(emb, 8), self.attn2(hidden_states=nh2, **kw))
        n2 = _native.adaln_modulate(hidden_states, ch(emb, 4), ch(emb, 3), 1e-6)
        hidden_states = self.ff(n2, residual=hidden_states, gate=ch(emb, 5))
        if self.context_pre_only:
            return None, hidden_states
        n2c = _native.adaln_modulate(enc, ch(cemb, 4), ch(cemb, 3), 1e-6)
        enc = self.ff_context(n2c, residual=enc, gate=ch(cemb, 5))
        return enc, hidden_states


def _sincos_1d(dim, pos):
    omega = 1.0 / 10000 ** (torch.arange(dim // 2, dtype=torch.float64) / (dim / 2.0))
    out = pos.reshape(-1)[:, None] * omega[None]
    return torch.cat([out.sin(), out.cos()], dim=1)


def get_2d_sincos_pos_embed(dim, grid_size, base_size, interpolation_scale=1.0):
    """diffusers embeddings.get_2d_sincos_pos_embed (restated): [grid_size^2, dim]; only used for a randomly initialised model —
    a checkpoint carries the table as the persistent buffer ``pos_embed.pos_embed``."""
    gh = torch.arange(grid_size, dtype=torch.float64) / (grid_size / base_size) / interpolation_scale
    gw = torch.arange(grid_size, dtype=torch.float64) / (grid_size / base_size) / interpolation_scale
    ww, hh = torch.meshgrid(gw, gh, indexing="xy")          # np.meshgrid(grid_w, grid_h): w goes first
    return torch.cat([_sincos_1d(dim // 2, ww), _sincos_1d(dim // 2, hh)], dim=1).float()


class PatchEmbed(nn.Module):
    def __init__(self, sample_size, patch_size, in_channels, dim, pos_embed_max_size):
        super().__init__()
        self.patch_size, self.pos_embed_max_size = patch_size, pos_embed_max_size
        self.proj = nn.Conv2d(in_channels, dim, kernel_size=patch_size, stride=patch_size)
        self.register_buffer("pos_embed", get_2d_sincos_pos_embed(dim, pos_embed_max_size, sample_size // patch_size)[None], persistent=True)
        self._crop = {}

    def cropped_pos_embed(self, hp, wp):
        """centre crop of the [max, max] table (PatchEmbed.cropped_pos_embed), cached per latent size, fp16 [hp*wp, dim]."""
        key = (hp, wp, self.pos_embed.data_ptr(), self.pos_embed._version)
        if key not in self._crop:
            m = self.pos_embed_max_size
            if hp > m or wp > m:
                raise ValueError(f"latent of {hp}x{wp} patches exceeds pos_embed_max_size {m}")
            top, left = (m - hp) // 2, (m - wp) // 2
            t = self.pos_embed.reshape(m, m, -1)[top:top + hp, left:left + wp].reshape(hp * wp, -1)
            self._crop = {key: t.to(torch.float16).contiguous()}
        return self._crop[key]

    def forward(self, latent):
        B, Cc, H, W = latent.shape
        p = self.patch_size
        hp, wp = H // p, W // p
        rows = _native.sd3_patchify(latent, p)
        pos = self.cropped_pos_embed(hp, wp)
        out = torch.empty(B, hp * wp, self.proj.weight.shape[0], device=latent.device, dtype=torch.float16)
        for b in range(B):                                           # the positional table rides in as the linear's residual operand
            _linear(rows[b * hp * wp:(b + 1) * hp * wp], self.proj, residual=pos, out=out[b])
        return out


class _MLP2(nn.Module):
    """TimestepEmbedding / PixArtAlphaTextProjection(act_fn="silu"): linear_1 -> SiLU -> linear_2"""

    def __init__(self, din, dim):
        super().__init__()
        self.linear_1, self.linear_2 = nn.Linear(din, dim), nn.Linear(dim, dim)

    def forward(self, x, residual=None):
        h = _linear(x, self.linear_1)
        _native.activation(h, _native.ACT_SILU, out=h)
        return _linear(h, self.linear_2, residual=residual)


class CombinedTimestepTextProjEmbeddings(nn.Module):
    def __init__(self, dim, pooled_dim):
        super().__init__()
        self.timestep_embedder = _MLP2(256, dim)
        self.text_embedder = _MLP2(pooled_dim, dim)

    def forward(self, timestep, pooled_projection):
        t_proj = _native.timestep_embedding(timestep, 256, flip_sin_to_cos=True, downscale_freq_shift=0.0)
        t_emb = self.timestep_embedder(t_proj)
        return self.text_embedder(pooled_projection, residual=t_emb)          # conditioning = timesteps_emb + pooled_projections


SD35_MEDIUM_CONFIG = dict(sample_size=128, patch_size=2, in_channels=16, num_layers=24, attention_head_dim=64, num_attention_heads=24,
                          joint_attention_dim=4096, caption_projection_dim=1536, pooled_projection_dim=2048, out_channels=16,
                          pos_embed_max_size=384, dual_attention_layers=tuple(range(13)), qk_norm="rms_norm")


class CustomSD3Transformer2DModel(nn.Module):
    def __init__(self, sample_size=128, patch_size=2, in_channels=16, num_layers=18, attention_head_dim=64, num_attention_heads=18,
                 joint_attention_dim=4096, caption_projection_dim=1152, pooled_projection_dim=2048, out_channels=16, pos_embed_max_size=96,
                 dual_attention_layers=(), qk_norm=None):
        super().__init__()
        self.config = SimpleNamespace(sample_size=sample_size, patch_size=patch_size, in_channels=in_channels, num_layers=num_layers,
                                      attention_head_dim=attention_head_dim, num_attention_heads=num_attention_heads,
                                      joint_attention_dim=joint_attention_dim, caption_projection_dim=caption_projection_dim,
                                      pooled_projection_dim=pooled_projection_dim, out_channels=out_channels,
                                      pos_embed_max_size=pos_embed_max_size, dual_attention_layers=tuple(dual_attention_layers), qk_norm=qk_norm)
        self.out_channels = out_channels
        self.inner_dim = dim = num_attention_heads * attention_head_dim
        if caption_projection_dim != dim:
            raise ValueError("caption_projection_dim must equal num_attention_heads * attention_head_dim (as in every SD3 checkpoint)")
        self.pos_embed = PatchEmbed(sample_size, patch_size, in_channels, dim, pos_embed_max_size)
        self.time_text_embed = CombinedTimestepTextProjEmbeddings(dim, pooled_projection_dim)
        self.context_embedder = nn.Linear(joint_attention_dim, caption_projection_dim)
        self.transformer_blocks = nn.ModuleList([
            JointTransformerBlock(dim, num_attention_heads, attention_head_dim, context_pre_only=(i == num_layers - 1), qk_norm=qk_norm,
                                  use_dual_attention=(i in self.config.dual_attention_layers)) for i in range(num_layers)])
        self.norm_out = _AdaNorm(dim, 2)
        self.proj_out = nn.Linear(dim, patch_size * patch_size * out_channels)

    @property
    def dtype(self):
        return self.proj_out.weight.dtype

    @property
    def device(self):
        return self.proj_out.weight.device

    # ------------------------------------------------------------------ diffusers' processor registry (pnp_utils.py:276-286 uses it)
    @property
    def attn_processors(self):
        out = {}
        for i, blk in enumerate(self.transformer_blocks):
            out[f"transformer_blocks.{i}.attn.processor"] = blk.attn.processor
            if blk.attn2 is not None:
                out[f"transformer_blocks.{i}.attn2.processor"] = blk.attn2.processor
        return out

    def set_attn_processor(self, processor):
        names = list(self.attn_processors)
        if isinstance(processor, dict) and len(processor) != len(names):
            raise ValueError(f"A dict of processors was passed, but the number of processors {len(processor)} does not match the number "
                             f"of attention layers: {len(names)}.")
        for name in names:
            _, i, which, _ = name.split(".")
            getattr(self.transformer_blocks[int(i)], which).set_processor(processor[name] if isinstance(processor, dict) else processor)

    # ------------------------------------------------------------------ transformer_3D_model.py:13-113
    @torch.no_grad()
    def forward(self, hidden_states, encoder_hidden_states=None, pooled_projections=None, timestep=None, block_controlnet_hidden_states=None,
                joint_attention_kwargs=None, return_dict=True, skip_layers=None, idx=0, ft_indices=None, ft_timesteps=None, ft_path=None):
        if block_controlnet_hidden_states is not None:
            raise NotImplementedError("ControlNet residuals are not on the UniVST path")
        kw = dict(joint_attention_kwargs) if joint_attention_kwargs is not None else {}
        kw.pop("scale", None)                                        # LoRA scale: no PEFT layers here
        if "ip_adapter_image_embeds" in kw:
            raise NotImplementedError("IP-Adapter inputs are not on the UniVST path")
        if hidden_states.device.type != "cuda":
            raise RuntimeError("univst_amd SD3 transformer runs on the GPU only (no CPU path)")
        in_dtype = hidden_states.dtype
        x = hidden_states.to(torch.float16).contiguous()
        B, _, height, width = x.shape
        p = self.config.patch_size
        h = self.pos_embed(x)                                        # [B, N, D] incl. positional table
        # a batch-1 prompt against F frames (rf_inversion / rf_solver / reconstruction pass it so): torch broadcasting in diffusers
        bc = lambda t: t.expand(B, *t.shape[1:]) if t.shape[0] == 1 and B > 1 else t          # noqa: E731
        cond = self.time_text_embed(timestep.reshape(-1).expand(B), bc(pooled_projections).to(torch.float16).contiguous())
        temb = _native.activation(cond, _native.ACT_SILU)
        enc_in = bc(encoder_hidden_states).to(torch.float16).contiguous()
        T = enc_in.shape[1]
        enc = _linear(enc_in.reshape(B * T, -1), self.context_embedder).view(B, T, -1)
        for index_block, block in enumerate(self.transformer_blocks):
            if skip_layers is None or index_block not in skip_layers:
                enc, h = block(h, enc, temb, kw)
            if ft_indices is not None and ft_timesteps and ft_path is not None:          # transformer_3D_model.py:76-82
                if index_block in ft_indices and idx in ft_timesteps:
                    save_path = os.path.join(ft_path, f"inversion_feature_map_{index_block}_block_{idx}_step.pt")
                    print(f"save feature map at: {save_path}")
                    torch.save(h.view(B, height // 2, width // 2, -1).detach().to(in_dtype), save_path)
        emb = _linear(temb, self.norm_out.linear)                    # AdaLayerNormContinuous: (scale, shift)
        D = self.inner_dim
        h = _native.adaln_modulate(h, emb[:, :D], emb[:, D:], 1e-6)
        rows = _linear(h.view(-1, D), self.proj_out)
        out = _native.sd3_unpatchify(rows, B, self.out_channels, height, width, p).to(in_dtype)
        if not return_dict:
            return (out,)
        return Transformer2DModelOutput(sample=out)


def sd35_medium(**overrides):
    """the MMDiT-X of stabilityai/stable-diffusion-3.5-medium (the reference's default checkpoint, src/sd3/run_*_sd3.py:103)."""
    return CustomSD3Transformer2DModel(**{**SD35_MEDIUM_CONFIG, **overrides})

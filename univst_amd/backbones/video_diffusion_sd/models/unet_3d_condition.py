"""Host-side mirror of the reference plugin class ``UNetPseudo3DConditionModel``
(backbones/video_diffusion_sd/models/unet_3d_condition.py:44-509 of the reference).

What is kept, so that the class drops into the reference's scripts:
  * the module tree and therefore every ``state_dict()`` key (incl. the 216 ``*_temporal*`` keys), built from
    the same ``torch.nn`` layer types in the same construction order, so that after ``seed_everything(s)`` the
    never-loaded temporal parameters get the same random init as in the reference (SURVEY "seed-dependent bias");
  * ``from_2d_model`` / ``load_2d_state_dict`` semantics and error behaviour (:445-509);
  * ``forward(sample, timestep, encoder_hidden_states, ..., ft_indices, ft_timesteps, ft_path)`` returning an
    object with ``.sample`` (and ``["sample"]``), including the feature dump file (:430-436);
  * ``unet.up_blocks[r].attentions[b].transformer_blocks[0].attn1/attn2`` modules that accept the ``idx``,
    ``eta1``, ``eta2`` attributes poked by ``pnp_utils`` (the native graph reads them back at forward time).

What is different: the modules carry parameters only.  All arithmetic of ``forward`` runs in the hand-written
gfx950 kernels behind ``libunivst_hip.so`` (one C-ABI call per step); there is no PyTorch fallback.
"""
import glob
import json
import os
import threading
from dataclasses import dataclass
from typing import List, Optional, Tuple, Union

import torch
import torch.nn as nn

from .... import _native


class _ParamOnly(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError(f"{type(self).__name__} holds parameters only; it runs as part of the native UNet graph "
                           "(UNetPseudo3DConditionModel.forward)")


class PseudoConv3d(nn.Conv2d):
    """resnet.py:12-56 (ctor only): Conv2d + dirac-initialised temporal Conv1d when kernel_size > 1."""

    def __init__(self, in_channels, out_channels, kernel_size, **kwargs):
        super().__init__(in_channels=in_channels, out_channels=out_channels, kernel_size=kernel_size, **kwargs)
        self.conv_temporal = (nn.Conv1d(out_channels, out_channels, kernel_size=kernel_size, padding=kernel_size // 2)
                              if kernel_size > 1 else None)
        if self.conv_temporal is not None:
            nn.init.dirac_(self.conv_temporal.weight.data)
            nn.init.zeros_(self.conv_temporal.bias.data)

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("PseudoConv3d runs as part of the native UNet graph")


class ResnetBlockPseudo3D(_ParamOnly):
    """resnet.py:239-333 (ctor)."""

    def __init__(self, in_channels, out_channels, temb_channels, groups, eps):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.norm1 = nn.GroupNorm(num_groups=groups, num_channels=in_channels, eps=eps, affine=True)
        self.conv1 = PseudoConv3d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(num_groups=groups, num_channels=out_channels, eps=eps, affine=True)
        self.dropout = nn.Dropout(0.0)
        self.conv2 = PseudoConv3d(out_channels, out_channels, kernel_size=3, stride=1, padding=1)
        self.conv_shortcut = None
        if in_channels != out_channels:
            self.conv_shortcut = PseudoConv3d(in_channels, out_channels, kernel_size=1, stride=1, padding=0)


class Attention(_ParamOnly):
    """Parameter layout of diffusers.models.attention.Attention / SparseCausalAttention (attention.py:349)."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        kv = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=False)
        self.to_k = nn.Linear(kv, inner, bias=False)
        self.to_v = nn.Linear(kv, inner, bias=False)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(0.0)])


class SparseCausalAttention(Attention):
    pass


class GEGLU(_ParamOnly):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2)


class FeedForward(_ParamOnly):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Dropout(0.0), nn.Linear(dim * 4, dim)])


class SpatioTemporalTransformerBlock(_ParamOnly):
    """attention.py:156-243 (ctor)."""

    def __init__(self, dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.attn1 = SparseCausalAttention(query_dim=dim, heads=heads, dim_head=dim_head)
        self.norm1 = nn.LayerNorm(dim)
        self.attn2 = Attention(query_dim=dim, cross_attention_dim=cross_attention_dim, heads=heads, dim_head=dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn_temporal = Attention(query_dim=dim, heads=heads, dim_head=dim_head)
        nn.init.zeros_(self.attn_temporal.to_out[0].weight.data)
        self.norm_temporal = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)
        self.norm3 = nn.LayerNorm(dim)


# construction-time switch read by SpatioTemporalTransformerModel (thread-local: the loopback tests build one UNet per thread)
_BUILD = threading.local()


class SpatioTemporalTransformerModel(_ParamOnly):
    """attention.py:40-102 (ctor; 1x1-conv projections for SD-v1.x, Linear for SD-v2.x use_linear_projection)."""

    def __init__(self, heads, dim_head, in_channels, cross_attention_dim, groups):
        super().__init__()
        inner = heads * dim_head
        lin = getattr(_BUILD, "use_linear_projection", False)      # set by the UNet constructor of THIS thread
        self.norm = nn.GroupNorm(num_groups=groups, num_channels=in_channels, eps=1e-6, affine=True)
        self.proj_in = nn.Linear(in_channels, inner) if lin else nn.Conv2d(in_channels, inner, kernel_size=1, stride=1, padding=0)
        self.transformer_blocks = nn.ModuleList([SpatioTemporalTransformerBlock(inner, heads, dim_head, cross_attention_dim)])
        self.proj_out = nn.Linear(in_channels, inner) if lin else nn.Conv2d(inner, in_channels, kernel_size=1, stride=1, padding=0)


class DownsamplePseudo3D(_ParamOnly):
    def __init__(self, channels):
        super().__init__()
        self.conv = PseudoConv3d(channels, channels, 3, stride=2, padding=1)


class UpsamplePseudo3D(_ParamOnly):
    def __init__(self, channels):
        super().__init__()
        self.conv = PseudoConv3d(channels, channels, 3, padding=1)


class _Block(_ParamOnly):
    has_cross_attention = False


class CrossAttnDownBlockPseudo3D(_Block):
    has_cross_attention = True

    def __init__(self, in_c, out_c, temb, layers, groups, eps, heads, xdim, add_downsample):
        super().__init__()
        resnets, attns = [], []
        for i in range(layers):
            resnets.append(ResnetBlockPseudo3D(in_c if i == 0 else out_c, out_c, temb, groups, eps))
            attns.append(SpatioTemporalTransformerModel(heads, out_c // heads, out_c, xdim, groups))
        self.attentions = nn.ModuleList(attns)
        self.resnets = nn.ModuleList(resnets)
        self.downsamplers = nn.ModuleList([DownsamplePseudo3D(out_c)]) if add_downsample else None


class DownBlockPseudo3D(_Block):
    def __init__(self, in_c, out_c, temb, layers, groups, eps, add_downsample):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlockPseudo3D(in_c if i == 0 else out_c, out_c, temb, groups, eps)
                                      for i in range(layers)])
        self.downsamplers = nn.ModuleList([DownsamplePseudo3D(out_c)]) if add_downsample else None


class UNetMidBlockPseudo3DCrossAttn(_Block):
    has_cross_attention = True

    def __init__(self, c, temb, groups, eps, heads, xdim):
        super().__init__()
        resnets = [ResnetBlockPseudo3D(c, c, temb, groups, eps)]
        attns = [SpatioTemporalTransformerModel(heads, c // heads, c, xdim, groups)]
        resnets.append(ResnetBlockPseudo3D(c, c, temb, groups, eps))
        self.attentions = nn.ModuleList(attns)
        self.resnets = nn.ModuleList(resnets)


class CrossAttnUpBlockPseudo3D(_Block):
    has_cross_attention = True

    def __init__(self, in_c, out_c, prev_c, temb, layers, groups, eps, heads, xdim, add_upsample):
        super().__init__()
        resnets, attns = [], []
        for i in range(layers):
            skip = in_c if i == layers - 1 else out_c
            rin = prev_c if i == 0 else out_c
            resnets.append(ResnetBlockPseudo3D(rin + skip, out_c, temb, groups, eps))
            attns.append(SpatioTemporalTransformerModel(heads, out_c // heads, out_c, xdim, groups))
        self.attentions = nn.ModuleList(attns)
        self.resnets = nn.ModuleList(resnets)
        self.upsamplers = nn.ModuleList([UpsamplePseudo3D(out_c)]) if add_upsample else None


class UpBlockPseudo3D(_Block):
    def __init__(self, in_c, out_c, prev_c, temb, layers, groups, eps, add_upsample):
        super().__init__()
        resnets = []
        for i in range(layers):
            skip = in_c if i == layers - 1 else out_c
            rin = prev_c if i == 0 else out_c
            resnets.append(ResnetBlockPseudo3D(rin + skip, out_c, temb, groups, eps))
        self.resnets = nn.ModuleList(resnets)
        self.upsamplers = nn.ModuleList([UpsamplePseudo3D(out_c)]) if add_upsample else None


class TimestepEmbedding(_ParamOnly):
    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)


class _Config(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


class UNetPseudo3DConditionOutput(dict):
    """attribute + key access like diffusers' BaseOutput (the reference uses both ``.sample`` and ``["sample"]``)."""

    def __init__(self, sample):
        super().__init__(sample=sample)
        self.sample = sample


_DOWN = ("CrossAttnDownBlockPseudo3D", "CrossAttnDownBlockPseudo3D", "CrossAttnDownBlockPseudo3D", "DownBlockPseudo3D")
_UP = ("UpBlockPseudo3D", "CrossAttnUpBlockPseudo3D", "CrossAttnUpBlockPseudo3D", "CrossAttnUpBlockPseudo3D")


_REGISTRATION_EPOCH = [0]


def _bump_registration_epoch(*_args):
    _REGISTRATION_EPOCH[0] += 1


try:        # (torch >= 2.0)
    from torch.nn.modules.module import register_module_parameter_registration_hook, register_module_buffer_registration_hook
    register_module_parameter_registration_hook(_bump_registration_epoch)
    register_module_buffer_registration_hook(_bump_registration_epoch)
except ImportError:       # pragma: no cover
    pass


class UNetPseudo3DConditionModel(nn.Module):
    def __init__(self, sample_size: Optional[int] = None, in_channels: int = 4, out_channels: int = 4,
                 center_input_sample: bool = False, flip_sin_to_cos: bool = True, freq_shift: int = 0,
                 down_block_types: Tuple[str] = _DOWN, mid_block_type: str = "UNetMidBlockPseudo3DCrossAttn",
                 up_block_types: Tuple[str] = _UP, only_cross_attention=False,
                 block_out_channels: Tuple[int] = (320, 640, 1280, 1280), layers_per_block: int = 2,
                 downsample_padding: int = 1, mid_block_scale_factor: float = 1, act_fn: str = "silu",
                 norm_num_groups: int = 32, norm_eps: float = 1e-5, cross_attention_dim: int = 1280,
                 attention_head_dim: Union[int, Tuple[int]] = 8, dual_cross_attention: bool = False,
                 use_linear_projection: bool = False, class_embed_type=None, num_class_embeds=None,
                 upcast_attention: bool = False, resnet_time_scale_shift: str = "default", **kwargs):
        super().__init__()
        cfg = dict(locals())
        for k in ("self", "kwargs", "__class__"):
            cfg.pop(k, None)
        cfg.update(kwargs)
        self._internal_dict = _Config(cfg)
        # ---- what the native graph implements (everything the SD-v1.5 path uses); fail loudly otherwise
        unsupported = []
        if tuple(down_block_types) != _DOWN or tuple(up_block_types) != _UP:
            unsupported.append("block types other than the SD-v1.x layout")
        if class_embed_type is not None or num_class_embeds is not None:
            unsupported.append("class embeddings")
        if center_input_sample or dual_cross_attention or only_cross_attention or resnet_time_scale_shift != "default":
            unsupported.append("center_input_sample / dual_cross_attention / only_cross_attention / scale_shift")
        if act_fn not in ("silu", "swish") or len(block_out_channels) != 4 or downsample_padding != 1 or mid_block_scale_factor != 1:
            unsupported.append("act_fn / depth / padding / scale-factor variants")
        if any(k in kwargs for k in ("lora", "temporal_downsample", "SparseCausalAttention_index")):
            unsupported.append("model_config extras (lora / temporal_downsample / SparseCausalAttention_index)")
        if unsupported:
            raise NotImplementedError("univst_amd native UNet does not implement: " + "; ".join(unsupported))

        boc, groups, eps, xdim = block_out_channels, norm_num_groups, norm_eps, cross_attention_dim
        hl = (attention_head_dim,) * 4 if isinstance(attention_head_dim, int) else tuple(attention_head_dim)     # head COUNT per level
        if len(hl) != 4:
            raise NotImplementedError("attention_head_dim must be an int or a 4-tuple")
        self._heads_per_level = hl
        _BUILD.use_linear_projection = bool(use_linear_projection)
        ted = boc[0] * 4
        self.sample_size = sample_size
        self.conv_in = PseudoConv3d(in_channels, boc[0], kernel_size=3, padding=(1, 1))
        self.time_embedding = TimestepEmbedding(boc[0], ted)
        self.class_embedding = None
        self.down_blocks = nn.ModuleList([])
        self.up_blocks = nn.ModuleList([])
        out_c = boc[0]
        for i, bt in enumerate(down_block_types):
            in_c, out_c = out_c, boc[i]
            final = i == len(boc) - 1
            if bt == "DownBlockPseudo3D":
                blk = DownBlockPseudo3D(in_c, out_c, ted, layers_per_block, groups, eps, not final)
            else:
                blk = CrossAttnDownBlockPseudo3D(in_c, out_c, ted, layers_per_block, groups, eps, hl[i], xdim, not final)
            self.down_blocks.append(blk)
        self.mid_block = UNetMidBlockPseudo3DCrossAttn(boc[-1], ted, groups, eps, hl[-1], xdim)
        self.num_upsamplers = 0
        rev = list(reversed(boc))
        out_c = rev[0]
        for i, bt in enumerate(up_block_types):
            final = i == len(boc) - 1
            prev_c, out_c = out_c, rev[i]
            in_c = rev[min(i + 1, len(boc) - 1)]
            if not final:
                self.num_upsamplers += 1
            if bt == "UpBlockPseudo3D":
                blk = UpBlockPseudo3D(in_c, out_c, prev_c, ted, layers_per_block + 1, groups, eps, not final)
            else:
                blk = CrossAttnUpBlockPseudo3D(in_c, out_c, prev_c, ted, layers_per_block + 1, groups, eps, hl[len(boc) - 1 - i], xdim, not final)
            self.up_blocks.append(blk)
        _BUILD.use_linear_projection = False
        self.conv_norm_out = nn.GroupNorm(num_channels=boc[0], num_groups=groups, eps=eps)
        self.conv_act = nn.SiLU()
        self.conv_out = PseudoConv3d(boc[0], out_channels, kernel_size=3, padding=1)
        self._native_handle = None
        self._native_dirty = True
        self._native_fp = None

    # ------------------------------------------------------------------ nn.Module plumbing
    @property
    def config(self):
        return self._internal_dict

    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device

    def _apply(self, fn, *a, **k):
        self._native_dirty = True
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._native_dirty = True
        return super().load_state_dict(*a, **k)

    def __del__(self):
        h = getattr(self, "_native_handle", None)
        if h is not None:
            try:
                _native.load().univst_unet_destroy(h)
            except Exception:
                pass

    # ------------------------------------------------------------------ native handle
    def _sync_native(self):
        import ctypes as C
        lib = _native.load()
        if self.device.type != "cuda":
            raise RuntimeError("UNetPseudo3DConditionModel.forward runs only on an AMD GPU: call .cuda() first "
                               "(univst_amd has no CPU path)")
        if self._native_handle is not None and os.environ.get("UNIVST_STRICT_WEIGHTS") == "1":
            self.verify_native_weights()
        fp = self._weights_fingerprint()
        if self._native_handle is not None and not self._native_dirty and fp == self._native_fp:
            return
        if self._native_handle is not None:
            lib.univst_unet_destroy(self._native_handle)
            self._native_handle = None
        c = self.config
        cfg = _native.UnetCfg(c.in_channels, c.out_channels, (C.c_int * 4)(*c.block_out_channels), c.layers_per_block,
                              c.cross_attention_dim, (C.c_int * 4)(*self._heads_per_level), c.norm_num_groups, c.norm_eps,
                              int(c.flip_sin_to_cos), float(c.freq_shift))
        h = C.c_void_p()
        _native.check(lib.univst_unet_create(C.byref(cfg), C.byref(h)), "unet_create")
        stream = _native.stream_ptr()
        for name, t in self.state_dict().items():
            if t.dtype == torch.float16:
                dt = 0
            elif t.dtype == torch.float32:
                dt = 1
            else:
                t, dt = t.float(), 1
            t = t.contiguous()
            shape = (C.c_int64 * t.dim())(*t.shape)
            _native.check(lib.univst_unet_load_tensor(h, name.encode(), t.data_ptr(), dt, shape, t.dim(), stream),
                          f"load_tensor({name})")
        torch.cuda.current_stream().synchronize()   # sources may be temporaries
        _native.check(lib.univst_unet_finalize(h, stream), "unet_finalize")
        for k, v in getattr(self, "_native_options", {}).items():
            _native.check(lib.univst_unet_set_option(h, k.encode(), int(v)), f"unet_set_option({k})")
        self._native_handle = h
        self._native_sum = self._content_checksum()
        self._native_dirty = False
        self._native_fp = fp

    def _weights_fingerprint(self):
        """(storage address, autograd version counter) of every parameter / buffer, over a tensor list cached until the next
        ``_apply`` / ``load_state_dict``.  The native copy (incl. the derived fused QKV, tap-inner conv, LayerNorm-folded and stacked
        time_emb_proj tensors) is rebuilt when a tensor was re-allocated or edited in place THROUGH THE TENSOR ITSELF (``p.copy_``,
        ``p.add_`` under ``no_grad``, optimizer steps): those bump ``p._version``.  Edits through ``p.data`` (``p.data.copy_``,
        ``w.data += delta`` — the usual LoRA-merge idiom) carry their own version counter and are NOT seen here: call
        ``invalidate_native()`` after them, or ``verify_native_weights()`` (one device checksum + a sync) to find out; setting
        ``UNIVST_STRICT_WEIGHTS=1`` runs that check before every forward."""
        ts = self.__dict__.get("_native_tensors")
        n = self.__dict__["_native_fp_calls"] = self.__dict__.get("_native_fp_calls", 0) + 1
        # a Parameter / buffer object (re)registered on ANY module of the process (``m.weight = nn.Parameter(...)``, parametrize / LoRA
        # wrappers) bumps _REGISTRATION_EPOCH through torch's global registration hooks: the cached tensor list is then rebuilt on
        # the very next forward, not at the periodic refresh
        if ts is None or self._native_dirty or n % 64 == 0 or self.__dict__.get("_native_epoch") != _REGISTRATION_EPOCH[0]:
            self.__dict__["_native_epoch"] = _REGISTRATION_EPOCH[0]
            ts = self._native_tensors = list(self.state_dict(keep_vars=True).values())
        acc = 0
        for t in ts:
            acc = (acc * 1000003 + t.data_ptr() + 7919 * t._version) & 0xFFFFFFFFFFFFFFFF
        return acc

    def _content_checksum(self):
        """per-tensor fp64 sums of the CURRENT parameter values, one device tensor (no sync here)."""
        return torch.stack([t.detach().sum(dtype=torch.float64) for t in self.state_dict(keep_vars=True).values()])

    def verify_native_weights(self) -> bool:
        """True when the native weight copy still matches the module's parameters by content (catches ``p.data`` edits the
        fingerprint cannot see); on a mismatch the copy is marked stale and rebuilt by the next forward.  Costs one pass over the
        weights and a host sync: a debugging / safety aid, not part of the per-step path."""
        ref = self.__dict__.get("_native_sum")
        if self._native_handle is None or ref is None:
            return False
        ok = bool(torch.equal(self._content_checksum(), ref))
        if not ok:
            self._native_dirty = True
        return ok

    def set_native_option(self, name: str, value: int):
        """tuning switch of the native graph (include/univst.h ``univst_unet_set_option``), e.g. ``("ln_fold", 0)``; survives rebuilds."""
        if not hasattr(self, "_native_options"):
            self._native_options = {}
        self._native_options[name] = int(value)
        if getattr(self, "_native_handle", None) is not None:
            _native.check(_native.load().univst_unet_set_option(self._native_handle, name.encode(), int(value)), f"unet_set_option({name})")

    def invalidate_native(self):
        """force a rebuild of the native weight copy at the next forward."""
        self._native_dirty = True

    def _pnp_state(self):
        """read back what pnp_utils.register_spatial_attention_pnp / register_time poked into the modules."""
        from ..pnp_utils import PNP_LAYERS
        mods = [self.up_blocks[r].attentions[b].transformer_blocks[0].attn1 for r, bs in PNP_LAYERS.items() for b in bs]
        for m in mods:
            if "forward" in m.__dict__ and not getattr(m, "_univst_native_pnp", False):
                raise RuntimeError("attn1.forward was replaced by a foreign closure; the native UNet cannot call it. "
                                   "Use univst_amd's pnp_utils.register_spatial_attention_pnp instead.")
        reg = [getattr(m, "_univst_native_pnp", False) for m in mods]
        if not any(reg):
            return None
        if not all(reg):
            raise NotImplementedError("PnP registered on a subset of the 8 decoder layers")
        vals = {(getattr(m, "idx", None), float(m.eta1), float(m.eta2)) for m in mods}
        if len(vals) != 1:
            raise NotImplementedError(f"per-layer differing PnP state is not supported: {vals}")
        idx, eta1, eta2 = vals.pop()
        if idx is None:
            raise RuntimeError("PnP registered but register_time() was never called (attn1.idx missing)")
        return _native.PnP(1, int(idx), eta1, eta2, 0.65, 3.0)

    # ------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, sample: torch.Tensor, timestep, encoder_hidden_states: torch.Tensor, class_labels=None,
                attention_mask=None, ft_indices: List[int] = None, ft_timesteps: List[int] = None, ft_path: str = None,
                **args):
        import ctypes as C
        if attention_mask is not None or class_labels is not None:
            raise NotImplementedError("attention_mask / class_labels are not used on the UniVST path")
        if sample.dim() != 5:
            raise ValueError(f"sample must be [B,C,F,H,W], got {tuple(sample.shape)}")
        self._sync_native()
        lib = _native.load()
        B, Cin, F_, H, W = sample.shape
        x = sample.to(torch.float16).contiguous()
        txt = encoder_hidden_states.to(torch.float16).contiguous()
        if txt.shape[0] != B:
            raise ValueError(f"encoder_hidden_states batch {txt.shape[0]} != sample batch {B}")
        t = float(timestep.reshape(-1)[0].item()) if torch.is_tensor(timestep) else float(timestep)
        eps = torch.empty(B, self.config.out_channels, F_, H, W, device=x.device, dtype=torch.float16)
        feat, ft_index = None, -1
        if ft_indices is not None and ft_timesteps is not None and ft_path is not None:
            hits = [i for i in ft_indices if i is not None and 0 <= i < 4]
            tt = int(t) if float(int(t)) == t else t
            if hits and any(tt == v for v in ft_timesteps):
                if len(hits) > 1:
                    raise NotImplementedError("one feature-dump block per call")
                ft_index = hits[0]
                boc = list(reversed(self.config.block_out_channels))
                s = min(ft_index + 1, 3)
                hh, ww = (H >> 3) << s, (W >> 3) << s
                feat = torch.empty(F_, hh, ww, boc[ft_index], device=x.device, dtype=torch.float16)
        pnp = self._pnp_state()
        shard = getattr(self, "_frame_shard", None)
        if shard is not None:
            shard.ensure(self, H * W)      # comm hooks on the current handle, workspace sized for this latent
        rc = lib.univst_unet_forward(self._native_handle, x.data_ptr(), t, txt.data_ptr(), B, F_, H, W, txt.shape[1],
                                     C.byref(pnp) if pnp is not None else None, eps.data_ptr(),
                                     feat.data_ptr() if feat is not None else None, ft_index, _native.stream_ptr())
        if rc and shard is not None:
            shard.raise_pending("unet_forward")
        _native.check(rc, "unet_forward")
        if feat is not None:
            tt = int(t) if float(int(t)) == t else t
            save_path = os.path.join(ft_path, f"inversion_feature_map_{ft_index}_block_{tt}_step.pt")
            torch.save(feat, save_path)
            print(f"save feature map at: {save_path}")
            self.last_feature_map = feat
        return UNetPseudo3DConditionOutput(sample=eps.to(sample.dtype) if sample.dtype != torch.float16 else eps)

    # ------------------------------------------------------------------ loading (unet_3d_condition.py:445-509)
    @classmethod
    def from_2d_model(cls, model_path, model_config=None):
        config_path = os.path.join(model_path, "config.json")
        if not os.path.isfile(config_path):
            raise RuntimeError(f"{config_path} does not exist")
        with open(config_path, "r") as f:
            config = json.load(f)
        config.pop("_class_name", None)
        config.pop("_diffusers_version", None)
        rep = {"CrossAttnDownBlock2D": "CrossAttnDownBlockPseudo3D", "DownBlock2D": "DownBlockPseudo3D",
               "UpBlock2D": "UpBlockPseudo3D", "CrossAttnUpBlock2D": "CrossAttnUpBlockPseudo3D"}
        config["mid_block_type"] = "UNetMidBlockPseudo3DCrossAttn"
        config["down_block_types"] = [rep.get(b, b) for b in config["down_block_types"]]
        config["up_block_types"] = [rep.get(b, b) for b in config["up_block_types"]]
        if model_config is not None:
            config.update(model_config)
        known = set(cls.__init__.__code__.co_varnames)
        model = cls(**{k: v for k, v in config.items() if k in known})
        cands = glob.glob(os.path.join(model_path, "*.bin"))
        if cands:
            state_dict = torch.load(cands[0], map_location="cpu", weights_only=True)
            model.load_2d_state_dict(state_dict=state_dict)
        return model

    def load_2d_state_dict(self, state_dict, **kwargs):
        sd3 = self.state_dict()
        for k, v in state_dict.items():
            if k not in sd3:
                raise KeyError(f"2d state_dict key {k} does not exist in 3d model")
            elif v.shape != sd3[k].shape:
                raise ValueError(f"state_dict shape mismatch, 2d {v.shape}, 3d {sd3[k].shape}")
        for k in sd3:
            if "_temporal" in k:
                continue
            if k not in state_dict:
                raise KeyError(f"3d state_dict key {k} does not exist in 2d model")
        sd3.update(state_dict)
        self.load_state_dict(sd3, **kwargs)

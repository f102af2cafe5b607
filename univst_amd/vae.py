"""The temporal VAE behind the pipeline's call sites, on the native library.

``NativeTemporalVAE`` stands where the reference keeps diffusers' ``AutoencoderKLTemporalDecoder`` (src/sd/run_video_style_transfer_sd.py:36-42;
used at pipelines/stable_diffusion.py:369-394, :793-834 and inversion_tools/ddim_inversion.py:28-31,52-55): ``.decode(z, num_frames=F).sample``,
``.encode(x).latent_dist.sample()``, ``.config.scaling_factor``, ``.parameters()`` / ``.dtype``, a ``forward`` whose signature carries ``num_frames``
(the pipeline inspects it).  It takes that class's state dict unchanged (``from_module`` wraps a loaded stock VAE, ``from_state_dict`` a checkpoint
dict) and runs one C-ABI call per decode / encode (univst_vae_*, csrc/vae.hip).  The network is third-party: restated from its published definition,
parity unpinned by the reference (see csrc/vae.hip)."""
import ctypes as C
import types

import torch

from . import _native

DEFAULT_CONFIG = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                      norm_num_groups=32, scaling_factor=0.18215, force_upcast=True)


class _LatentDist:
    """diffusers DiagonalGaussianDistribution over the native encoder's moments (sampling consumes torch's RNG exactly like the stock class)"""

    def __init__(self, moments):
        self.parameters = moments
        self.mean, self.logvar = torch.chunk(moments, 2, dim=1)
        self.logvar = torch.clamp(self.logvar, -30.0, 20.0)
        self.std = torch.exp(0.5 * self.logvar)
        self.var = torch.exp(self.logvar)

    def sample(self, generator=None):
        noise = torch.randn(self.mean.shape, generator=generator, device=self.parameters.device, dtype=self.parameters.dtype)
        return self.mean + self.std * noise

    def mode(self):
        return self.mean


class NativeTemporalVAE(torch.nn.Module):
    def __init__(self, state_dict, config=None, device="cuda"):
        super().__init__()
        cfg = dict(DEFAULT_CONFIG)
        if config is not None:
            cfg.update({k: (config[k] if isinstance(config, dict) else getattr(config, k)) for k in DEFAULT_CONFIG
                        if (k in config if isinstance(config, dict) else hasattr(config, k))})
        self.config = types.SimpleNamespace(**cfg)
        self._dummy = torch.nn.Parameter(torch.zeros(1, device=device, dtype=torch.float16), requires_grad=False)   # .parameters() / .dtype / .device for the call sites
        lib = _native.load()
        c = _native.VaeCfg(cfg["in_channels"], cfg["out_channels"], cfg["latent_channels"], (C.c_int * 4)(*cfg["block_out_channels"]),
                           cfg["layers_per_block"], cfg["norm_num_groups"])
        h = C.c_void_p()
        _native.check(lib.univst_vae_create(C.byref(c), C.byref(h)), "vae_create")
        self._h = h
        st = _native.stream_ptr()
        for k, v in state_dict.items():
            t = v.detach().to(device=device)
            t = t.to(torch.float16 if t.dtype not in (torch.float16, torch.float32) else t.dtype).contiguous()
            shape = (C.c_int64 * t.dim())(*t.shape)
            _native.check(lib.univst_vae_load_tensor(h, k.encode(), _native.ptr(t), 0 if t.dtype == torch.float16 else 1, shape, t.dim(), st),
                          f"vae_load_tensor({k})")
        _native.check(lib.univst_vae_finalize(h, st), "vae_finalize")
        torch.cuda.current_stream().synchronize()

    @classmethod
    def from_module(cls, vae, device="cuda"):
        return cls(vae.state_dict(), config=getattr(vae, "config", None), device=device)

    from_state_dict = classmethod(lambda cls, sd, config=None, device="cuda": cls(sd, config=config, device=device))

    @classmethod
    def from_pretrained(cls, path, subfolder="vae", device="cuda", **_):
        """Load a diffusers-format VAE directory WITHOUT diffusers: ``<path>/<subfolder>/config.json`` + ``diffusion_pytorch_model[.fp16].safetensors``
        (or ``.bin``) — what ``AutoencoderKLTemporalDecoder.from_pretrained(path, subfolder="vae")`` reads (src/sd/run_*_sd.py:36-42).  Local directories
        only (the target boxes have no hub access); raises FileNotFoundError otherwise so that a caller can fall back to diffusers."""
        import json
        import os
        d = os.path.join(path, subfolder) if subfolder else path
        cfg_file = os.path.join(d, "config.json")
        if not os.path.isfile(cfg_file):
            raise FileNotFoundError(f"{cfg_file} not found (NativeTemporalVAE.from_pretrained needs a local diffusers-format directory)")
        with open(cfg_file) as f:
            cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
        cls_name = json.load(open(cfg_file)).get("_class_name", "AutoencoderKLTemporalDecoder")
        if cls_name != "AutoencoderKLTemporalDecoder":
            raise ValueError(f"{cfg_file}: _class_name = {cls_name}; the native VAE restates AutoencoderKLTemporalDecoder only")
        for name in ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors", "diffusion_pytorch_model.bin"):
            w = os.path.join(d, name)
            if os.path.isfile(w):
                if name.endswith(".safetensors"):
                    from safetensors.torch import load_file
                    sd = load_file(w)
                else:
                    sd = torch.load(w, map_location="cpu")
                return cls(sd, config=cfg, device=device)
        raise FileNotFoundError(f"no diffusion_pytorch_model.safetensors / .bin under {d}")

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            try:
                _native.load().univst_vae_destroy(h)
            except Exception:
                pass
            self._h = None

    @property
    def dtype(self):
        return torch.float16

    @property
    def device(self):
        return self._dummy.device

    def _check(self, t, what):
        if not t.is_cuda:
            raise RuntimeError(f"NativeTemporalVAE.{what}: the native VAE runs on the GPU only (no CPU / eager fallback); got a {t.device} tensor")
        return t.to(torch.float16).contiguous()

    @torch.no_grad()
    def decode(self, z, num_frames=1, return_dict=True, **_):
        z = self._check(z, "decode")
        n, c, h, w = z.shape
        out = torch.empty(n, self.config.out_channels, 8 * h, 8 * w, device=z.device, dtype=torch.float16)
        _native.check(_native.load().univst_vae_decode(self._h, _native.ptr(z), n, int(num_frames), h, w, _native.ptr(out), _native.stream_ptr()), "vae_decode")
        return types.SimpleNamespace(sample=out) if return_dict else (out,)

    @torch.no_grad()
    def encode(self, x, return_dict=True):
        x = self._check(x, "encode")
        n, c, H, W = x.shape
        mom = torch.empty(n, 2 * self.config.latent_channels, H // 8, W // 8, device=x.device, dtype=torch.float16)
        _native.check(_native.load().univst_vae_encode(self._h, _native.ptr(x), n, H, W, _native.ptr(mom), _native.stream_ptr()), "vae_encode")
        d = _LatentDist(mom)
        return types.SimpleNamespace(latent_dist=d) if return_dict else (d,)

    def forward(self, sample, sample_posterior=False, return_dict=True, generator=None, num_frames=1):
        d = self.encode(sample).latent_dist
        z = d.sample(generator=generator) if sample_posterior else d.mode()
        return self.decode(z, num_frames=num_frames, return_dict=return_dict)

    def enable_slicing(self):
        pass

    def disable_slicing(self):
        pass

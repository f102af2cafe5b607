"""Oracle: DDIM schedule, inversion loops and the three-branch transfer loop (in-memory, no file I/O).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Reference files restated:
  backbones/video_diffusion_sd/pipelines/stable_diffusion.py:631-791  (video_style_transfer, return_to_timestep)
  inversion_tools/ddim_inversion.py:87-204                            (ddim_loop, ddim_loop_plus, next_step)
  src/util.py:133-144                                                 (load_mask value semantics)
Third-party restated (diffusers 0.35.1 DDIMScheduler, eta=0; parity unpinned by the reference).
"""
from typing import Callable, List, Optional

import numpy as np
import torch
import torch.nn.functional as F

from .unet_ref import latent_adain


class DDIMSchedule:
    """scaled_linear betas in [0.00085, 0.012], 1000 train steps, steps_offset=1, leading spacing,
    set_alpha_to_one=False (SD-v1.5 scheduler_config.json)."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.num_inference_steps = None
        self.timesteps = None

    def set_timesteps(self, n):
        self.num_inference_steps = n
        ratio = self.num_train_timesteps // n
        ts = (np.arange(0, n) * ratio).round()[::-1].copy().astype(np.int64) + self.steps_offset
        self.timesteps = torch.from_numpy(ts)

    def step(self, eps, t, sample):
        """DDIMScheduler.step, eta=0 -> (prev_sample, pred_original_sample)."""
        t = int(t)
        prev_t = t - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        x0 = (sample - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
        prev = a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps
        return prev, x0

    def next_step(self, eps, t, sample):
        """ddim_inversion.py:190-204."""
        t = int(t)
        cur, nxt = min(t - self.num_train_timesteps // self.num_inference_steps, 999), t
        a_t = self.alphas_cumprod[cur] if cur >= 0 else self.final_alpha_cumprod
        a_next = self.alphas_cumprod[nxt]
        x0 = (sample - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
        return a_next ** 0.5 * x0 + (1 - a_next) ** 0.5 * eps

    def return_to_timestep(self, t, sample, x0):
        """stable_diffusion.py:782-791."""
        a_t = self.alphas_cumprod[int(t)]
        return (sample - a_t ** 0.5 * x0) / (1 - a_t) ** 0.5


def mask_from_png_values(arr_u8: np.ndarray) -> np.ndarray:
    """util.py:133-144 value semantics: ``np.array(img) * 255`` wraps in uint8, then clip(0,1)  =>  v != 0."""
    return (((arr_u8.astype(np.uint8) * np.uint8(255)).astype(np.uint8)).clip(0, 1)).astype(np.uint8)


def resize_mask(mask_u8: torch.Tensor, h: int, w: int, dtype=torch.float32) -> torch.Tensor:
    """stable_diffusion.py:688-691: mask [1,F,Hm,Wm] {0,1} -> bilinear (align_corners=False) -> [1,1,F,h,w]."""
    m = F.interpolate(mask_u8.to(dtype), size=(h, w), mode="bilinear", align_corners=False)
    return m[None, :]


def ddim_inversion_loop(eps_fn: Callable, sched: DDIMSchedule, latent: torch.Tensor, num_inv_steps: int,
                        easy_inv: bool) -> List[torch.Tensor]:
    """ddim_inversion.py:87-113 (``ddim_loop``) / :116-167 (``ddim_loop_plus``, Easy-Inv; the 'fix' loop is
    dead because num_fix_itr = 0).  ``eps_fn(latent, t, i)`` is the single-branch UNet call."""
    all_latent = [latent]
    latent = latent.clone()
    last_latent = None
    for i in range(num_inv_steps):
        t = sched.timesteps[len(sched.timesteps) - i - 1]
        eps = eps_fn(latent, t, i)
        if easy_inv and (0.05 + 0.2) * 50 > i > 0.05 * 50 and i > 0:
            latent = 0.5 * latent + 0.5 * last_latent        # AFTER eps was computed from the un-averaged latent
        last_latent = latent
        latent = sched.next_step(eps, t, latent)
        all_latent.append(latent)
    return all_latent


def video_style_transfer_loop(unet_fn: Callable, sched: DDIMSchedule, latents: torch.Tensor,
                              content_inv: List[torch.Tensor], style_inv: List[torch.Tensor],
                              mask_u8: Optional[torch.Tensor], num_inference_steps: int = 50,
                              smoother: Optional[Callable] = None,
                              callback: Optional[Callable] = None) -> torch.Tensor:
    """stable_diffusion.py:680-766.  ``unet_fn(x[3,4,F,h,w], t, i)`` -> eps[3,4,F,h,w];
    ``content_inv[k]`` / ``style_inv[k]`` = ddim_latents_k; ``mask_u8`` = load_mask() output or None.
    ``smoother(i, t, latents, eps)`` -> eps replaces the sliding-window block (:713-759)."""
    sched.set_timesteps(num_inference_steps)
    n = num_inference_steps
    for i, t in enumerate(sched.timesteps):
        c_t = content_inv[n - i].to(latents.dtype)
        s_t = style_inv[n - i].to(latents.dtype)
        if mask_u8 is not None and i <= 0.9 * n:
            m = resize_mask(mask_u8, latents.shape[-2], latents.shape[-1], latents.dtype)
            latents = (1 - m) * latents + m * c_t
        if i > 0.8 * n and i <= 0.9 * n:
            m = resize_mask(mask_u8, latents.shape[-2], latents.shape[-1], latents.dtype) if mask_u8 is not None else 0.0
            latents = (1.0 - m) * latent_adain(latents, s_t) + m * c_t
        x = torch.cat([c_t, s_t, latents])
        eps = unet_fn(x, t, i).chunk(3)[2]
        if smoother is not None and 20 <= i < 25:
            eps = smoother(i, t, latents, eps)
        latents, _ = sched.step(eps, t, latents)
        if callback is not None:
            callback(i, t, latents)
    return latents


def get_images_from_latents(vae_decode: Callable, latents: torch.Tensor) -> np.ndarray:
    """stable_diffusion.py:793-819: 1/0.18215 * z -> VAE decode -> (x/2+0.5).clamp(0,1) -> (x*255).round() uint8,
    [b,3,F,H,W] numpy.  ``vae_decode(z[(b f),4,h,w]) -> [(b f),3,H,W]``."""
    b, c, f, h, w = latents.shape
    z = (1 / 0.18215 * latents).permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    x = (vae_decode(z) / 2 + 0.5).clamp(0, 1)
    x = (x.cpu().float().numpy() * 255).round().astype("uint8")
    return x.reshape(b, f, *x.shape[1:]).transpose(0, 2, 1, 3, 4)


def get_latent_image(vae_encode: Callable, frames_u8: np.ndarray, dtype=torch.float32) -> torch.Tensor:
    """stable_diffusion.py:821-834: frames/127.5 - 1 -> VAE encode (latent_dist.sample()) -> 0.18215 * z, [b,4,F,h,w]."""
    b, c, f, H, W = frames_u8.shape
    x = torch.from_numpy((frames_u8.transpose(0, 2, 1, 3, 4).reshape(b * f, c, H, W) / 127.5) - 1.0).to(dtype)
    z = vae_encode(x)
    return 0.18215 * z.reshape(b, f, *z.shape[1:]).permute(0, 2, 1, 3, 4)


def pixel_smoother(sched: DDIMSchedule, vae_decode: Callable, vae_encode: Callable, flow_fn: Callable,
                   mask01: np.ndarray) -> Callable:
    """the `smoother == 'pixel'` block of stable_diffusion.py:713-759 as the ``smoother(i, t, latents, eps)`` callable of
    video_style_transfer_loop: x0 from the scheduler step (:718), decode to uint8 frames, sliding-window smoothing
    (flow_ref.sliding_window_smooth = :723-751 incl. the masked restore), re-encode, and the noise that returns to
    timestep t (:782-791).  mask01: uint8 {0,1} [1,F,H,W] (load_mask output)."""
    from . import flow_ref

    def fn(i, t, latents, eps):
        _, x0 = sched.step(eps, t, latents)
        frames = get_images_from_latents(vae_decode, x0)
        frames = flow_ref.sliding_window_smooth(frames, flow_fn, mask01)
        x0s = get_latent_image(vae_encode, frames, latents.dtype).to(latents.device)
        return sched.return_to_timestep(t, latents, x0s)
    return fn

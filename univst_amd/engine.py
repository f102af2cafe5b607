"""Denoising-loop engine shared by the pipeline mirror, the inversion tools and bench.py.

Everything here is host orchestration over HIP kernels (univst_amd._native): per step it enqueues the mask
blend, the optional latent AdaIN, one native UNet forward (the whole graph is one C-ABI call) and the DDIM
update; the scheduler's timesteps are read to the host once before the loop, so nothing inside it synchronises
with the host even when a caller keeps them on the GPU (``set_timesteps(n, device=...)`` as the reference pipeline
does).  Inversion latents and masks are device-resident for the whole loop (the reference re-reads them from disk
every step: stable_diffusion.py:683-698).
"""
from typing import Callable, List, Optional, Sequence

import torch

from . import _native
from .backbones.video_diffusion_sd.pnp_utils import latent_adain, register_time


def _alpha(sched, t):
    t = int(t)
    return float(sched.alphas_cumprod[t]) if t >= 0 else float(sched.final_alpha_cumprod)


def ddim_step_coeffs(sched, t):
    """DDIMScheduler.step (eta=0) folded to prev = cx*x + ce*eps (computed in double on the host)."""
    t = int(t)
    prev_t = t - sched.config.num_train_timesteps // sched.num_inference_steps
    a_t, a_p = _alpha(sched, t), _alpha(sched, prev_t)
    cx = (a_p / a_t) ** 0.5
    ce = (1 - a_p) ** 0.5 - (a_p ** 0.5) * ((1 - a_t) ** 0.5) / (a_t ** 0.5)
    return cx, ce


def next_step_coeffs(sched, t):
    """ddim_inversion.py:190-204 folded the same way (runs the DDIM update upward)."""
    t = int(t)
    cur = min(t - sched.config.num_train_timesteps // sched.num_inference_steps, 999)
    a_t, a_n = _alpha(sched, cur), float(sched.alphas_cumprod[t])
    cx = (a_n / a_t) ** 0.5
    ce = (1 - a_n) ** 0.5 - (a_n ** 0.5) * ((1 - a_t) ** 0.5) / (a_t ** 0.5)
    return cx, ce


def ddim_step(sched, eps, t, latents):
    cx, ce = ddim_step_coeffs(sched, t)
    return _native.axpby(latents.contiguous(), eps.contiguous(), cx, ce)


def next_step(eps, t, sample, sched):
    cx, ce = next_step_coeffs(sched, t)
    return _native.axpby(sample.contiguous(), eps.contiguous(), cx, ce)


def pred_original_sample(sched, eps, t, latents):
    a_t = _alpha(sched, t)
    return _native.axpby(latents.contiguous(), eps.contiguous(), 1.0 / a_t ** 0.5, -((1 - a_t) ** 0.5) / a_t ** 0.5)


def return_to_timestep(sched, t, sample, x0):
    """stable_diffusion.py:782-791"""
    a_t = _alpha(sched, t)
    return _native.axpby(sample.contiguous(), x0.contiguous(), 1.0 / (1 - a_t) ** 0.5, -(a_t ** 0.5) / (1 - a_t) ** 0.5)


def _dev16(t, device):
    return t.to(device=device, dtype=torch.float16).contiguous()


def transfer_loop(pipe, latents: torch.Tensor, text3: torch.Tensor, content_inv: Sequence[torch.Tensor],
                  style_inv: Sequence[torch.Tensor], mask_u8: Optional[torch.Tensor], num_inference_steps: int = 50,
                  smoother: Optional[Callable] = None, callback: Optional[Callable] = None,
                  skip_dead_branches: bool = False, shard=None) -> torch.Tensor:
    """stable_diffusion.py:680-766 of the reference, device-resident.

    pipe            object with ``.unet`` (native UNetPseudo3DConditionModel) and ``.scheduler``
    latents         [1,4,F,h,w] fp16 cuda (already latent_adain'ed, run_video_style_transfer_sd.py:57)
    text3           [3,77,D] prompt embeddings (content-inv, style-inv, stylised)
    content_inv     ddim_latents_k for k = 0..n (index k); style_inv likewise
    mask_u8         None or uint8 {0,1} [1,F,H,W] / [F,H,W] (load_mask output): 1 keeps the content latent
    smoother        optional callable(i, t, latents, eps) -> eps (sliding-window block, i in [20,25))
    skip_dead_branches  run only the stylised branch once the PnP window is closed (idx > eta2*50): branches
                    0/1 no longer influence branch 2 and their eps is discarded (:712) — identical output,
                    2/3 less work on those steps.  Off by default (reference-equivalent work).
    shard           optional ``parallel.FrameShard`` already attached to ``pipe.unet`` (one process per GPU):
                    every argument is the FULL clip on every rank; this rank runs its frames of all three branches and the
                    full stylised latents come back on every rank (all-gather at the end).  The smoother is sequential over
                    frames (Gauss-Seidel over key frames): on its five steps the clip is gathered, smoothed by every rank
                    (replicas) and re-sliced.
    """
    sched, unet = pipe.scheduler, pipe.unet
    dev = latents.device
    n = num_inference_steps
    sched.set_timesteps(n)
    sharded = shard is not None and shard.world > 1
    if sharded and latents.shape[2] != shard.frames:
        raise ValueError(f"transfer_loop: the shard was built for {shard.frames} frames, the clip has {latents.shape[2]}")
    cut = shard.slice_frames if sharded else (lambda t: t)
    latents = cut(_dev16(latents, dev))
    text3 = _dev16(text3, dev)
    cinv = [cut(_dev16(t, dev)) for t in content_inv]
    sinv = [cut(_dev16(t, dev)) for t in style_inv]
    adain = shard.latent_adain if sharded else latent_adain
    m = None
    if mask_u8 is not None:
        mk = mask_u8.to(dev).to(torch.uint8).reshape(-1, *mask_u8.shape[-2:]).contiguous()
        m = _native.mask_resize(mk, latents.shape[-2], latents.shape[-1])
        if sharded:
            m = m.reshape(-1, latents.shape[-2], latents.shape[-1])[shard.f0:shard.f0 + shard.local].contiguous()
    eta2 = 0.5
    if skip_dead_branches:
        register_time(pipe, 0)
        st = unet._pnp_state()       # validates that all eight PnP layers agree (raises otherwise) and returns their window
        eta2 = float(st.eta2) if st is not None else -1.0     # no PnP registered: the branches never interact
    timesteps = [int(t) for t in sched.timesteps.tolist()]          # one D2H copy at most, before the loop
    for i, t in enumerate(timesteps):
        c_t, s_t = cinv[n - i], sinv[n - i]
        if m is not None and i <= 0.9 * n:
            latents = _native.mask_blend(latents, c_t, m)
        if i > 0.8 * n and i <= 0.9 * n:
            latents = _native.mask_blend(adain(latents, s_t), c_t, m)
        register_time(pipe, i)
        if skip_dead_branches and i > eta2 * 50:
            eps = _unet_single_branch(unet, latents, t, text3[2:3])
        else:
            x = torch.cat([c_t, s_t, latents])
            eps = unet(x, t, encoder_hidden_states=text3).sample[2:3]
        if smoother is not None and 20 <= i < 25:
            if sharded:
                eps = cut(smoother(i, t, shard.gather_frames(latents), shard.gather_frames(eps.contiguous())))
            else:
                eps = smoother(i, t, latents, eps)
        latents = ddim_step(sched, eps, t, latents)
        if callback is not None:
            callback(i, sched.timesteps[i], latents)
    return shard.gather_frames(latents) if sharded else latents


def _unet_single_branch(unet, latents, t, text1):
    """B=1 call of a PnP-registered UNet outside the shift window: the [-1,'first'] K/V sources stay, the
    shift is inactive, so the stylised branch is independent of the other two."""
    return unet(latents, t, encoder_hidden_states=text1).sample


def inversion_loop(pipe, sched, latent: torch.Tensor, text1: torch.Tensor, num_inv_steps: int, easy_inv,
                   ft_indices=None, ft_timesteps=None, ft_path=None, on_latent: Optional[Callable] = None) -> List[torch.Tensor]:
    """ddim_inversion.py:87-167 (``ddim_loop`` / ``ddim_loop_plus``), device-resident.  ``on_latent(k, z)`` is
    called for k = 0..num_inv_steps (the reference's torch.save points).

    ``latent`` may carry several independent trajectories along the batch axis ([B,4,F,h,w], B <= 8) with ``easy_inv`` a
    per-trajectory list: the content (Easy-Inv) and the style (plain DDIM) inversions of one job then share every UNet call
    — the same arithmetic per trajectory (GroupNorm / attention never mix batch elements), twice the rows per launch, which
    is what the single-branch shapes lack.  The feature dump is taken from trajectory 0, as in the reference
    (unet_3d_condition.py:430-436 dumps sample[0])."""
    dev = latent.device
    latent = _dev16(latent, dev)
    B = latent.shape[0]
    easy = list(easy_inv) if isinstance(easy_inv, (list, tuple)) else [bool(easy_inv)] * B
    if len(easy) != B:
        raise ValueError(f"easy_inv has {len(easy)} entries for a batch of {B} trajectories")
    text1 = _dev16(text1, dev)
    if text1.shape[0] == 1 and B > 1:
        text1 = text1.expand(B, -1, -1).contiguous()
    all_latent = [latent]
    if on_latent:
        on_latent(0, latent)
    last_latent = None
    timesteps = [int(t) for t in sched.timesteps.tolist()]
    for i in range(num_inv_steps):
        t = timesteps[len(timesteps) - i - 1]
        eps = pipe.unet(latent, t, encoder_hidden_states=text1, ft_indices=ft_indices, ft_timesteps=ft_timesteps,
                        ft_path=ft_path)["sample"]
        if any(easy) and (0.05 + 0.2) * 50 > i > 0.05 * 50 and i > 0:
            if all(easy):
                latent = _native.axpby(latent, last_latent, 0.5, 0.5)      # Easy-Inv averaging, AFTER eps
            else:
                latent = torch.cat([_native.axpby(latent[b:b + 1].contiguous(), last_latent[b:b + 1].contiguous(), 0.5, 0.5) if easy[b]
                                    else latent[b:b + 1] for b in range(B)])
        last_latent = latent
        latent = next_step(eps, t, latent, sched)
        if on_latent:
            on_latent(i + 1, latent)
        all_latent.append(latent)
    return all_latent

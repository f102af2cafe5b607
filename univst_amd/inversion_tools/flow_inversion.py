"""Mirror of inversion_tools/flow_inversion.py of the reference: the rectified-flow inversions of the SD3 / SD3.5 path
(rf_inversion :123-188, rf_solver :191-264) with their latent updates on the native kernels (csrc/sd3.hip ``univst_axpbypcz``;
coefficients in double on the host, as engine.ddim_step does for DDIM).  Same function names / arguments / files written.
``pipeline`` is duck-typed exactly as the reference uses it: ``encode_prompt``, ``scheduler.set_timesteps`` / ``.sigmas``,
``transformer(hidden_states=, timestep=, encoder_hidden_states=, pooled_projections=, idx=, ft_*=, return_dict=False)``,
``progress_bar``, ``device``.  The transformer itself (diffusers' MM-DiT) is third-party: no SD3 backbone is part of this build,
so the pipeline-level entry points below raise."""
import os

import torch

from .. import _native


def content_inversion_reconstruction(*a, **k):
    raise NotImplementedError("SD3 / SD3.5 backbone (diffusers SD3Transformer2DModel + SD3 pipeline) is not part of this build; "
                              "rf_inversion / rf_solver run with any object that duck-types the pipeline")


style_inversion_reconstruction = content_inversion_reconstruction


def _prep(pipeline, prompt, num_inference_steps):
    pe, _, pp, _ = pipeline.encode_prompt(prompt=prompt, prompt_2=prompt, prompt_3=prompt)
    pipeline.scheduler.set_timesteps(num_inference_steps, device=pipeline.device)
    ts = torch.flip(pipeline.scheduler.sigmas, dims=[0])
    return pe, pp, [float(t) for t in ts.double().tolist()]


def _velocity(pipeline, z, t1000, pe, pp, idx, ft_indices, ft_timesteps, ft_path):
    t_vec = torch.full((z.shape[0],), t1000, dtype=z.dtype, device=z.device)
    return pipeline.transformer(hidden_states=z, timestep=t_vec, encoder_hidden_states=pe, pooled_projections=pp, idx=idx, ft_indices=ft_indices,
                                ft_timesteps=ft_timesteps, ft_path=ft_path, return_dict=False)[0].to(torch.float16).contiguous()


def _save(inversion_path, k, z):
    if inversion_path is not None:
        torch.save(z.detach().clone(), os.path.join(inversion_path, f"ddim_latents_{k}.pt"))


@torch.no_grad()
def rf_inversion(pipeline, image_latents, prompt="", gamma=0.5, num_inference_steps=50, inversion_path=None, ft_indices=None, ft_timesteps=None,
                 ft_path=None):
    """flow_inversion.py:123-188: controlled forward ODE towards a seeded Gaussian target.
    x <- x + dt * (gamma * (noise - x) / (1 - t) + (1 - gamma) * v)  ==  (1 - dt*gamma/(1-t)) x + (dt*gamma/(1-t)) noise + dt*(1-gamma) v"""
    pe, pp, ts = _prep(pipeline, prompt, num_inference_steps)
    dt_in = image_latents.dtype
    _save(inversion_path, 0, image_latents)
    target_noise = torch.randn_like(image_latents)          # same RNG call, same place as the reference (:150)
    z = image_latents.to(device="cuda", dtype=torch.float16).contiguous()
    noise = target_noise.to(device="cuda", dtype=torch.float16).contiguous()
    with pipeline.progress_bar(total=len(ts) - 1) as bar:
        for idx, (t_curr, t_prev) in enumerate(zip(ts[:-1], ts[1:])):
            v = _velocity(pipeline, z, t_curr * 1000, pe, pp, idx, ft_indices, ft_timesteps, ft_path)
            dt = t_prev - t_curr
            a = dt * gamma / (1.0 - t_curr)
            z = _native.axpbypcz(z, noise, v, 1.0 - a, a, dt * (1.0 - gamma))
            _save(inversion_path, idx + 1, z.to(dt_in))
            bar.update()
    return z.to(dt_in)


@torch.no_grad()
def rf_solver(pipeline, image_latents, prompt="", num_inference_steps=50, inversion_path=None, ft_indices=None, ft_timesteps=None, ft_path=None):
    """flow_inversion.py:191-264: second-order inversion.  mid = x + dt/2 v;  x <- x + dt v + dt^2/2 * (v_mid - v) / (dt/2)
    == x + (dt - c) v + c v_mid with c = (dt^2/2) / (dt/2), evaluated in the reference's operation order on the host."""
    pe, pp, ts = _prep(pipeline, prompt, num_inference_steps)
    dt_in = image_latents.dtype
    _save(inversion_path, 0, image_latents)
    z = image_latents.to(device="cuda", dtype=torch.float16).contiguous()
    with pipeline.progress_bar(total=len(ts) - 1) as bar:
        for idx, (t_curr, t_prev) in enumerate(zip(ts[:-1], ts[1:])):
            v = _velocity(pipeline, z, 1000 * t_curr, pe, pp, idx, ft_indices, ft_timesteps, ft_path)
            dt = t_prev - t_curr
            mid = _native.axpby(z, v, 1.0, dt / 2)
            v_mid = _velocity(pipeline, mid, 1000 * (t_curr + dt / 2), pe, pp, idx, ft_indices, ft_timesteps, ft_path)
            c = (0.5 * dt ** 2) / (dt / 2)
            z = _native.axpbypcz(z, v, v_mid, 1.0, dt - c, c)
            _save(inversion_path, idx + 1, z.to(dt_in))
            bar.update()
    return z.to(dt_in)

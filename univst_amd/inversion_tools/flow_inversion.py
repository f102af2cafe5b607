"""Mirror of inversion_tools/flow_inversion.py of the reference: the rectified-flow inversions of the SD3 / SD3.5 path
(rf_inversion :123-188, rf_solver :191-264) with their latent updates on the native kernels (csrc/sd3.hip ``univst_axpbypcz``;
coefficients in double on the host, as engine.ddim_step does for DDIM).  Same function names / arguments / files written.
``pipeline`` is duck-typed exactly as the reference uses it: ``encode_prompt``, ``scheduler.set_timesteps`` / ``.sigmas``,
``transformer(hidden_states=, timestep=, encoder_hidden_states=, pooled_projections=, idx=, ft_*=, return_dict=False)``,
``progress_bar``, ``device``.  With the native MM-DiT (backbones/video_diffusion_sd3/models/transformer_3D_model.py) and the pipeline
mirror (pipelines/custom_pipeline.py) the pipeline-level entry points run too; the VAE and the text encoders stay stock."""
import os

import torch

from .. import _native


def _img_latents(pipe, pixel_values):
    z = pipe.vae.encode(pixel_values).latent_dist.sample()          # stock VAE (third-party), consumes torch RNG like the reference
    return (z - pipe.vae.config.shift_factor) * pipe.vae.config.scaling_factor


def _invert_and_reconstruct(pipe, img_latents, inversion_path, reconstruction_path, name, time_steps, weight_dtype, is_rf_solver, ft, reconstruct):
    print("inversion:")
    if is_rf_solver:
        inv = rf_solver(pipe, img_latents, prompt="", num_inference_steps=time_steps, inversion_path=inversion_path, **ft)
    else:
        inv = rf_inversion(pipe, img_latents, prompt="", DTYPE=weight_dtype, gamma=0.0, num_inference_steps=time_steps,
                           inversion_path=inversion_path, **ft)
    if reconstruct:
        print("reconstruction:")
        images = pipe.reconstruction(prompt="", img_latents=img_latents, inversed_latents=inv, eta_base=0.85, eta_trend="constant", start_step=25,
                                     end_step=39, guidance_scale=1.0, DTYPE=weight_dtype, num_inference_steps=time_steps)
        from ..src.util import save_images_as_mp4
        save_images_as_mp4(images, os.path.join(reconstruction_path, name), fps=8)      # diffusers.utils.export_to_video(fps=8) in the reference
    return inv


def content_inversion_reconstruction(pipe, content_path, inversion_path, reconstruction_path, num_frames, height, width, time_steps, weight_dtype,
                                     ft_indices, ft_timesteps, ft_path, is_rf_solver=False, reconstruct=True):
    """flow_inversion.py:16-69: frames -> stock VAE -> rf_solver / rf_inversion(gamma 0) -> preview reconstruction."""
    from .ddim_inversion import read_content_pixels
    pixel_values = read_content_pixels(content_path, num_frames, height, width).to(weight_dtype).cuda()
    return _invert_and_reconstruct(pipe, _img_latents(pipe, pixel_values), inversion_path, reconstruction_path, "content_video.mp4", time_steps,
                                   weight_dtype, is_rf_solver, dict(ft_indices=ft_indices, ft_timesteps=ft_timesteps, ft_path=ft_path), reconstruct)


def style_inversion_reconstruction(pipe, style_path, inversion_path, reconstruction_path, num_frames, height, width, time_steps, weight_dtype,
                                   is_rf_solver=False, reconstruct=True):
    """flow_inversion.py:72-118: the style image repeated num_frames times."""
    import numpy as np
    from PIL import Image
    img = Image.open(style_path).convert("RGB").resize((width, height))
    px = 2.0 * torch.from_numpy(np.array(img)).permute(2, 0, 1).float() / 255.0 - 1.0          # transforms.ToTensor() then 2x - 1
    pixel_values = px.repeat(num_frames, 1, 1, 1).to(weight_dtype).cuda()
    return _invert_and_reconstruct(pipe, _img_latents(pipe, pixel_values), inversion_path, reconstruction_path, "style_video.mp4", time_steps,
                                   weight_dtype, is_rf_solver, {}, reconstruct)


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
                 ft_path=None, DTYPE=None):
    """flow_inversion.py:123-188: controlled forward ODE towards a seeded Gaussian target.
    x <- x + dt * (gamma * (noise - x) / (1 - t) + (1 - gamma) * v)  ==  (1 - dt*gamma/(1-t)) x + (dt*gamma/(1-t)) noise + dt*(1-gamma) v
    ``DTYPE`` is accepted and ignored: the reference's content_/style_inversion_reconstruction pass it (flow_inversion.py:47,98) to a
    function that does not take it — the non-rf-solver branch raises TypeError at HEAD."""
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
            # the midpoint evaluation takes NO ft_* arguments (flow_inversion.py:242-249): the dumped feature map is that of the first
            # evaluation (hidden state at t_curr), not overwritten by the midpoint's
            v_mid = _velocity(pipeline, mid, 1000 * (t_curr + dt / 2), pe, pp, idx, None, None, None)
            c = (0.5 * dt ** 2) / (dt / 2)
            z = _native.axpbypcz(z, v, v_mid, 1.0, dt - c, c)
            _save(inversion_path, idx + 1, z.to(dt_in))
            bar.update()
    return z.to(dt_in)

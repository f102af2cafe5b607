"""Mirror of inversion_tools/ddim_inversion.py of the reference (same function names / arguments / files
written).  The UNet calls and the DDIM updates run in HIP kernels (univst_amd.engine)."""
import os

import torch

from .. import engine
from ..src.util import load_video_frames, save_videos_grid


def _encode_frames(pipe, pixel_values, num_frames):
    latents = pipe.vae.encode(pixel_values).latent_dist.sample()
    latents = latents.view(-1, num_frames, *latents.shape[1:]).permute(0, 2, 1, 3, 4)
    return latents * pipe.vae.config.scaling_factor


def read_content_pixels(content_path, num_frames, height, width):
    """[F, 3, H, W] float in [-1, 1] on the CPU.  ddim_inversion.py:21-27: an .mp4 goes through decord.VideoReader(width, height) and
    keeps the first num_frames frames; anything else is a frame folder (src/util.py load_video_frames).  decord is imported here,
    on first use, not at module import as the reference does (ddim_inversion.py:12-13): PNG-folder users do not need it."""
    if content_path.endswith(".mp4"):
        try:
            import decord
        except ImportError as e:
            raise ImportError(".mp4 content input is read with decord (as in the reference); install it or extract the frames to "
                              "a folder of %05d.png files") from e
        decord.bridge.set_bridge("torch")
        vr = decord.VideoReader(content_path, width=width, height=height)
        video = vr.get_batch(list(range(0, len(vr), 1))[:num_frames])          # [F, H, W, 3] uint8
        return (torch.as_tensor(video).float() / 127.5 - 1.0).permute(0, 3, 1, 2).contiguous()
    return load_video_frames(content_path, num_frames, image_size=(width, height))


def content_inversion_reconstruction(pipe, ddim_inv_scheduler, content_path, inversion_path, reconstruction_path, num_frames,
                                     height, width, time_steps, weight_dtype, ft_indices=None, ft_timesteps=None, ft_path=None,
                                     is_opt=True, reconstruct=True):
    """ddim_inversion.py:16-42: a folder of %05d.png frames or an .mp4 (read with decord, the reference's reader)."""
    pixel_values = read_content_pixels(content_path, num_frames, height, width).to(weight_dtype).cuda()
    latents = _encode_frames(pipe, pixel_values, num_frames)
    print("inversion:")
    z = ddim_inversion(pipe, ddim_inv_scheduler, video_latent=latents, num_inv_steps=time_steps, prompt="",
                       inversion_path=inversion_path, ft_indices=ft_indices, ft_timesteps=ft_timesteps, ft_path=ft_path,
                       is_opt=is_opt)[-1].to(weight_dtype)
    if reconstruct:
        print("reconstruction:")
        sample = pipe.reconstruction("", height=height, width=width, latents=z, video_length=num_frames, guidance_scale=1.0).images
        save_videos_grid(sample.permute(0, 4, 1, 2, 3).contiguous(), os.path.join(reconstruction_path, "content_video.mp4"), fps=8)
    return z


def style_inversion_reconstruction(pipe, ddim_inv_scheduler, style_path, inversion_path, reconstruction_path, num_frames, height,
                                   width, time_steps, weight_dtype, ft_indices=None, ft_timesteps=None, ft_path=None,
                                   is_opt=True, reconstruct=True):
    """ddim_inversion.py:45-65: the style image repeated num_frames times."""
    import numpy as np
    from PIL import Image
    img = Image.open(style_path).convert("RGB").resize((width, height))      # ddim_inversion.py:48 (no EXIF handling, RGB first)
    px = torch.from_numpy((np.array(img) / 127.5) - 1.0).permute(2, 0, 1).float()
    pixel_values = px.unsqueeze(0).repeat(num_frames, 1, 1, 1).to(weight_dtype).cuda()
    latents = _encode_frames(pipe, pixel_values, num_frames)
    print("inversion:")
    z = ddim_inversion(pipe, ddim_inv_scheduler, video_latent=latents, num_inv_steps=time_steps, prompt="",
                       inversion_path=inversion_path, ft_indices=ft_indices, ft_timesteps=ft_timesteps, ft_path=ft_path,
                       is_opt=is_opt)[-1].to(weight_dtype)
    if reconstruct:
        print("reconstruction:")
        sample = pipe.reconstruction("", height=height, width=width, latents=z, video_length=num_frames, guidance_scale=1.0).images
        save_videos_grid(sample.permute(0, 4, 1, 2, 3).contiguous(), os.path.join(reconstruction_path, "style_video.mp4"), fps=8)
    return z


@torch.no_grad()
def ddim_inversion(pipeline, ddim_scheduler, video_latent, num_inv_steps, prompt="", inversion_path=None, ft_indices=None,
                   ft_timesteps=None, ft_path=None, is_opt=False):
    """ddim_inversion.py:71-84"""
    loop = ddim_loop_plus if is_opt else ddim_loop
    return loop(pipeline, ddim_scheduler, video_latent, num_inv_steps, prompt, inversion_path, ft_indices=ft_indices,
                ft_timesteps=ft_timesteps, ft_path=ft_path)


def _run(pipeline, sched, latent, n, prompt, inversion_path, easy, ft_indices, ft_timesteps, ft_path):
    cond = init_prompt(pipeline, prompt).chunk(2)[1]

    def save(k, z):
        if inversion_path is not None:
            torch.save(z.detach().clone(), os.path.join(inversion_path, f"ddim_latents_{k}.pt"))
    return engine.inversion_loop(pipeline, sched, latent.cuda(), cond, n, easy, ft_indices, ft_timesteps, ft_path, save)


@torch.no_grad()
def ddim_loop(pipeline, ddim_scheduler, latent, num_inv_steps, prompt, inversion_path, ft_indices=None, ft_timesteps=None,
              ft_path=None):
    """ddim_inversion.py:87-113"""
    return _run(pipeline, ddim_scheduler, latent, num_inv_steps, prompt, inversion_path, False, ft_indices, ft_timesteps, ft_path)


@torch.no_grad()
def ddim_loop_plus(pipeline, ddim_scheduler, latent, num_inv_steps, prompt, inversion_path, ft_indices=None, ft_timesteps=None,
                   ft_path=None):
    """ddim_inversion.py:116-167 (Easy-Inv; the 'fix' iterations are dead code upstream: num_fix_itr = 0)"""
    return _run(pipeline, ddim_scheduler, latent, num_inv_steps, prompt, inversion_path, True, ft_indices, ft_timesteps, ft_path)


@torch.no_grad()
def init_prompt(pipeline, prompt):
    """ddim_inversion.py:170-187"""
    tok = pipeline.tokenizer
    un = tok([""], padding="max_length", max_length=tok.model_max_length, return_tensors="pt")
    ue = pipeline.text_encoder(un.input_ids.to(pipeline.device))[0]
    ti = tok([prompt], padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt")
    te = pipeline.text_encoder(ti.input_ids.to(pipeline.device))[0]
    return torch.cat([ue, te])


def next_step(model_output, timestep, sample, ddim_scheduler):
    """ddim_inversion.py:190-204"""
    return engine.next_step(model_output, timestep, sample, ddim_scheduler)


def get_noise_pred_single(pipeline, latents, t, context, ft_indices=None, ft_timesteps=None, ft_path=None):
    """ddim_inversion.py:207-212"""
    return pipeline.unet(latents, t, encoder_hidden_states=context, ft_indices=ft_indices, ft_timesteps=ft_timesteps,
                         ft_path=ft_path)["sample"]


@torch.no_grad()
def paired_ddim_inversion(pipeline, ddim_scheduler, content_latent, style_latent, num_inv_steps, prompt="", content_inversion_path=None,
                          style_inversion_path=None, ft_indices=None, ft_timesteps=None, ft_path=None, content_is_opt=True, style_is_opt=False):
    """Not in the reference (it runs the two inversions as two scripts, scripts/start_sd.sh): the content and the style inversion
    of one job as ONE batch-2 trajectory, so every UNet call works on twice the rows (the single-branch shapes leave the 256-CU
    chip half empty at the 32x32 / 16x16 levels).  Per trajectory the arithmetic is that of ddim_inversion(); the files written are
    the same ``ddim_latents_k.pt`` in the two folders and the same feature dump (content trajectory).  Returns (content, style)
    lists of 51 latents."""
    cond = init_prompt(pipeline, prompt).chunk(2)[1]
    z = torch.cat([content_latent, style_latent]).cuda()

    def save(k, zz):
        for path, b in ((content_inversion_path, 0), (style_inversion_path, 1)):
            if path is not None:
                torch.save(zz[b:b + 1].detach().clone(), os.path.join(path, f"ddim_latents_{k}.pt"))
    traj = engine.inversion_loop(pipeline, ddim_scheduler, z, cond, num_inv_steps, [content_is_opt, style_is_opt], ft_indices, ft_timesteps,
                                 ft_path, save)
    return [t[0:1] for t in traj], [t[1:2] for t in traj]

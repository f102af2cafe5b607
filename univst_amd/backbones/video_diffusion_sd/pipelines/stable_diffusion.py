"""Host-side mirror of ``SpatioTemporalStableDiffusionPipeline``
(backbones/video_diffusion_sd/pipelines/stable_diffusion.py:45-834 of the reference).

Same constructor, same ``reconstruction`` / ``video_style_transfer`` signatures, result in ``.images``.
The denoising loops run through univst_amd.engine (HIP kernels; device-resident latents/masks); the VAE and
the CLIP text encoder stay stock PyTorch-ROCm modules supplied by the caller (third-party weights).
Extras (all optional, reference defaults unchanged): ``smoother='pixel'`` + ``flow_fn`` switch on the
sliding-window smoothing that is dead code in the reference (:715), ``smoother='latent'`` + ``latent_flows`` the
latent-space variant the reference's README describes but does not implement, ``content_inv_latents`` /
``style_inv_latents`` / ``masks`` accept in-memory tensors instead of paths, ``output_type='latent'`` skips
the VAE, ``skip_dead_branches`` (see engine.transfer_loop).

Multi-GPU (BASELINE config 4): when the process is one of N started by ``torchrun --nproc-per-node N`` (a
``torch.distributed`` process group exists), ``video_style_transfer`` shards the clip's frames over the ranks by itself —
``shard=None`` (default) builds ``parallel.FrameShard`` for the group, checks one sharded forward against the unsharded one
and falls back IPC -> RCCL callbacks (``parallel.FrameShard.self_check``); ``shard=False`` keeps every rank on the whole clip;
an explicit ``FrameShard`` is used as is.  Every rank gets the full latents back; only rank 0 decodes (``.images`` is None elsewhere).
"""
import inspect
from typing import Callable, List, Optional, Union

import numpy as np
import torch

from .... import engine
from ....src.util import load_ddim_latents_at_t, load_mask
from ..pnp_utils import latent_adain, register_time  # noqa: F401  (re-exported like the reference module)


class StableDiffusionPipelineOutput:
    def __init__(self, images, nsfw_content_detected=None):
        self.images = images
        self.nsfw_content_detected = nsfw_content_detected


class SpatioTemporalStableDiffusionPipeline:
    _optional_components = []

    def __init__(self, vae, text_encoder, tokenizer, unet, scheduler):
        if hasattr(scheduler.config, "steps_offset") and scheduler.config.steps_offset != 1:
            raise ValueError("scheduler.config.steps_offset must be 1 (SD-v1.5 DDIM configuration)")
        if hasattr(scheduler.config, "clip_sample") and scheduler.config.clip_sample is True:
            raise ValueError("scheduler.config.clip_sample must be False")
        self.vae, self.text_encoder, self.tokenizer, self.unet, self.scheduler = vae, text_encoder, tokenizer, unet, scheduler
        boc = getattr(getattr(vae, "config", None), "block_out_channels", (1, 1, 1, 1))
        self.vae_scale_factor = 2 ** (len(boc) - 1)

    @property
    def device(self):
        return self.unet.device

    @property
    def _execution_device(self):
        return self.device

    # ------------------------------------------------------------------ plumbing kept from the reference (:142-176, 396-476)
    def enable_vae_slicing(self):
        self.vae.enable_slicing()

    def disable_vae_slicing(self):
        self.vae.disable_slicing()

    def enable_sequential_cpu_offload(self, gpu_id=0):
        """the reference offloads unet / text_encoder / vae through accelerate; the native UNet keeps its weights in a device
        arena and cannot be paged, so only the two third-party modules are offloaded."""
        try:
            from accelerate import cpu_offload
        except ImportError:
            raise ImportError("Please install accelerate via `pip install accelerate`")
        device = torch.device(f"cuda:{gpu_id}")
        for m in (self.text_encoder, self.vae):
            if m is not None and not hasattr(m, "_h"):       # (a NativeTemporalVAE keeps its weights in the library, like the UNet)
                cpu_offload(m, device)

    def prepare_extra_step_kwargs(self, generator, eta):
        keys = set(inspect.signature(self.scheduler.step).parameters.keys())
        kw = {}
        if "eta" in keys:
            kw["eta"] = eta
        if "generator" in keys:
            kw["generator"] = generator
        return kw

    def check_inputs(self, prompt, height, width, callback_steps):
        """stable_diffusion.py:413-428: same conditions, same messages."""
        if not isinstance(prompt, str) and not isinstance(prompt, list):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if (callback_steps is None) or (callback_steps is not None and (not isinstance(callback_steps, int) or callback_steps <= 0)):
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps} of type {type(callback_steps)}.")

    def prepare_latents(self, batch_size, num_channels_latents, clip_length, height, width, dtype, device, generator, latents=None):
        """stable_diffusion.py:430-476."""
        shape = (batch_size, num_channels_latents, clip_length, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch"
                             f" size of {batch_size}. Make sure the batch size matches the length of the generators.")
        if latents is None:
            if isinstance(generator, list):
                one = (1,) + shape[1:]
                latents = torch.cat([torch.randn(one, generator=generator[i], device=device, dtype=dtype) for i in range(batch_size)], dim=0)
            else:
                latents = torch.randn(shape, generator=generator, device=device, dtype=dtype)
        else:
            if tuple(latents.shape) != shape:
                raise ValueError(f"Unexpected latents shape, got {latents.shape}, expected {shape}")
            latents = latents.to(device)
        return latents * getattr(self.scheduler, "init_noise_sigma", 1.0)

    # ------------------------------------------------------------------ prompt (stable_diffusion.py:193-308)
    def _encode_prompt(self, prompt, device, num_images_per_prompt, do_classifier_free_guidance, negative_prompt):
        batch_size = len(prompt) if isinstance(prompt, list) else 1
        ti = self.tokenizer(prompt, padding="max_length", max_length=self.tokenizer.model_max_length, truncation=True,
                            return_tensors="pt")
        am = None
        if getattr(getattr(self.text_encoder, "config", None), "use_attention_mask", False):
            am = ti.attention_mask.to(device)
        emb = self.text_encoder(ti.input_ids.to(device), attention_mask=am)[0]
        bs, seq, _ = emb.shape
        emb = emb.repeat(1, num_images_per_prompt, 1).view(bs * num_images_per_prompt, seq, -1)
        if do_classifier_free_guidance:
            if negative_prompt is None:
                un = [""] * batch_size
            elif isinstance(negative_prompt, str):
                un = [negative_prompt]
            else:
                un = negative_prompt
            ui = self.tokenizer(un, padding="max_length", max_length=ti.input_ids.shape[-1], truncation=True, return_tensors="pt")
            ue = self.text_encoder(ui.input_ids.to(device), attention_mask=am)[0]
            ue = ue.repeat(1, num_images_per_prompt, 1).view(batch_size * num_images_per_prompt, ue.shape[1], -1)
            emb = torch.cat([ue, emb])
        return emb

    # ------------------------------------------------------------------ VAE (stable_diffusion.py:369-394, 793-834)
    def _vae_decode(self, latents, num_frames, chunk=16):
        latents = latents.permute(0, 2, 1, 3, 4).flatten(0, 1)
        latents = 1 / self.vae.config.scaling_factor * latents
        fwd = self.vae.forward
        accepts = "num_frames" in set(inspect.signature(fwd).parameters.keys())
        frames = []
        for i in range(0, latents.shape[0], chunk):
            z = latents[i:i + chunk].to(next(self.vae.parameters()).dtype)
            kw = {"num_frames": z.shape[0]} if accepts else {}
            frames.append(self.vae.decode(z, **kw).sample)
        return (torch.cat(frames, dim=0) / 2 + 0.5).clamp(0, 1)

    def decode_latents(self, latents, num_frames=None, decode_chunk_size=16):
        F_ = latents.shape[2]
        frames = self._vae_decode(latents, F_, decode_chunk_size)
        n, c, h, w = frames.shape
        frames = frames.view(n // F_, F_, c, h, w).permute(0, 1, 3, 4, 2)
        return frames.cpu().float().numpy()

    def get_images_from_latents(self, latents, decode_chunk_size=16):
        """uint8 frames [b,3,F,H,W] on the device (the reference returns numpy, :793-819)."""
        F_ = latents.shape[2]
        frames = self._vae_decode(latents * self.vae.config.scaling_factor / 0.18215, F_, decode_chunk_size)
        frames = (frames.float() * 255).round().to(torch.uint8)
        n, c, h, w = frames.shape
        return frames.view(n // F_, F_, c, h, w).permute(0, 2, 1, 3, 4).contiguous()

    def get_latent_image(self, frames_u8):
        """uint8 [b,3,F,H,W] -> latents [b,4,F,h,w] (:821-834; consumes torch RNG like the reference)."""
        b, c, F_, H, W = frames_u8.shape
        img = frames_u8.permute(0, 2, 1, 3, 4).reshape(b * F_, c, H, W).float() / 127.5 - 1.0
        img = img.to(device=self.device, dtype=next(self.vae.parameters()).dtype)
        z = self.vae.encode(img).latent_dist.sample()
        z = z.view(b, F_, *z.shape[1:]).permute(0, 2, 1, 3, 4)
        return (0.18215 * z).to(torch.float16).contiguous()

    def return_to_timestep(self, timestep, sample, sample_stablized, schedules):
        return engine.return_to_timestep(schedules, timestep, sample, sample_stablized)

    # ------------------------------------------------------------------ reconstruction (:479-628)
    @torch.no_grad()
    def reconstruction(self, prompt, height=512, width=512, num_inference_steps=50, video_length=8, guidance_scale=7.5,
                       negative_prompt=None, num_images_per_prompt=1, eta=0.0, generator=None, latents=None,
                       output_type="tensor", return_dict=True, callback=None, callback_steps=1, **kwargs):
        if eta != 0.0:
            raise NotImplementedError("eta != 0")
        self.check_inputs(prompt, height, width, callback_steps)
        batch_size = 1 if isinstance(prompt, str) else len(prompt)
        device = self._execution_device
        cfg = guidance_scale > 1.0
        text = self._encode_prompt(prompt, device, num_images_per_prompt, cfg, negative_prompt)
        self.scheduler.set_timesteps(num_inference_steps)
        latents = self.prepare_latents(batch_size * num_images_per_prompt, self.unet.config.in_channels, video_length, height, width,
                                       text.dtype, device, generator, latents)
        latents = latents.to(device=device, dtype=torch.float16).contiguous()
        for i, t in enumerate(self.scheduler.timesteps):
            x = torch.cat([latents] * 2) if cfg else latents
            eps = self.unet(x, t, encoder_hidden_states=text).sample
            if cfg:
                eu, et = eps.chunk(2)
                eps = (eu.float() + guidance_scale * (et.float() - eu.float())).to(torch.float16)
            latents = engine.ddim_step(self.scheduler, eps, t, latents)
            if callback is not None and i % callback_steps == 0:
                callback(i, t, latents)
        return self._finish(latents, output_type, return_dict)

    def _finish(self, latents, output_type, return_dict):
        if output_type == "latent":
            image = latents
        else:
            image = self.decode_latents(latents)
            if output_type == "tensor":
                image = torch.from_numpy(image)
        if not return_dict:
            return (image, None)
        return StableDiffusionPipelineOutput(images=image, nsfw_content_detected=None)

    # ------------------------------------------------------------------ video_style_transfer (:631-780)
    @torch.no_grad()
    def video_style_transfer(self, prompt, num_inference_steps=50, negative_prompt=None, num_videos_per_prompt=1, eta=0.0,
                             generator=None, latents=None, output_type="tensor", return_dict=True, callback=None,
                             callback_steps=1, content_inv_path=None, style_inv_path=None, mask_path=None,
                             content_inv_latents=None, style_inv_latents=None, masks=None, smoother=None, flow_fn=None,
                             latent_flows=None, skip_dead_branches=False, shard=None, **kwargs):
        if eta != 0.0:
            raise NotImplementedError("eta != 0")
        device = self._execution_device
        n = num_inference_steps
        pe = self._encode_prompt(prompt, device, num_videos_per_prompt, False, negative_prompt)
        ie = self._encode_prompt("", device, num_videos_per_prompt, False, None)
        text3 = torch.cat([ie, ie, pe])
        F_ = latents.shape[2]
        if content_inv_latents is None:
            content_inv_latents = [load_ddim_latents_at_t(k, content_inv_path) for k in range(n + 1)]
        if style_inv_latents is None:
            style_inv_latents = [load_ddim_latents_at_t(k, style_inv_path) for k in range(n + 1)]
        if masks is None and mask_path:
            masks = load_mask(mask_path, n_frames=F_)
        sm = None
        if smoother == "latent":
            # SURVEY §8f-2 (README.md:59 of the reference; no reference code): warp + window blend on the x0 latents themselves
            # with the content clip's flows at latent resolution — no VAE decode / encode and no RAFT call inside the loop
            if latent_flows is None or masks is None:
                raise ValueError("smoother='latent' needs latent_flows (src.cal_optica_flow.make_latent_flows on the content frames) and masks")
            from ....src.cal_optica_flow import latent_sliding_window_smooth
            from .... import _native
            lf = latent_flows.to(device=device, dtype=torch.float32).contiguous()
            mm = _native.mask_resize(masks.to(device).to(torch.uint8).reshape(-1, *masks.shape[-2:]).contiguous(), latents.shape[-2], latents.shape[-1])

            def sm(i, t, lat, eps):
                x0 = engine.pred_original_sample(self.scheduler, eps, t, lat)
                return engine.return_to_timestep(self.scheduler, t, lat, latent_sliding_window_smooth(x0, lf, mm))
        elif smoother is not None:
            if smoother != "pixel":
                print("error")
                return
            if flow_fn is None or masks is None:
                raise ValueError("smoother='pixel' needs flow_fn (RAFT stand-in) and masks (the reference raises NameError "
                                 "without mask_path, stable_diffusion.py:751)")
            from ....src.cal_optica_flow import sliding_window_smooth
            m01 = masks.to(device).to(torch.uint8).reshape(-1, *masks.shape[-2:])

            def sm(i, t, lat, eps):
                x0 = engine.pred_original_sample(self.scheduler, eps, t, lat)
                frames = self.get_images_from_latents(x0)
                frames = sliding_window_smooth(frames, flow_fn, m01)
                return engine.return_to_timestep(self.scheduler, t, lat, self.get_latent_image(frames))
        cb = (lambda i, t, l: callback(i, t, l) if i % callback_steps == 0 else None) if callback is not None else None
        if shard is None:            # one process per GPU under torchrun: shard the frames (checked once per pipeline and geometry)
            from ....parallel import auto_frame_shard, dist_rank_world
            if dist_rank_world()[1] > 1:
                shard = auto_frame_shard(self, F_, latents.shape[-2:], check_inputs=(content_inv_latents[n].to(device), style_inv_latents[n].to(device), text3))
        if shard is False or (shard is not None and shard.world == 1):
            shard = None
        if shard is not None and skip_dead_branches:
            raise NotImplementedError("skip_dead_branches with a frame shard (the single-branch call has its own K/V exchange schedule)")
        latents = engine.transfer_loop(self, latents.to(device), text3, content_inv_latents, style_inv_latents, masks, n,
                                       smoother=sm, callback=cb, skip_dead_branches=skip_dead_branches, shard=shard)
        if shard is not None and shard.rank != 0 and output_type != "latent":      # rank 0 decodes and writes; the others are done
            return StableDiffusionPipelineOutput(images=None) if return_dict else (None, None)
        return self._finish(latents, output_type, return_dict)

"""Mirror of backbones/video_diffusion_sd3/pipelines/custom_pipeline.py of the reference: ``CustomStableDiffusion3Pipeline`` with
``generate_eta_values`` (:18-43), ``reconstruction`` (:45-124) and ``video_style_transfer`` (:126-346), same argument names.

The reference subclasses diffusers' ``StableDiffusion3Pipeline`` (absent from both boxes); this mirror is a plain class holding the
same components.  The text encoders and the VAE stay stock third-party modules — ``encode_prompt`` / ``vae.decode`` are call sites,
as for the SD-v1.5 pipeline — while the transformer is the native MM-DiT (models/transformer_3D_model.py) and every latent update
runs in a HIP kernel: mask blend (``univst_mask_blend``), the plugin's ``latent_adain``, and the eta-interpolated velocity + Euler
step folded into ONE three-term launch:

    v'  = v + eta * (-(target - x) / t - v)                       custom_pipeline.py:324-328
    x'  = x + (sigma_next - sigma) * v'                           FlowMatchEulerDiscreteScheduler.step
        = (1 + d*eta/t) x  +  d*(1 - eta) v  +  (-d*eta/t) target         with d = sigma_next - sigma, coefficients in double on the host

Reference defects on this path kept visible: ``video_style_transfer`` reads an undefined name
``ddim_inv_latents_at_t`` in the latent-AdaIN window (:303) — the reference raises NameError at step 40.  The documented fixed reading
implemented here is ``content_inv_latents_at_t`` (the tensor the SD-v1.5 loop blends at the same place, stable_diffusion.py:699-702);
golden G19 pins the no-mask loop, where that term is multiplied by 0.0, to the reference's own code."""
import torch

from .... import _native
from ....src.util import load_ddim_latents_at_t, load_mask
from ..pnp_utils import latent_adain


class StableDiffusion3PipelineOutput:
    def __init__(self, images):
        self.images = images


class _Bar:
    def __init__(self, total):
        self.total = total

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def update(self, n=1):
        pass


def _f16(t):
    return t.to(device="cuda", dtype=torch.float16).contiguous()


def _blend_frames(a, b, m):
    """(1 - m) a + m b for a, b [F, C, h, w], m [F, h, w] (the reference's resized_mask.permute(1, 0, 2, 3) broadcast over C)."""
    out = torch.empty_like(a)
    for f in range(a.shape[0]):
        _native.mask_blend(a[f][None, :, None], b[f][None, :, None], m[f:f + 1], out=out[f][None, :, None])
    return out


class CustomStableDiffusion3Pipeline:
    def __init__(self, transformer, scheduler, vae=None, text_encoder=None, tokenizer=None, text_encoder_2=None, tokenizer_2=None,
                 text_encoder_3=None, tokenizer_3=None, image_encoder=None, feature_extractor=None):
        self.transformer, self.scheduler, self.vae = transformer, scheduler, vae
        self.text_encoder, self.text_encoder_2, self.text_encoder_3 = text_encoder, text_encoder_2, text_encoder_3
        self.tokenizer, self.tokenizer_2, self.tokenizer_3 = tokenizer, tokenizer_2, tokenizer_3
        self.vae_scale_factor = 8
        self.default_sample_size = getattr(getattr(transformer, "config", None), "sample_size", 128)
        self._interrupt = False

    @property
    def device(self):
        return self.transformer.device

    _execution_device = device

    @property
    def interrupt(self):
        return self._interrupt

    def progress_bar(self, iterable=None, total=None):
        try:
            from tqdm import tqdm
            return tqdm(iterable, total=total) if iterable is not None else tqdm(total=total)
        except ImportError:
            return _Bar(total)

    def maybe_free_model_hooks(self):
        pass

    # ------------------------------------------------------------------ prompt: stock CLIP x2 + T5 (call site only)
    def encode_prompt(self, prompt=None, prompt_2=None, prompt_3=None, device=None, num_images_per_prompt=1, do_classifier_free_guidance=False,
                      negative_prompt=None, negative_prompt_2=None, negative_prompt_3=None, prompt_embeds=None, negative_prompt_embeds=None,
                      pooled_prompt_embeds=None, negative_pooled_prompt_embeds=None, max_sequence_length=256, **unused):
        """StableDiffusion3Pipeline.encode_prompt restated for the unguided call the UniVST loops make (do_classifier_free_guidance
        False: the negative outputs are None): CLIP-L and CLIP-G penultimate hidden states concatenated on the feature axis, zero-padded
        to T5's width and concatenated with the T5 states on the token axis; pooled = the two projected CLIP embeddings."""
        if prompt_embeds is not None:
            return prompt_embeds, negative_prompt_embeds, pooled_prompt_embeds, negative_pooled_prompt_embeds
        if do_classifier_free_guidance:
            raise NotImplementedError("classifier-free guidance is not used by the UniVST SD3 loops (guidance_scale 1.0)")
        if self.text_encoder is None or self.text_encoder_2 is None:
            raise RuntimeError("encode_prompt needs the stock CLIP text encoders (third-party); pass prompt_embeds / pooled_prompt_embeds instead")
        device = device or self.device
        prompts = [prompt] if isinstance(prompt, str) else list(prompt)
        clip, pooled = [], []
        for tok, enc_ in ((self.tokenizer, self.text_encoder), (self.tokenizer_2, self.text_encoder_2)):
            ids = tok(prompts, padding="max_length", max_length=tok.model_max_length, truncation=True, return_tensors="pt").input_ids.to(device)
            out = enc_(ids, output_hidden_states=True)
            pooled.append(out[0])
            clip.append(out.hidden_states[-2])
        clip = torch.cat(clip, dim=-1)
        if self.text_encoder_3 is None:
            t5 = torch.zeros(len(prompts), max_sequence_length, self.transformer.config.joint_attention_dim, device=device, dtype=clip.dtype)
        else:
            ids = self.tokenizer_3(prompts, padding="max_length", max_length=max_sequence_length, truncation=True, add_special_tokens=True,
                                   return_tensors="pt").input_ids.to(device)
            t5 = self.text_encoder_3(ids)[0]
        clip = torch.nn.functional.pad(clip, (0, t5.shape[-1] - clip.shape[-1]))
        pe = torch.cat([clip, t5.to(clip.dtype)], dim=-2).repeat_interleave(num_images_per_prompt, dim=0)
        pp = torch.cat(pooled, dim=-1).repeat_interleave(num_images_per_prompt, dim=0)
        return pe, None, pp, None

    # ------------------------------------------------------------------ custom_pipeline.py:18-43
    def generate_eta_values(self, timesteps, start_step, end_step, eta, eta_trend):
        assert start_step < end_step and start_step >= 0 and end_step <= len(timesteps), "Invalid start_step and end_step"
        eta_values = [0.0] * len(timesteps)
        if eta_trend == "constant":
            for i in range(start_step, end_step):
                eta_values[i] = eta
        elif eta_trend == "linear_increase":
            total_time = timesteps[start_step] - timesteps[end_step - 1]
            for i in range(start_step, end_step):
                eta_values[i] = eta * (timesteps[start_step] - timesteps[i]) / total_time
        elif eta_trend == "linear_decrease":
            total_time = timesteps[start_step] - timesteps[end_step - 1]
            for i in range(start_step, end_step):
                eta_values[i] = eta * (timesteps[i] - timesteps[end_step - 1]) / total_time
        else:
            raise NotImplementedError(f"Unsupported eta_trend: {eta_trend}")
        return eta_values

    def _schedule(self, num_inference_steps, sigmas=None):
        """retrieve_timesteps + the per-step Euler increments, on the host in double: (timesteps tensor, t list, d_sigma list)"""
        if sigmas is not None:
            self.scheduler.set_timesteps(sigmas=sigmas, device=self.device)
        else:
            self.scheduler.set_timesteps(num_inference_steps, device=self.device)
        ts = self.scheduler.timesteps
        sig = [float(s) for s in self.scheduler.sigmas.double().tolist()]
        return ts, [float(t) for t in ts.double().tolist()], [sig[i + 1] - sig[i] for i in range(len(sig) - 1)]

    def _euler_eta(self, latents, v, target, d, eta, t_curr):
        k = d * float(eta) / t_curr
        return _native.axpbypcz(latents, v, target, 1.0 + k, d * (1.0 - float(eta)), -k)

    def _decode(self, latents, output_type):
        if output_type == "latent":
            return latents
        if self.vae is None:
            raise RuntimeError("decoding needs the stock VAE (third-party); pass output_type='latent'")
        z = (latents / self.vae.config.scaling_factor) + self.vae.config.shift_factor
        img = self.vae.decode(z.to(next(self.vae.parameters()).dtype), return_dict=False)[0]
        if output_type == "pt":
            return img
        img = (img / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).float().cpu().numpy()
        if output_type == "np":
            return img
        from PIL import Image
        return [Image.fromarray((f * 255).round().astype("uint8")) for f in img]

    # ------------------------------------------------------------------ custom_pipeline.py:45-124
    @torch.no_grad()
    def reconstruction(self, img_latents, inversed_latents, eta_base, eta_trend, start_step, end_step, guidance_scale=1.0, prompt="",
                       DTYPE=torch.float16, num_inference_steps=50, output_type="pil"):
        if guidance_scale > 1.0:
            raise NotImplementedError("classifier-free guidance is not used by the UniVST SD3 path (guidance_scale 1.0)")
        ts, tl, ds = self._schedule(num_inference_steps)
        pe, _, pp, _ = self.encode_prompt(prompt=prompt, prompt_2=prompt, prompt_3=prompt)
        B = inversed_latents.shape[0]
        pe, pp = _f16(pe).expand(B, -1, -1).contiguous(), _f16(pp).expand(B, -1).contiguous()
        latents, target = _f16(inversed_latents), _f16(img_latents)
        eta_values = self.generate_eta_values(tl, start_step, end_step, eta_base, eta_trend)
        T = self.scheduler.config.num_train_timesteps
        with self.progress_bar(total=num_inference_steps) as bar:
            for i, t in enumerate(ts):
                v = self.transformer(hidden_states=latents, timestep=t.expand(B), encoder_hidden_states=pe, pooled_projections=pp,
                                     return_dict=False)[0]
                latents = self._euler_eta(latents, _f16(v), target, ds[i], eta_values[i], tl[i] / T)
                bar.update()
        return self._decode(latents.to(DTYPE), output_type)

    # ------------------------------------------------------------------ custom_pipeline.py:126-346
    @torch.no_grad()
    def video_style_transfer(self, prompt=None, prompt_2=None, prompt_3=None, height=None, width=None, num_inference_steps=50, sigmas=None,
                             negative_prompt=None, negative_prompt_2=None, negative_prompt_3=None, num_images_per_prompt=1, generator=None,
                             latents=None, prompt_embeds=None, negative_prompt_embeds=None, pooled_prompt_embeds=None,
                             negative_pooled_prompt_embeds=None, output_type="pil", return_dict=True, callback_on_step_end=None,
                             callback_on_step_end_tensor_inputs=("latents",), max_sequence_length=256, mu=None,
                             content_inv_path=None, style_inv_path=None, mask_path=None, eta_base=0.95, eta_trend="constant", start_step=10,
                             end_step=20, img_latents=None, content_inv_latents=None, style_inv_latents=None, shard=None, masks=None):
        """``content_inv_latents`` / ``style_inv_latents`` (addition): the per-step inversion latents in memory (lists indexed by the
        step label k of ``ddim_latents_{k}.pt``) instead of ``*_inv_path`` — the in-process hand-off of SURVEY §8f-1; ``masks`` the
        ``load_mask`` tensor in memory instead of ``mask_path``.

        ``shard`` (addition, BASELINE config 5 on N GPUs): under ``torchrun --nproc-per-node N`` (a process group exists) the clip's
        frames are split over the ranks — ``None`` builds ``parallel.Sd3FrameShard`` for the group (K | V of the first and the previous
        frame travel through the library's IPC communicator inside every joint attention), ``False`` keeps the whole clip on every rank,
        an explicit ``Sd3FrameShard`` is used as is.  All arguments are the full clip on every rank; every rank gets the full latents
        back and rank 0 decodes (``.images`` is None elsewhere unless ``output_type='latent'``)."""
        if latents is None or img_latents is None:
            raise ValueError("video_style_transfer needs `latents` (the shifted content noise) and `img_latents`")
        if callback_on_step_end is not None:
            raise NotImplementedError("step callbacks are not on the UniVST path")
        self._interrupt = False
        pe, _, pp, _ = self.encode_prompt(prompt=prompt, prompt_2=prompt_2, prompt_3=prompt_3, do_classifier_free_guidance=False,
                                          prompt_embeds=prompt_embeds, pooled_prompt_embeds=pooled_prompt_embeds,
                                          num_images_per_prompt=num_images_per_prompt, max_sequence_length=max_sequence_length)
        F_all = latents.shape[0]
        if shard is None:
            from ....parallel import Sd3FrameShard, dist_rank_world
            rank, world = dist_rank_world()
            if world > 1:
                key = (rank, world, F_all, tuple(latents.shape[-2:]))
                cache = self.__dict__.setdefault("_univst_shards", {})
                if key not in cache:
                    ps = self.transformer.config.patch_size
                    cache[key] = Sd3FrameShard(rank, world, F_all).attach(self.transformer, tokens=(latents.shape[-2] // ps) * (latents.shape[-1] // ps))
                shard = cache[key]
        if shard is False or (shard is not None and shard.world == 1):
            shard = None
        cut = shard.slice_frames if shard is not None else (lambda t: t)
        latents, target = cut(_f16(latents)), cut(_f16(img_latents))
        F_ = latents.shape[0]
        pe_all = _f16(pe).repeat(3 * F_, 1, 1)                       # custom_pipeline.py:226-227
        pp_all = _f16(pp).repeat(3 * F_, 1)
        ts, tl, ds = self._schedule(num_inference_steps, sigmas)
        n = len(tl)
        eta_values = self.generate_eta_values(tl, start_step, end_step, eta_base, eta_trend)
        T = self.scheduler.config.num_train_timesteps

        def inv(store, path, k):
            return cut(_f16(store[k] if store is not None else load_ddim_latents_at_t(k, path)))

        mask = None
        if mask_path or masks is not None:                           # load_mask + bilinear resize once (the reference redoes both per step)
            m = masks if masks is not None else load_mask(mask_path, n_frames=F_all)
            mask = _native.mask_resize(m[0].to("cuda").contiguous(), latents.shape[-2], latents.shape[-1])
            if shard is not None:
                mask = mask.reshape(F_all, *latents.shape[-2:])[shard.f0:shard.f0 + shard.local].contiguous()
        with self.progress_bar(total=n) as bar:
            for i, t in enumerate(ts):
                if self.interrupt:
                    continue
                c_t, s_t = inv(content_inv_latents, content_inv_path, 50 - i), inv(style_inv_latents, style_inv_path, 50 - i)
                if mask is not None and i <= 0.9 * n:                # localized latent blending (:289-294)
                    latents = _blend_frames(latents, c_t, mask)
                if i >= 0.8 * n and i <= 0.9 * n:                    # :296-303, fixed reading of the undefined name: the content latents
                    shifted = _f16(latent_adain(latents, s_t))
                    latents = _blend_frames(shifted, c_t, mask) if mask is not None else shifted
                x = torch.cat([c_t, s_t, latents])
                v = self.transformer(hidden_states=x, timestep=t.expand(x.shape[0]), encoder_hidden_states=pe_all, pooled_projections=pp_all,
                                     return_dict=False, joint_attention_kwargs={"idx": i})[0]
                latents = self._euler_eta(latents, _f16(v[2 * F_:]), target, ds[i], eta_values[i], tl[i] / T)
                bar.update()
        if shard is not None:
            latents = shard.gather_frames(latents)
            if shard.rank != 0 and output_type != "latent":          # rank 0 decodes and writes
                self.maybe_free_model_hooks()
                return StableDiffusion3PipelineOutput(images=None) if return_dict else (None,)
        image = self._decode(latents, output_type)
        self.maybe_free_model_hooks()
        if not return_dict:
            return (image,)
        return StableDiffusion3PipelineOutput(images=image)

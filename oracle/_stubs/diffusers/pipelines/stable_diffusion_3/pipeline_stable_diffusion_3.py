"""Stand-in for the parts of diffusers 0.35.1 StableDiffusion3Pipeline that the reference's CustomStableDiffusion3Pipeline
(backbones/video_diffusion_sd3/pipelines/custom_pipeline.py) calls from `reconstruction` / `video_style_transfer`: component
registration, `check_inputs`, `encode_prompt` (embeddings must be handed in or come from `fixed_prompt`: no text encoders here),
`prepare_latents` for given latents, `retrieve_timesteps`.  Generator-side only (see ../../../README.md)."""
from ..pipeline_utils import DiffusionPipeline


def retrieve_timesteps(scheduler, num_inference_steps=None, device=None, timesteps=None, sigmas=None, **kwargs):
    if sigmas is not None:
        scheduler.set_timesteps(sigmas=sigmas, device=device, **kwargs)
        return scheduler.timesteps, len(scheduler.timesteps)
    scheduler.set_timesteps(num_inference_steps, device=device, **kwargs)
    return scheduler.timesteps, num_inference_steps


class StableDiffusion3Pipeline(DiffusionPipeline):
    def __init__(self, transformer, scheduler, vae=None, text_encoder=None, tokenizer=None, text_encoder_2=None, tokenizer_2=None,
                 text_encoder_3=None, tokenizer_3=None, image_encoder=None, feature_extractor=None):
        super().__init__()
        self.register_modules(transformer=transformer, scheduler=scheduler, vae=vae, text_encoder=text_encoder, tokenizer=tokenizer,
                              text_encoder_2=text_encoder_2, tokenizer_2=tokenizer_2, text_encoder_3=text_encoder_3, tokenizer_3=tokenizer_3)
        self.vae_scale_factor = 8
        self.default_sample_size = 128
        self._interrupt = False
        self.fixed_prompt = None            # (prompt_embeds, pooled_prompt_embeds) returned by encode_prompt when none are passed

    @property
    def _execution_device(self):
        return self.device

    @property
    def interrupt(self):
        return self._interrupt

    def check_inputs(self, *args, **kwargs):
        pass

    def encode_prompt(self, prompt=None, prompt_2=None, prompt_3=None, prompt_embeds=None, pooled_prompt_embeds=None, **kwargs):
        if prompt_embeds is None:
            prompt_embeds, pooled_prompt_embeds = self.fixed_prompt
        return prompt_embeds, None, pooled_prompt_embeds, None

    def prepare_latents(self, batch_size, num_channels_latents, height, width, dtype, device, generator, latents=None):
        if latents is None:
            raise NotImplementedError("stub: latents must be given")
        return latents.to(device=device, dtype=dtype)

    def maybe_free_model_hooks(self):
        pass

from univst_amd.backbones.video_diffusion_sd3.pipelines.custom_pipeline import CustomStableDiffusion3Pipeline, StableDiffusion3PipelineOutput  # noqa: F401

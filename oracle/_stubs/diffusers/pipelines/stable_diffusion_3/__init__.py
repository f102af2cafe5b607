from .pipeline_stable_diffusion_3 import StableDiffusion3Pipeline, retrieve_timesteps  # noqa: F401
from .pipeline_output import StableDiffusion3PipelineOutput  # noqa: F401

"""Minimal restatement of the diffusers==0.35.1 symbols used by the UniVST SD-v1.5 path (see ../README.md)."""
import torch
from torch import nn

__version__ = "0.35.1-stub"


class ModelMixin(nn.Module):
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device


from .schedulers import DDIMScheduler, DDPMScheduler, FlowMatchEulerDiscreteScheduler  # noqa: E402
from .models import AutoencoderKL, AutoencoderKLTemporalDecoder  # noqa: E402
from .pipelines.stable_diffusion_3 import StableDiffusion3Pipeline  # noqa: E402

from univst_amd.backbones.video_diffusion_sd.pnp_utils import *  # noqa: F401,F403
from univst_amd.backbones.video_diffusion_sd.pnp_utils import register_time, register_spatial_attention_pnp, attention_adain, latent_adain  # noqa: F401

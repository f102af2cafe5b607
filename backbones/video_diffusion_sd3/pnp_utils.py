from univst_amd.backbones.video_diffusion_sd3.pnp_utils import *  # noqa: F401,F403
from univst_amd.backbones.video_diffusion_sd3.pnp_utils import CrossFrameProcessor, AttentionShiftProcessor, register_spatial_attention_pnp, attention_adain, latent_adain  # noqa: F401

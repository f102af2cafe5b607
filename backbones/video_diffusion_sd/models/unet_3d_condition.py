from univst_amd.backbones.video_diffusion_sd.models.unet_3d_condition import UNetPseudo3DConditionModel, UNetPseudo3DConditionOutput  # noqa: F401

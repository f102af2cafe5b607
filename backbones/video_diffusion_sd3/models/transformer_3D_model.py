from univst_amd.backbones.video_diffusion_sd3.models.transformer_3D_model import CustomSD3Transformer2DModel, Transformer2DModelOutput  # noqa: F401

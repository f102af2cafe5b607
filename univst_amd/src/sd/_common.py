"""Shared pieces of the three SD-v1.5 command-line entry points (mirrors of src/sd/run_*_sd.py of the reference).

CLIP (transformers) is a third-party model and stays a stock PyTorch-ROCm module; the SVD temporal VAE is LOADED through diffusers
(its checkpoint format and config) and then runs on the native library (univst_amd.vae.NativeTemporalVAE takes the stock module's state dict;
round 5, SURVEY §8 f2 — ``UNIVST_VAE=stock`` keeps the diffusers module).  Both must be available locally — there is no hub access on the target
boxes.  The UNet, the DDIM loops, the PnP injection, mask blending and AdaIN run in the native HIP library."""
import os

import torch


def build_pipeline(pretrained_model_path, weight_dtype=torch.float16, vae_path="stabilityai/stable-video-diffusion-img2vid"):
    from transformers import CLIPTextModel, CLIPTokenizer
    try:
        from diffusers import AutoencoderKLTemporalDecoder, DDIMScheduler
    except ImportError as e:  # pragma: no cover
        raise RuntimeError("the CLI needs `diffusers` for the temporal VAE (third-party model, not re-implemented); "
                           "the native UNet / pipeline classes themselves do not") from e
    from ...backbones.video_diffusion_sd.models.unet_3d_condition import UNetPseudo3DConditionModel
    from ...backbones.video_diffusion_sd.pipelines.stable_diffusion import SpatioTemporalStableDiffusionPipeline
    tokenizer = CLIPTokenizer.from_pretrained(pretrained_model_path, subfolder="tokenizer")
    text_encoder = CLIPTextModel.from_pretrained(pretrained_model_path, subfolder="text_encoder").requires_grad_(False)
    vae = AutoencoderKLTemporalDecoder.from_pretrained(vae_path, subfolder="vae").requires_grad_(False)
    unet = UNetPseudo3DConditionModel.from_2d_model(os.path.join(pretrained_model_path, "unet")).requires_grad_(False)
    vae = vae.to(weight_dtype).cuda()
    if os.environ.get("UNIVST_VAE", "native") != "stock":
        from ...vae import NativeTemporalVAE
        vae = NativeTemporalVAE.from_module(vae)
    pipe = SpatioTemporalStableDiffusionPipeline(
        vae=vae, text_encoder=text_encoder.to(weight_dtype).cuda(), tokenizer=tokenizer,
        unet=unet.to(weight_dtype).cuda(), scheduler=DDIMScheduler.from_pretrained(pretrained_model_path, subfolder="scheduler"))
    return pipe, DDIMScheduler


def add_common_args(parser):
    parser.add_argument("--pretrained_model_path", type=str, default="stable-diffusion-v1-5/stable-diffusion-v1-5")
    parser.add_argument("--weight_dtype", type=torch.dtype, default=torch.float16)
    parser.add_argument("--time_steps", type=int, default=50)
    parser.add_argument("--seed", type=int, default=33)
    return parser

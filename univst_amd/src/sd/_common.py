"""Shared pieces of the three SD-v1.5 command-line entry points (mirrors of src/sd/run_*_sd.py of the reference).

CLIP (transformers) is a third-party model and stays a stock PyTorch-ROCm module; the SVD temporal VAE is LOADED through diffusers
(its checkpoint format and config) and then runs on the native library (univst_amd.vae.NativeTemporalVAE takes the stock module's state dict;
round 5, SURVEY §8 f2 — ``UNIVST_VAE=stock`` keeps the diffusers module).  Both must be available locally — there is no hub access on the target
boxes.  The UNet, the DDIM loops, the PnP injection, mask blending and AdaIN run in the native HIP library."""
import os

import torch


def build_pipeline(pretrained_model_path, weight_dtype=torch.float16, vae_path="stabilityai/stable-video-diffusion-img2vid"):
    """The objects run_*_sd.py build (src/sd/run_video_style_transfer_sd.py:30-47).  diffusers is needed only where a local directory cannot be read
    without it: the VAE loads natively from ``<vae_path>/vae`` (config.json + safetensors / bin) when that is a directory, and the DDIM scheduler
    is the native one (same ``from_pretrained(path, subfolder="scheduler")``) when diffusers is absent."""
    from transformers import CLIPTextModel, CLIPTokenizer
    from ...backbones.video_diffusion_sd.models.unet_3d_condition import UNetPseudo3DConditionModel
    from ...backbones.video_diffusion_sd.pipelines.stable_diffusion import SpatioTemporalStableDiffusionPipeline
    from ...vae import NativeTemporalVAE
    try:
        from diffusers import DDIMScheduler
    except ImportError:
        from ...schedulers import DDIMScheduler
    tokenizer = CLIPTokenizer.from_pretrained(pretrained_model_path, subfolder="tokenizer")
    text_encoder = CLIPTextModel.from_pretrained(pretrained_model_path, subfolder="text_encoder").requires_grad_(False)
    vae = None
    stock = os.environ.get("UNIVST_VAE", "native") == "stock"
    if not stock and weight_dtype != torch.float16:
        # the native VAE computes and returns fp16 only (univst_amd/vae.py): a pipeline asked for another dtype keeps the stock module, as the reference does
        print(f"[univst_amd] weight_dtype {weight_dtype}: the native temporal VAE is fp16-only, using the stock AutoencoderKLTemporalDecoder", flush=True)
        stock = True
    if not stock and os.path.isdir(os.path.join(vae_path, "vae")):
        vae = NativeTemporalVAE.from_pretrained(vae_path, subfolder="vae")
    if vae is None:
        try:
            from diffusers import AutoencoderKLTemporalDecoder
        except ImportError as e:  # pragma: no cover
            raise RuntimeError(f"the temporal VAE could not be loaded: {vae_path}/vae is not a local diffusers-format directory and `diffusers` (which would resolve "
                               "it from the hub cache) is not installed") from e
        vae = AutoencoderKLTemporalDecoder.from_pretrained(vae_path, subfolder="vae").requires_grad_(False).to(weight_dtype).cuda()
        if not stock:
            vae = NativeTemporalVAE.from_module(vae)
    unet = UNetPseudo3DConditionModel.from_2d_model(os.path.join(pretrained_model_path, "unet")).requires_grad_(False)
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

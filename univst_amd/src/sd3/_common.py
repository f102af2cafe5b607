"""Shared pieces of the three SD3 / SD3.5 command-line entry points (mirrors of src/sd3/run_*_sd3.py of the reference).

The CLIP / T5 text encoders (transformers) and the VAE (diffusers AutoencoderKL) are third-party models and stay stock
PyTorch-ROCm modules; they must be available locally — there is no hub access on the target boxes.  The MM-DiT, the processors, the
rectified-flow inversions and the transfer loop run on the native HIP library."""
import json
import os

import torch


def load_transformer(pretrained_model_path, weight_dtype=torch.float16):
    """``transformer/`` of a local SD3 / SD3.5 checkpoint folder -> the native CustomSD3Transformer2DModel (config.json keys are
    diffusers'; weights from *.safetensors under diffusers' parameter names)."""
    from ...backbones.video_diffusion_sd3.models.transformer_3D_model import CustomSD3Transformer2DModel
    root = os.path.join(pretrained_model_path, "transformer")
    with open(os.path.join(root, "config.json")) as f:
        cfg = {k: v for k, v in json.load(f).items() if not k.startswith("_")}
    keys = ("sample_size", "patch_size", "in_channels", "num_layers", "attention_head_dim", "num_attention_heads", "joint_attention_dim",
            "caption_projection_dim", "pooled_projection_dim", "out_channels", "pos_embed_max_size", "dual_attention_layers", "qk_norm")
    model = CustomSD3Transformer2DModel(**{k: cfg[k] for k in keys if k in cfg})
    from safetensors.torch import load_file
    sd = {}
    for fn in sorted(os.listdir(root)):
        if fn.endswith(".safetensors"):
            sd.update(load_file(os.path.join(root, fn)))
    model.load_state_dict(sd, strict=True)
    if weight_dtype != torch.float16:
        print(f"[univst_amd] the native MM-DiT computes in fp16; --weight_dtype {weight_dtype} applies to the stock VAE / text encoders only")
    return model.half().cuda().requires_grad_(False)


def build_pipeline(pretrained_model_path, weight_dtype=torch.float16):
    from transformers import CLIPTextModelWithProjection, CLIPTokenizer, T5EncoderModel, T5TokenizerFast
    try:
        from diffusers import AutoencoderKL, FlowMatchEulerDiscreteScheduler
    except ImportError as e:  # pragma: no cover
        raise RuntimeError("the CLI needs `diffusers` for the SD3 VAE (third-party model, not re-implemented); the native MM-DiT / pipeline "
                           "classes themselves do not") from e
    from ...backbones.video_diffusion_sd3.pipelines.custom_pipeline import CustomStableDiffusion3Pipeline
    from ...backbones.video_diffusion_sd3.pnp_utils import CrossFrameProcessor
    sub = lambda cls, name: cls.from_pretrained(pretrained_model_path, subfolder=name)          # noqa: E731
    transformer = load_transformer(pretrained_model_path, weight_dtype)
    transformer.set_attn_processor({n: CrossFrameProcessor() for n in transformer.attn_processors})      # run_*_sd3.py:58-69
    enc = lambda cls, name: sub(cls, name).requires_grad_(False).to(weight_dtype).cuda()         # noqa: E731
    return CustomStableDiffusion3Pipeline(
        tokenizer=sub(CLIPTokenizer, "tokenizer"), tokenizer_2=sub(CLIPTokenizer, "tokenizer_2"), tokenizer_3=sub(T5TokenizerFast, "tokenizer_3"),
        text_encoder=enc(CLIPTextModelWithProjection, "text_encoder"), text_encoder_2=enc(CLIPTextModelWithProjection, "text_encoder_2"),
        text_encoder_3=enc(T5EncoderModel, "text_encoder_3"), vae=enc(AutoencoderKL, "vae"), transformer=transformer,
        scheduler=sub(FlowMatchEulerDiscreteScheduler, "scheduler"))


def add_common_args(parser, weight_dtype=torch.float16):
    parser.add_argument("--pretrained_model_path", type=str, default="stabilityai/stable-diffusion-3.5-medium")
    parser.add_argument("--weight_dtype", type=torch.dtype, default=weight_dtype)
    parser.add_argument("--time_steps", type=int, default=50)
    parser.add_argument("--seed", type=int, default=33)
    return parser

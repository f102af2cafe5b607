"""CLI mirror of src/sd/run_video_style_transfer_sd.py (reference :74-83 flags)."""
import argparse
import os

from ._common import add_common_args, build_pipeline
from ...backbones.video_diffusion_sd.pnp_utils import latent_adain, register_spatial_attention_pnp
from ..util import load_ddim_latents_at_t, save_folder, seed_everything


def main(a):
    if a.seed is not None:
        seed_everything(a.seed)
    pipe, _ = build_pipeline(a.pretrained_model_path, a.weight_dtype)
    content_noise = load_ddim_latents_at_t(a.time_steps, ddim_latents_path=a.content_inv_path).to(a.weight_dtype).cuda()
    style_noise = load_ddim_latents_at_t(a.time_steps, ddim_latents_path=a.style_inv_path).to(a.weight_dtype).cuda()
    latents = latent_adain(content_noise, style_noise)                     # init latent-shift
    register_spatial_attention_pnp(pipe)                                   # PnP: AdaIN-guided attention injection
    sample = pipe.video_style_transfer("", latents=latents, num_inference_steps=a.time_steps,
                                       content_inv_path=a.content_inv_path, style_inv_path=a.style_inv_path,
                                       mask_path=a.mask_path or None, skip_dead_branches=a.skip_dead_branches).images
    sample = sample.permute(0, 4, 1, 2, 3).contiguous()
    out = os.path.join(a.output_path, "sd", f'{a.content_inv_path.split("/")[-2]}_{a.style_inv_path.split("/")[-2]}')
    os.makedirs(out, exist_ok=True)
    save_folder(sample, out)


def parser():
    p = add_common_args(argparse.ArgumentParser())
    p.add_argument("--content_inv_path", type=str, default="results/contents-inv/sd/mallard-fly/inversion")
    p.add_argument("--style_inv_path", type=str, default="results/styles-inv/sd/00033/inversion")
    p.add_argument("--mask_path", type=str, default="results/masks/sd/mallard-fly")
    p.add_argument("--output_path", type=str, default="results/stylizations")
    p.add_argument("--skip_dead_branches", action="store_true",
                   help="extra: drop the content/style branches once the PnP window is closed (identical output)")
    return p


if __name__ == "__main__":
    main(parser().parse_args())

"""CLI mirror of src/sd3/run_video_style_transfer_sd3.py (reference :107-120 flags).  The reference file does not run at HEAD
(`from util import`, and a nested same-quote f-string that is a SyntaxError on its pinned Python 3.10); the output
folder name below is what that f-string spells: <content clip>_<style name>, the [-2] path components of the two inversion paths."""
import argparse
import os

import torch

from ._common import add_common_args, build_pipeline
from ...backbones.video_diffusion_sd3.pnp_utils import latent_adain, register_spatial_attention_pnp
from ..util import load_ddim_latents_at_t, seed_everything


def main(a):
    from ...parallel import init_distributed
    init_distributed()            # torchrun --nproc-per-node N: the pipeline shards the frames over the N GPUs; rank 0 writes the PNGs
    if a.seed is not None:
        seed_everything(a.seed)
    pipe = build_pipeline(a.pretrained_model_path, a.weight_dtype)
    content_inv_noises = load_ddim_latents_at_t(a.time_steps, ddim_latents_path=a.content_inv_path).to(a.weight_dtype).cuda()
    content_inv_latents = load_ddim_latents_at_t(0, ddim_latents_path=a.content_inv_path).to(a.weight_dtype).cuda()
    style_inv_noises = load_ddim_latents_at_t(a.time_steps, ddim_latents_path=a.style_inv_path).to(a.weight_dtype).cuda()
    content_inv_noises = latent_adain(content_inv_noises, style_inv_noises)          # init latent-shift, [f, c, h, w]
    register_spatial_attention_pnp(pipe)
    samples = pipe.video_style_transfer("", latents=content_inv_noises, img_latents=content_inv_latents, num_inference_steps=a.time_steps,
                                        content_inv_path=a.content_inv_path, style_inv_path=a.style_inv_path, mask_path=a.mask_path,
                                        eta_base=0.85, eta_trend="constant", start_step=25, end_step=39,
                                        shard=False if a.no_shard else None).images
    if samples is None:
        return
    out = os.path.join(a.output_path, "sd3", f"{a.content_inv_path.split('/')[-2]}_{a.style_inv_path.split('/')[-2]}")
    os.makedirs(out, exist_ok=True)
    for idx, sample in enumerate(samples):
        sample.save(os.path.join(out, "%05d.png" % idx))


def parser():
    p = add_common_args(argparse.ArgumentParser())
    p.set_defaults(pretrained_model_path="stabilityai/stable-diffusion-3-medium-diffusers")
    p.add_argument("--content_inv_path", type=str,
                   default="results/contents-inv/sd3/mallard-fly/inversion/content-inv/sd3-rf-solver/davis2016/blackswan/inversion")
    p.add_argument("--style_inv_path", type=str, default="results/styles-inv/sd3/00033/inversion")
    p.add_argument("--mask_path", type=str, default="results/masks/sd3/mallard-fly")
    p.add_argument("--output_path", type=str, default="output/")
    p.add_argument("--no_shard", action="store_true", help="under torchrun: keep every rank on the whole clip (no frame sharding)")
    return p


if __name__ == "__main__":
    main(parser().parse_args())

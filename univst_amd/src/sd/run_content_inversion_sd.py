"""CLI mirror of src/sd/run_content_inversion_sd.py (reference :77-91 flags)."""
import argparse
import os

import torch

from ._common import add_common_args, build_pipeline
from ...inversion_tools.ddim_inversion import content_inversion_reconstruction
from ..util import seed_everything


def main(a):
    if a.seed is not None:
        seed_everything(a.seed)
    pipe, DDIMScheduler = build_pipeline(a.pretrained_model_path, a.weight_dtype)
    inv_sched = DDIMScheduler.from_pretrained(a.pretrained_model_path, subfolder="scheduler")
    inv_sched.set_timesteps(a.time_steps)
    out = os.path.join(a.output_path, "sd", a.content_path.split("/")[-1])
    paths = {k: os.path.join(out, k) for k in ("inversion", "reconstruction", "features")}
    for p in paths.values():
        os.makedirs(p, exist_ok=True)
    with torch.no_grad():
        content_inversion_reconstruction(pipe, inv_sched, a.content_path, paths["inversion"], paths["reconstruction"], a.num_frames,
                                         a.height, a.width, a.time_steps, a.weight_dtype, ft_indices=[a.ft_indices],
                                         ft_timesteps=[a.ft_timesteps], ft_path=paths["features"], is_opt=a.is_opt,
                                         reconstruct=not a.skip_reconstruction)


def parser():
    p = add_common_args(argparse.ArgumentParser())
    p.add_argument("--content_path", type=str, default="examples/contents/mallard-fly")
    p.add_argument("--output_path", type=str, default="results/contents-inv")
    p.add_argument("--num_frames", type=int, default=16)
    p.add_argument("--height", type=int, default=512)
    p.add_argument("--width", type=int, default=512)
    p.add_argument("--ft_indices", type=int, default=2)
    p.add_argument("--ft_timesteps", type=int, default=301)
    p.add_argument("--is_opt", action="store_true", help="use Easy-Inv")
    p.add_argument("--skip_reconstruction", action="store_true", help="extra: skip the preview reconstruction (50 UNet calls)")
    return p


if __name__ == "__main__":
    main(parser().parse_args())

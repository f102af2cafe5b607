"""CLI mirror of src/sd/run_style_inversion_sd.py."""
import argparse
import os

import torch

from ._common import add_common_args, build_pipeline
from ...inversion_tools.ddim_inversion import style_inversion_reconstruction
from ..util import seed_everything


def main(a):
    if a.seed is not None:
        seed_everything(a.seed)
    pipe, DDIMScheduler = build_pipeline(a.pretrained_model_path, a.weight_dtype)
    inv_sched = DDIMScheduler.from_pretrained(a.pretrained_model_path, subfolder="scheduler")
    inv_sched.set_timesteps(a.time_steps)
    out = os.path.join(a.output_path, "sd", a.style_path.split("/")[-1].split(".")[0])
    inv, rec = os.path.join(out, "inversion"), os.path.join(out, "reconstruction")
    os.makedirs(inv, exist_ok=True)
    os.makedirs(rec, exist_ok=True)
    with torch.no_grad():
        style_inversion_reconstruction(pipe, inv_sched, a.style_path, inv, rec, a.num_frames, a.height, a.width, a.time_steps,
                                       a.weight_dtype, is_opt=a.is_opt, reconstruct=not a.skip_reconstruction)


def parser():
    p = add_common_args(argparse.ArgumentParser())
    p.add_argument("--style_path", type=str, default="examples/styles/00033.png")
    p.add_argument("--output_path", type=str, default="results/styles-inv")
    p.add_argument("--num_frames", type=int, default=16)
    p.add_argument("--height", type=int, default=512)
    p.add_argument("--width", type=int, default=512)
    p.add_argument("--is_opt", action="store_true", help="use Easy-Inv")
    p.add_argument("--skip_reconstruction", action="store_true")
    return p


if __name__ == "__main__":
    main(parser().parse_args())

"""CLI mirror of src/sd3/run_style_inversion_sd3.py (reference :90-106 flags)."""
import argparse
import os

import torch

from ._common import add_common_args, build_pipeline
from ...inversion_tools.flow_inversion import style_inversion_reconstruction
from ..util import seed_everything


def main(a):
    if a.seed is not None:
        seed_everything(a.seed)
    pipe = build_pipeline(a.pretrained_model_path, a.weight_dtype)
    out = os.path.join(a.output_path, "sd3", a.style_path.split("/")[-1].split(".")[0])
    paths = {k: os.path.join(out, k) for k in ("inversion", "reconstruction")}
    for p in paths.values():
        os.makedirs(p, exist_ok=True)
    with torch.no_grad():
        style_inversion_reconstruction(pipe, a.style_path, paths["inversion"], paths["reconstruction"], a.num_frames, a.height, a.width,
                                       a.time_steps, a.weight_dtype, is_rf_solver=a.is_rf_solver, reconstruct=not a.skip_reconstruction)


def parser():
    p = add_common_args(argparse.ArgumentParser())
    p.add_argument("--style_path", type=str, default="examples/styles/00033.png")
    p.add_argument("--output_path", type=str, default="results/styles-inv")
    p.add_argument("--num_frames", type=int, default=16)
    p.add_argument("--height", type=int, default=1024)
    p.add_argument("--width", type=int, default=1024)
    p.add_argument("--is_rf_solver", action="store_true", help="use rf-solver")
    p.add_argument("--skip_reconstruction", action="store_true", help="extra: skip the preview reconstruction (50 transformer calls)")
    return p


if __name__ == "__main__":
    main(parser().parse_args())

"""CLI mirror of src/sd3/run_content_inversion_sd3.py (reference :100-118 flags)."""
import argparse
import os

import torch

from ._common import add_common_args, build_pipeline
from ...inversion_tools.flow_inversion import content_inversion_reconstruction
from ..util import seed_everything


def main(a):
    if a.seed is not None:
        seed_everything(a.seed)
    pipe = build_pipeline(a.pretrained_model_path, a.weight_dtype)
    out = os.path.join(a.output_path, "sd3", a.content_path.split("/")[-1])
    paths = {k: os.path.join(out, k) for k in ("inversion", "reconstruction", "features")}
    for p in paths.values():
        os.makedirs(p, exist_ok=True)
    with torch.no_grad():
        content_inversion_reconstruction(pipe, a.content_path, paths["inversion"], paths["reconstruction"], a.num_frames, a.height, a.width,
                                         a.time_steps, a.weight_dtype, ft_indices=[a.ft_indices], ft_timesteps=[a.ft_timesteps],
                                         ft_path=paths["features"], is_rf_solver=a.is_rf_solver, reconstruct=not a.skip_reconstruction)


def parser():
    p = add_common_args(argparse.ArgumentParser(), weight_dtype=torch.bfloat16)
    p.add_argument("--content_path", type=str, default="examples/contents/mallard-fly")
    p.add_argument("--output_path", type=str, default="results/contents-inv")
    p.add_argument("--num_frames", type=int, default=16)
    p.add_argument("--height", type=int, default=1024)
    p.add_argument("--width", type=int, default=1024)
    p.add_argument("--ft_indices", type=int, default=20)
    p.add_argument("--ft_timesteps", type=int, default=5)
    p.add_argument("--is_rf_solver", action="store_true", help="use rf-solver")
    p.add_argument("--skip_reconstruction", action="store_true", help="extra: skip the preview reconstruction (50 transformer calls)")
    return p


if __name__ == "__main__":
    main(parser().parse_args())

"""CLI mirror of src/sd/run_video_style_transfer_sd.py (reference :74-83 flags).

Multi-GPU: ``torchrun --nproc-per-node N -m univst_amd.src.sd.run_video_style_transfer_sd ...`` (or the top-level shim
``src/sd/run_video_style_transfer_sd.py``) shards the clip's frames over the N GPUs of the node (``parallel.FrameShard``: K/V halo +
GroupNorm statistics through the library's IPC communicator, RCCL callbacks as the fall-back); rank 0 decodes and writes the PNGs.
``--smoother pixel`` switches on the sliding-window flow smoothing of stable_diffusion.py:713-759 (dead code in the reference:
``smoother = None`` is hard-wired at :715) with torchvision's RAFT-large as in src/cal_optica_flow.py; ``--smoother latent`` the
latent-space variant (flows of the content frames, ``--content_path``)."""
import argparse
import os

from ._common import add_common_args, build_pipeline
from ...backbones.video_diffusion_sd.pnp_utils import latent_adain, register_spatial_attention_pnp
from ..util import load_ddim_latents_at_t, save_folder, seed_everything


def smoother_kwargs(a, pipe, n_frames, hw=(64, 64)):
    """--smoother {none,pixel,latent} -> the keyword arguments of ``video_style_transfer`` (both need masks: the reference's window
    block composites with ``mask`` at stable_diffusion.py:751)."""
    if a.smoother in (None, "none"):
        return {}
    if not a.mask_path:
        raise SystemExit("--smoother needs --mask_path (stable_diffusion.py:751 composites the smoothed frames with the mask)")
    from ..cal_optica_flow import make_raft_flow_fn, make_latent_flows
    flow_fn = make_raft_flow_fn("cuda")          # torchvision raft_large + Raft_Large_Weights.DEFAULT (third-party, as in the reference)
    if a.smoother == "pixel":
        return {"smoother": "pixel", "flow_fn": flow_fn}
    if not a.content_path:
        raise SystemExit("--smoother latent needs --content_path (the content frames whose motion the smoother follows)")
    from ...inversion_tools.ddim_inversion import read_content_pixels
    px = read_content_pixels(a.content_path, n_frames, 8 * hw[0], 8 * hw[1])          # [F,3,H,W] float in [-1,1]
    frames = ((px + 1.0) * 127.5).round().clamp(0, 255).to("cuda").byte().permute(0, 2, 3, 1).contiguous()
    return {"smoother": "latent", "latent_flows": make_latent_flows(frames, flow_fn)}


def main(a):
    from ...parallel import init_distributed
    rank, world = init_distributed()             # (0, 1) and a no-op unless started by torchrun / torch.distributed.run
    if a.seed is not None:
        seed_everything(a.seed)
    pipe, _ = build_pipeline(a.pretrained_model_path, a.weight_dtype)
    content_noise = load_ddim_latents_at_t(a.time_steps, ddim_latents_path=a.content_inv_path).to(a.weight_dtype).cuda()
    style_noise = load_ddim_latents_at_t(a.time_steps, ddim_latents_path=a.style_inv_path).to(a.weight_dtype).cuda()
    latents = latent_adain(content_noise, style_noise)                     # init latent-shift
    register_spatial_attention_pnp(pipe)                                   # PnP: AdaIN-guided attention injection
    sample = pipe.video_style_transfer("", latents=latents, num_inference_steps=a.time_steps,
                                       content_inv_path=a.content_inv_path, style_inv_path=a.style_inv_path,
                                       mask_path=a.mask_path or None, skip_dead_branches=a.skip_dead_branches,
                                       shard=False if a.no_shard else None, **smoother_kwargs(a, pipe, latents.shape[2], latents.shape[-2:])).images
    if sample is None:            # ranks > 0 of a frame-sharded run: rank 0 decodes and writes
        return
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
    p.add_argument("--smoother", choices=["none", "pixel", "latent"], default="none",
                   help="extra: sliding-window optical-flow smoothing on steps 20..24 (stable_diffusion.py:713-759; the reference hard-wires None)")
    p.add_argument("--content_path", type=str, default="", help="content frames (folder / .mp4) for --smoother latent")
    p.add_argument("--no_shard", action="store_true", help="under torchrun: keep every rank on the whole clip (no frame sharding)")
    return p


if __name__ == "__main__":
    main(parser().parse_args())

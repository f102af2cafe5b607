"""The drop-in surface of INTEGRATION.md level 1: with this repository first on the path, the reference's import paths resolve to
the native implementation and expose the names the reference's scripts use (reference files cited per entry).  CPU only: nothing is
computed, the modules are imported and their public names compared with the package they re-export."""
import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SURFACE = {
    # reference module path -> (product module, names the reference's callers import from it)
    "backbones.video_diffusion_sd.pnp_utils": ("univst_amd.backbones.video_diffusion_sd.pnp_utils",
                                               ["register_time", "register_spatial_attention_pnp", "attention_adain", "latent_adain"]),      # pnp_utils.py:7,18,114,128
    "backbones.video_diffusion_sd.models.unet_3d_condition": ("univst_amd.backbones.video_diffusion_sd.models.unet_3d_condition",
                                                              ["UNetPseudo3DConditionModel"]),                                              # unet_3d_condition.py:37
    "backbones.video_diffusion_sd.pipelines.stable_diffusion": ("univst_amd.backbones.video_diffusion_sd.pipelines.stable_diffusion",
                                                                ["SpatioTemporalStableDiffusionPipeline"]),                                 # stable_diffusion.py:45
    "inversion_tools.ddim_inversion": ("univst_amd.inversion_tools.ddim_inversion",
                                       ["ddim_inversion", "ddim_loop", "ddim_loop_plus", "next_step", "content_inversion_reconstruction",
                                        "style_inversion_reconstruction", "load_video_frames"]),                                           # ddim_inversion.py:16-212
    "src.util": ("univst_amd.src.util", ["load_ddim_latents_at_t", "load_mask", "save_folder", "save_videos_grid", "save_images_as_mp4",
                                         "load_image", "seed_everything"]),                                                                # util.py
    "src.mask_propagation": ("univst_amd.src.mask_propagation", ["video_mask_propogation", "mask_propogation", "read_feature", "norm_mask",
                                                                 "to_one_hot"]),                                                            # mask_propagation.py:15-140
    "src.cal_optica_flow": ("univst_amd.src.cal_optica_flow", ["get_warp"]),                                                               # cal_optica_flow.py:51
    # first vertical slice of the SD3 / SD3.5 path: the plugin's processors + helpers and the rectified-flow inversions
    "backbones.video_diffusion_sd3.pnp_utils": ("univst_amd.backbones.video_diffusion_sd3.pnp_utils",
                                                ["CrossFrameProcessor", "AttentionShiftProcessor", "register_spatial_attention_pnp",
                                                 "attention_adain", "latent_adain"]),                                                      # video_diffusion_sd3/pnp_utils.py:9,135,276,289,305
    "inversion_tools.flow_inversion": ("univst_amd.inversion_tools.flow_inversion",
                                       ["rf_inversion", "rf_solver", "content_inversion_reconstruction", "style_inversion_reconstruction"]),  # flow_inversion.py:16-264
    "backbones.video_diffusion_sd3.models.transformer_3D_model": ("univst_amd.backbones.video_diffusion_sd3.models.transformer_3D_model",
                                                                  ["CustomSD3Transformer2DModel"]),                                        # transformer_3D_model.py:12
    "backbones.video_diffusion_sd3.pipelines.custom_pipeline": ("univst_amd.backbones.video_diffusion_sd3.pipelines.custom_pipeline",
                                                                ["CustomStableDiffusion3Pipeline"]),                                      # custom_pipeline.py:17
}


@pytest.mark.parametrize("ref_path", sorted(SURFACE))
def test_reference_import_paths_resolve_to_the_native_package(ref_path):
    prod_path, names = SURFACE[ref_path]
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    shim, prod = importlib.import_module(ref_path), importlib.import_module(prod_path)
    assert os.path.abspath(shim.__file__).startswith(ROOT), f"{ref_path} resolved outside this repository: {shim.__file__}"
    for n in names:
        assert hasattr(prod, n), f"{prod_path} lacks {n}"
        assert getattr(shim, n) is getattr(prod, n), f"{ref_path}.{n} is not the native implementation's"


@pytest.mark.parametrize("script", ["run_content_inversion_sd", "run_style_inversion_sd", "run_video_style_transfer_sd",
                                    "run_content_inversion_sd3", "run_style_inversion_sd3", "run_video_style_transfer_sd3"])
def test_cli_scripts_keep_the_reference_flags(script):
    """src/sd/run_*_sd.py and src/sd3/run_*_sd3.py --help work without a GPU and list the reference's flags (their argparse blocks)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "src", "sd3" if script.endswith("sd3") else "sd", script + ".py"), "--help"],
                         capture_output=True, text=True,
                         cwd=ROOT, env=dict(os.environ, PYTHONPATH=ROOT), timeout=120)
    assert out.returncode == 0, out.stderr[-400:]
    # the argparse blocks of the reference's three scripts (src/sd/run_*_sd.py)
    want = {"run_content_inversion_sd": ["--pretrained_model_path", "--content_path", "--output_path", "--weight_dtype", "--num_frames", "--height",
                                         "--width", "--time_steps", "--ft_indices", "--ft_timesteps", "--is_opt", "--seed"],
            "run_style_inversion_sd": ["--pretrained_model_path", "--style_path", "--output_path", "--weight_dtype", "--num_frames", "--height",
                                       "--width", "--time_steps", "--is_opt", "--seed"],
            "run_video_style_transfer_sd": ["--pretrained_model_path", "--content_inv_path", "--style_inv_path", "--mask_path", "--output_path",
                                            "--weight_dtype", "--time_steps", "--seed"],
            "run_content_inversion_sd3": ["--pretrained_model_path", "--content_path", "--output_path", "--weight_dtype", "--num_frames", "--height",
                                          "--width", "--time_steps", "--ft_indices", "--ft_timesteps", "--is_rf_solver", "--seed"],
            "run_style_inversion_sd3": ["--pretrained_model_path", "--style_path", "--output_path", "--weight_dtype", "--num_frames", "--height",
                                        "--width", "--time_steps", "--is_rf_solver", "--seed"],
            "run_video_style_transfer_sd3": ["--pretrained_model_path", "--content_inv_path", "--style_inv_path", "--mask_path", "--output_path",
                                             "--weight_dtype", "--time_steps", "--seed"]}[script]
    for flag in want:
        assert flag in out.stdout, f"{script}: flag {flag} missing from --help"
    # round 4: the multi-GPU / smoother switches of the transfer scripts (additions; the reference's flags are untouched)
    if script == "run_video_style_transfer_sd":
        for flag in ("--smoother", "--content_path", "--no_shard", "--skip_dead_branches"):
            assert flag in out.stdout, f"{script}: flag {flag} missing from --help"
        assert "{none,pixel,latent}" in out.stdout
    if script == "run_video_style_transfer_sd3":
        assert "--no_shard" in out.stdout


def test_mp4_content_input_goes_through_decord(monkeypatch, tmp_path):
    """ddim_inversion.py:21-27: VideoReader(path, width=, height=), first num_frames frames, /127.5 - 1, (b f) c h w.  decord is not in
    this image, so a stand-in module records the calls; without it the branch raises ImportError naming decord."""
    import sys
    import types
    import numpy as np
    import torch
    from univst_amd.inversion_tools import ddim_inversion as di

    monkeypatch.setitem(sys.modules, "decord", None)
    import pytest
    with pytest.raises(ImportError, match="decord"):
        di.read_content_pixels(str(tmp_path / "clip.mp4"), 4, 32, 48)

    calls = {}
    rng = np.random.default_rng(0)
    frames = torch.from_numpy(rng.integers(0, 256, (9, 32, 48, 3), dtype=np.uint8))

    class Reader:
        def __init__(self, path, width=None, height=None):
            calls["open"] = (path, width, height)

        def __len__(self):
            return frames.shape[0]

        def get_batch(self, idx):
            calls["idx"] = list(idx)
            return frames[idx]

    fake = types.ModuleType("decord")
    fake.VideoReader = Reader
    fake.bridge = types.SimpleNamespace(set_bridge=lambda name: calls.setdefault("bridge", name))
    monkeypatch.setitem(sys.modules, "decord", fake)
    px = di.read_content_pixels(str(tmp_path / "clip.mp4"), 4, 32, 48)
    assert calls == {"bridge": "torch", "open": (str(tmp_path / "clip.mp4"), 48, 32), "idx": [0, 1, 2, 3]}
    assert px.shape == (4, 3, 32, 48) and px.dtype == torch.float32
    want = (frames[:4].float() / 127.5 - 1.0).permute(0, 3, 1, 2)
    assert torch.equal(px, want)

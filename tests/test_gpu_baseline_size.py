"""Parity at the BASELINE size (SD-v1.5 widths, F = 16 frames, 64x64 latents = 16x512x512, three branches, 50 DDIM steps):
north_star's own acceptance numbers — output PSNR >= 40 dB vs the reference path, mask indices bit-exact.

The oracle (oracle/unet_ref + oracle/pipeline_ref, pinned to the reference's modules by tests/golden) is evaluated in
fp32 with torch ops ON THE DEVICE with the same fp16-valued weights (the CPU would need ~2 h for the 50 steps); the
native path goes through the pipeline mirror -> engine -> one C-ABI call per UNet step.  Reference lines followed:
backbones/video_diffusion_sd/pipelines/stable_diffusion.py:680-766, src/mask_propagation.py:15-99.

Every test writes the numbers it measured to gpurun_out/parity_baseline_size.json (copied into DESIGN.md / BASELINE.md).
"""
import json
import math
import os
import time
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import unet_ref, pipeline_ref, maskprop_ref, synth_inputs as si  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEEP = (0, 25, 26, 40, 41, 45, 46, 49)


def record(key, value):
    path = os.path.join(ROOT, "gpurun_out", "parity_baseline_size.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    data = {}
    if os.path.exists(path):
        with open(path) as f:
            data = json.load(f)
    data[key] = value
    with open(path, "w") as f:
        json.dump(data, f, indent=1, sort_keys=True)


def errs(got, ref):
    got, ref = got.float(), ref.float().to(got.device)
    e = got - ref
    return e.abs().max().item() / ref.abs().max().item(), (e.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt()).item()


def psnr(got, ref):
    got, ref = got.float(), ref.float().to(got.device)
    mse = (got - ref).pow(2).mean().item()
    peak = (ref.max() - ref.min()).item()
    return float("inf") if mse == 0 else 10 * math.log10(peak * peak / mse)


@pytest.fixture(scope="module")
def sd15():
    """the synthetic SD-v1.5-shaped UNet (1.06 B parameters) + its weights as fp32 tensors for the oracle (same values)."""
    from univst_amd import synth
    unet = synth.build_unet(device="cuda", seed=7)
    sd = {k: v.float() for k, v in unet.state_dict().items()}
    unet_ref.SDPA_MAX_BATCH = 4            # fp32 math SDPA: 4*8*4096*12288*4 B = 6.4 GB of scores per call at most
    yield unet, sd
    unet_ref.SDPA_MAX_BATCH = None


class _Tok:
    model_max_length = 77

    def __call__(self, prompt, **kw):
        n = len(prompt) if isinstance(prompt, list) else 1
        return types.SimpleNamespace(input_ids=torch.zeros(n, 77, dtype=torch.long), attention_mask=None)


def _enc(text):
    class Enc(torch.nn.Module):
        config = types.SimpleNamespace()

        def forward(self, ids, attention_mask=None):
            return (text.half().cuda().expand(ids.shape[0], -1, -1),)
    return Enc()


@pytest.mark.parametrize("F_", [16, 2])
def test_f16_forward_inside_and_outside_pnp_window(sd15, F_):
    """one three-branch forward at F = 16 (48 frames x 4096 tokens: the big-tile GEMM / tap-inner conv / d=40 pipelined
    attention dispatches the headline bench runs, 16 distinct key-source frames) inside (idx 12) and outside (idx 40)
    the PnP window.  Tolerance: max err <= 5e-3 * max|ref|, relative RMS <= 3e-3 over 22 ResBlocks + 16 transformer
    blocks of fp16 storage (measured on MI355X: 1.6e-3 / 1.6e-3; the F = 2 test keeps the looser round-1 bound).
    F = 2 is the problem size of ONE RANK of an 8-GPU job (6 frame-branches): the 128-wide kernels with the LayerNorm fold, split-K
    at the deep levels, the one-launch GroupNorm — same bar."""
    from univst_amd.backbones.video_diffusion_sd import pnp_utils
    unet, sd = sd15
    cfg = unet_ref.SD15_CONFIG
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3, 4, F_, 64, 64, generator=g).half().cuda()
    ctx = torch.randn(1, 77, 768, generator=g).half().cuda().expand(3, -1, -1).contiguous()
    pipe = types.SimpleNamespace(unet=unet)
    pnp_utils.register_spatial_attention_pnp(pipe)
    out = {}
    for idx, t in ((12, 741), (40, 181)):
        pnp_utils.register_time(pipe, idx)
        got = unet(x, t, encoder_hidden_states=ctx).sample
        with torch.no_grad():
            ref, _ = unet_ref.unet_forward(sd, cfg, x.float(), t, ctx.float(), pnp_idx=idx, exact_temporal=False)
        mx, rms = errs(got, ref)
        out[f"idx{idx}"] = dict(max_rel=mx, rms_rel=rms)
        assert torch.isfinite(got.float()).all()
        assert mx < 5e-3 and rms < 3e-3, (idx, mx, rms)
    record("f16_forward" if F_ == 16 else f"f{F_}_forward", out)


@pytest.mark.parametrize("tag", ["mask", "nomask"])
def test_f16_fifty_step_transfer_vs_oracle_on_device(sd15, tag):
    """BASELINE config 3 without the optional smoother: the 50-step three-branch localized transfer at 16x512x512 with
    moving-disc masks, native pipeline vs the oracle loop (fp32, device), same synthetic inversion trajectories.
    PSNR of the stylised latents >= 40 dB at i in {0,25,26,40,41,45,46,49} (every branch of the step logic)."""
    from univst_amd.backbones.video_diffusion_sd.pipelines.stable_diffusion import SpatioTemporalStableDiffusionPipeline
    from univst_amd.backbones.video_diffusion_sd import pnp_utils
    from univst_amd.schedulers import DDIMScheduler
    unet, sd = sd15
    cfg = unet_ref.SD15_CONFIG
    F_, h, w, n = 16, 64, 64, 50
    text = si.text_embedding(768)
    pipe = SpatioTemporalStableDiffusionPipeline(vae=None, text_encoder=_enc(text), tokenizer=_Tok(), unet=unet, scheduler=DDIMScheduler())
    ci = [si.content_latent(k, F_, h, w).half() for k in range(n + 1)]
    sy = [si.style_latent(k, F_, h, w).half() for k in range(n + 1)]
    masks = torch.from_numpy(pipeline_ref.mask_from_png_values(si.disc_masks(F_, 512, 512)))[None] if tag == "mask" else None
    lat0 = pnp_utils.latent_adain(ci[n].cuda(), sy[n].cuda())
    pnp_utils.register_spatial_attention_pnp(pipe)
    got = {}
    t0 = time.time()
    out = pipe.video_style_transfer("", latents=lat0, num_inference_steps=n, content_inv_latents=ci, style_inv_latents=sy,
                                    masks=masks, output_type="latent",
                                    callback=lambda i, t, l: got.__setitem__(i, l.clone()) if i in KEEP else None).images
    torch.cuda.synchronize()
    t_native = time.time() - t0
    ctx = text.half().float().cuda().expand(3, -1, -1).contiguous()
    cif = [t.float().cuda() for t in ci]
    syf = [t.float().cuda() for t in sy]
    ref = {}
    t0 = time.time()
    with torch.no_grad():
        pipeline_ref.video_style_transfer_loop(
            lambda x, t, i: unet_ref.unet_forward(sd, cfg, x, int(t), ctx, pnp_idx=i, exact_temporal=False)[0],
            pipeline_ref.DDIMSchedule(), unet_ref.latent_adain(cif[n], syf[n]), cif, syf,
            masks.cuda() if masks is not None else None, n,
            callback=lambda i, t, l: ref.__setitem__(i, l.clone()) if i in KEEP else None)
    torch.cuda.synchronize()
    t_oracle = time.time() - t0
    vals = {f"i{i}": psnr(got[i], ref[i]) for i in KEEP}
    record(f"transfer50_{tag}", dict(psnr_db=vals, native_s=t_native, oracle_fp32_device_s=t_oracle))
    assert torch.equal(out, got[49])
    for i in KEEP:
        assert torch.isfinite(got[i].float()).all() and vals[f"i{i}"] >= 40.0, vals


def test_f16_fifty_step_easy_inversion_vs_oracle_on_device(sd15):
    """BASELINE config 2 (and the producer of config 3's inputs): the 50-step single-branch Easy-Inv inversion
    (inversion_tools/ddim_inversion.py:116-167) at 16x512x512 with the t = 301 feature dump, native engine vs the oracle loop
    (fp32, device).  PSNR >= 40 dB on ddim_latents_k for k in {1,12,13,25,50}; the dumped up_blocks[2] features (the input of mask
    propagation) within 5e-3 relative RMS of the oracle forward at the same latent, and within 5e-2 of the oracle loop's own dump
    (15 steps of drift through a random-weight UNet)."""
    from univst_amd import engine
    from univst_amd.schedulers import DDIMScheduler
    from univst_amd.backbones.video_diffusion_sd.pnp_utils import PNP_LAYERS
    unet, sd = sd15
    for r, bs in PNP_LAYERS.items():       # the shared UNet may carry the PnP registration of an earlier test: inversion runs the stock layers
        for b in bs:
            unet.up_blocks[r].attentions[b].transformer_blocks[0].attn1.__dict__.pop("_univst_native_pnp", None)
    cfg = unet_ref.SD15_CONFIG
    F_, h, w, n = 16, 64, 64, 50
    z0 = (0.8 * si.content_latent(0, F_, h, w)).half()
    text = si.text_embedding(768).half()
    pipe = types.SimpleNamespace(unet=unet)
    sched = DDIMScheduler()
    sched.set_timesteps(n)
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        t0 = time.time()
        got = engine.inversion_loop(pipe, sched, z0.cuda(), text.cuda(), n, True, ft_indices=[2], ft_timesteps=[301], ft_path=td)
        torch.cuda.synchronize()
        t_native = time.time() - t0
        feat = torch.load(os.path.join(td, "inversion_feature_map_2_block_301_step.pt"))
    osch = pipeline_ref.DDIMSchedule()
    osch.set_timesteps(n)
    ctx = text.float().cuda()
    dump = {}

    def eps_fn(z, t, i):
        e, f = unet_ref.unet_forward(sd, cfg, z, int(t), ctx, None, ft_indices=[2] if int(t) == 301 else None, exact_temporal=False)
        if f:
            dump["feat"] = f[2]
        return e
    t0 = time.time()
    with torch.no_grad():
        ref = pipeline_ref.ddim_inversion_loop(eps_fn, osch, z0.float().cuda(), n, True)
    torch.cuda.synchronize()
    t_oracle = time.time() - t0
    vals = {f"k{k}": psnr(got[k], ref[k]) for k in (1, 12, 13, 25, 50)}
    mx, rms = errs(feat, dump["feat"])                  # includes 15 steps of trajectory drift, amplified by the (random-weight) UNet
    with torch.no_grad():                               # the dump itself: oracle forward at the NATIVE loop's own latent of that step (t = 301 is i = 15)
        _, f_same = unet_ref.unet_forward(sd, cfg, got[15].float(), 301, ctx, None, ft_indices=[2], exact_temporal=False)
    mx1, rms1 = errs(feat, f_same[2])
    record("inversion50_easy", dict(psnr_db=vals, feature_dump_along_trajectories=dict(max_rel=mx, rms_rel=rms),
                                    feature_dump_same_input=dict(max_rel=mx1, rms_rel=rms1), native_s=t_native, oracle_fp32_device_s=t_oracle))
    assert tuple(feat.shape) == (F_, 64, 64, 640) and feat.dtype == torch.float16
    assert all(v >= 40.0 for v in vals.values()), vals
    assert rms1 < 5e-3 and rms < 5e-2, (mx1, rms1, mx, rms)


def test_sd21_widths_step_vs_oracle_on_device():
    """SURVEY §8f-3 at the real size: the SD-v2.1 layout (Linear proj_in / proj_out, head_dim 64 at every level => 5/10/20/20
    heads, 1024-wide text states) with SD widths, 64x64 latents, three branches, F = 4, inside and outside the PnP window, native
    graph vs the oracle in fp32 on the device.  Same bound as the SD-v1.5 forward (max 5e-3, relative RMS 3e-3)."""
    from univst_amd import synth
    from univst_amd.backbones.video_diffusion_sd import pnp_utils
    unet = synth.build_unet(config=synth.SD21_UNET_CONFIG, device="cuda", seed=9)
    sd = {k: v.float() for k, v in unet.state_dict().items()}
    cfg = unet_ref.SD21_CONFIG
    unet_ref.SDPA_MAX_BATCH = 4
    try:
        g = torch.Generator().manual_seed(5)
        x = torch.randn(3, 4, 4, 64, 64, generator=g).half().cuda()
        ctx = torch.randn(1, 77, 1024, generator=g).half().cuda().expand(3, -1, -1).contiguous()
        pipe = types.SimpleNamespace(unet=unet)
        pnp_utils.register_spatial_attention_pnp(pipe)
        out = {}
        for idx, t in ((12, 741), (40, 181)):
            pnp_utils.register_time(pipe, idx)
            got = unet(x, t, encoder_hidden_states=ctx).sample
            with torch.no_grad():
                ref, _ = unet_ref.unet_forward(sd, cfg, x.float(), t, ctx.float(), pnp_idx=idx, exact_temporal=False)
            mx, rms = errs(got, ref)
            out[f"idx{idx}"] = dict(max_rel=mx, rms_rel=rms)
            assert torch.isfinite(got.float()).all() and mx < 5e-3 and rms < 3e-3, (idx, mx, rms)
        record("sd21_forward_f4", out)
    finally:
        unet_ref.SDPA_MAX_BATCH = None


def test_mask_propagation_bit_exact_full_size():
    """mask indices bit-exact at the BASELINE size: 16 x 64x64 x 640 features (the up_blocks[2] dump), a multi-valued
    anti-aliased 512^2 first mask (256 one-hot classes), 512^2 outputs — native kernels vs oracle/maskprop_ref on the CPU,
    same torch.manual_seed(33) randperm stream (src/mask_propagation.py:15-99)."""
    from univst_amd.src import mask_propagation as mp
    feats = si.maskprop_features(F=16, h=64, w=64, C=640, seed=11)
    first = si.soft_first_mask(512, 512)
    assert int(np.array(first).max()) == 255 and len(np.unique(first)) > 100
    args = mp.build_parser().parse_args([])
    torch.manual_seed(33)
    t0 = time.time()
    got = np.stack(mp.propagate_masks(feats, first, args))
    t_native = time.time() - t0
    torch.manual_seed(33)
    t0 = time.time()
    ref = np.stack(maskprop_ref.video_mask_propagation(feats, first))
    t_oracle = time.time() - t0
    diff = [int((a != b).sum()) for a, b in zip(got, ref)]
    record("maskprop_full", dict(mismatching_pixels_per_frame=diff, native_s=t_native, oracle_cpu_s=t_oracle,
                                 foreground_px_last=int((ref[-1] != 0).sum())))
    assert got.shape == ref.shape == (16, 512, 512) and got.dtype == np.uint8
    assert sum(diff) == 0, diff


def test_warp_and_sliding_window_bit_exact_16x512x512():
    """SURVEY §8 a16 at the BASELINE size: the fused occlusion + fixed-point remap + composite kernel and the whole Gauss-Seidel
    sliding window (stable_diffusion.py:723-751, cal_optica_flow.py:20-46: 58 warps, 116 flows) on 16 frames of 512x512, BIT-EXACT
    against oracle/flow_ref (numpy; its remap is pinned by the hand-computed OpenCV vectors of tests/golden/remap_opencv_cases.json).
    The call-counted flows cover sub-pixel offsets on and off the 1/32 grid, flows that leave the image on every border, and
    patches that trip the 1.5 px forward-backward test; a 30 % keep-mask is restored at the end."""
    from oracle import flow_ref
    from univst_amd.src import cal_optica_flow as cf
    H = W = 512
    F_ = 16
    rs = np.random.RandomState(11)
    frames = rs.randint(0, 256, (1, 3, F_, H, W)).astype(np.uint8)
    frames[0, :, 3, 100:200, 50:300] = 255          # saturated and flat regions: rounding at 255 / exact reproduction
    frames[0, :, 7, 300:400, :] = 0
    mask = (rs.rand(F_, H, W) > 0.7).astype(np.uint8)

    class Flows:
        def __init__(self):
            self.k = 0

        def __call__(self, a=None, b=None):
            k = self.k
            self.k += 1
            sg = 1.0 if k % 2 == 0 else -1.0
            f = si.translation_flow(H, W, sg * (0.03125 * (k % 64) + (k % 5)), -sg * (0.015625 * (k % 32) + (k % 3)), 900 + k, noise=0.3)
            if k % 3 == 0:
                f[:, :12, 0] -= 20.0                 # left border: samples from outside the image
                f[-9:, :, 1] += 17.0                 # bottom border
            if k % 2 == 1:
                f[H // 8:H // 4, W // 4:W // 2] += 3.0      # occluded patch (forward + backward no longer cancel)
            return f.astype(np.float32)
    fo, fn = Flows(), Flows()
    t0 = time.time()
    ref = flow_ref.sliding_window_smooth(frames, fo, mask[None])
    t_oracle = time.time() - t0
    torch.cuda.synchronize()
    t0 = time.time()
    got = cf.sliding_window_smooth(torch.from_numpy(frames).cuda(), lambda a, b: torch.from_numpy(fn()).cuda(), torch.from_numpy(mask).cuda())
    torch.cuda.synchronize()
    t_native = time.time() - t0
    got = got.cpu().numpy()
    assert fo.k == fn.k == 2 * 58
    ndiff = int((got != ref).sum())
    record("sliding_window_16x512x512", dict(differing_values=ndiff, of=int(ref.size), native_s=t_native, oracle_numpy_s=t_oracle,
                                             changed_by_smoothing=int((ref != frames).sum())))
    assert ndiff == 0, f"{ndiff} of {ref.size} smoothed values differ"
    assert (ref != frames).mean() > 0.3, "the window must actually change the unmasked pixels"


def test_f16_pixel_smoother_leg_vs_oracle_on_device(sd15):
    """BASELINE config 3's smoothing leg at the BASELINE size (stable_diffusion.py:713-759, 782-834): SD-v1.5 widths, F = 16, 64x64
    latents = 16 x 512 x 512 frames, steps 0..25 of the localized transfer with the pixel smoother active on steps 20..24 —
    pred_original_sample, VAE decode to uint8, HIP sliding window (58 warps per step) with the masked restore, VAE encode,
    return_to_timestep — native pipeline vs the oracle loop (UNet oracle in fp32 on the device, flow_ref smoothing in numpy), with
    the deterministic linear fake VAE and call-counted analytic flows in place of the third-party VAE / RAFT.
    PSNR >= 40 dB on the latents after steps 19..25; the un-smoothed run must be clearly further away (the leg is reproduced)."""
    from univst_amd.backbones.video_diffusion_sd.pipelines.stable_diffusion import SpatioTemporalStableDiffusionPipeline
    from univst_amd.backbones.video_diffusion_sd import pnp_utils
    from univst_amd.schedulers import DDIMScheduler
    unet, sd = sd15
    cfg = unet_ref.SD15_CONFIG
    F_, h, w, n = 16, 64, 64, 50
    text = si.text_embedding(768)
    vae = si.FakeLinearVAE().cuda()
    pipe = SpatioTemporalStableDiffusionPipeline(vae=vae, text_encoder=_enc(text), tokenizer=_Tok(), unet=unet, scheduler=DDIMScheduler())
    ci = [si.content_latent(k, F_, h, w).half() for k in range(n + 1)]
    sy = [si.style_latent(k, F_, h, w).half() for k in range(n + 1)]
    masks = torch.from_numpy(pipeline_ref.mask_from_png_values(si.disc_masks(F_, 512, 512)))[None]
    lat0 = pnp_utils.latent_adain(ci[n].cuda(), sy[n].cuda())
    pnp_utils.register_spatial_attention_pnp(pipe)
    keep = (19, 20, 21, 22, 23, 24, 25)

    class Stop(Exception):
        pass

    def collect(store):
        def cb(i, t, l):
            if i in keep:
                store[i] = l.clone()
            if i == keep[-1]:
                raise Stop()
        return cb
    gflow = si.CountingFlow(512, 512)
    got, plain = {}, {}
    t0 = time.time()
    with pytest.raises(Stop):
        pipe.video_style_transfer("", latents=lat0, num_inference_steps=n, content_inv_latents=ci, style_inv_latents=sy, masks=masks,
                                  output_type="latent", smoother="pixel", flow_fn=lambda a, b: torch.from_numpy(gflow()).cuda(),
                                  callback=collect(got))
    torch.cuda.synchronize()
    t_native = time.time() - t0
    assert gflow.k == 5 * 2 * 58
    with pytest.raises(Stop):
        pipe.video_style_transfer("", latents=lat0, num_inference_steps=n, content_inv_latents=ci, style_inv_latents=sy, masks=masks,
                                  output_type="latent", callback=collect(plain))
    ctx = text.half().float().cuda().expand(3, -1, -1).contiguous()
    cif = [t.float().cuda() for t in ci]
    syf = [t.float().cuda() for t in sy]
    osch = pipeline_ref.DDIMSchedule()
    osch.set_timesteps(n)
    oflow = si.CountingFlow(512, 512)
    ref = {}
    t0 = time.time()
    with torch.no_grad(), pytest.raises(Stop):
        pipeline_ref.video_style_transfer_loop(
            lambda x, t, i: unet_ref.unet_forward(sd, cfg, x, int(t), ctx, pnp_idx=i, exact_temporal=False)[0],
            osch, unet_ref.latent_adain(cif[n], syf[n]), cif, syf, masks.cuda(), n,
            smoother=pipeline_ref.pixel_smoother(osch, vae.decode_tensor, lambda x: vae.encode_tensor(x.cuda()), oflow, masks.numpy()),
            callback=collect(ref))
    torch.cuda.synchronize()
    t_oracle = time.time() - t0
    vals = {f"i{i}": psnr(got[i], ref[i]) for i in keep}
    off = {f"i{i}": psnr(plain[i], ref[i]) for i in keep}
    record("pixel_smoother_leg_f16", dict(psnr_db=vals, unsmoothed_psnr_db=off, native_s=t_native, oracle_s=t_oracle))
    assert torch.equal(plain[19], got[19]) and not torch.equal(plain[20], got[20])
    assert all(v >= 40.0 for v in vals.values()), vals
    assert all(vals[f"i{i}"] >= off[f"i{i}"] + 6.0 for i in (20, 21, 22, 23, 24)), (vals, off)


def test_trained_temporal_layers_at_sd15_widths_vs_oracle_on_device():
    """a fine-tuned 3-D checkpoint at SD-v1.5 widths (320 .. 1280 channels, head dims 40 / 80 / 160): every conv_temporal perturbed
    away from the dirac kernel, every attn_temporal.to_out away from zero — three branches x 4 frames x 32x32 latents inside the PnP
    window, native graph (Conv1d over frames through the implicit-GEMM conv kernels, frame attention kernel) vs the oracle's exact
    temporal path in fp32 on the device (oracle pinned to the reference's modules by G4 / G5-general).  Same bar as the 2-D forward:
    max <= 5e-3 of max|ref|, relative RMS <= 3e-3; and the run must differ clearly from the identity-initialised one."""
    from univst_amd import synth
    from univst_amd.backbones.video_diffusion_sd import pnp_utils
    unet = synth.build_unet(device="cuda", seed=17)
    g = torch.Generator(device="cuda").manual_seed(5)
    with torch.no_grad():
        for name, p_ in unet.named_parameters():
            if "conv_temporal.weight" in name:
                p_.add_((0.35 / math.sqrt(3 * p_.shape[1])) * torch.randn(p_.shape, generator=g, device="cuda").to(p_))
            elif "conv_temporal.bias" in name:
                p_.add_(0.02 * torch.randn(p_.shape, generator=g, device="cuda").to(p_))
            elif "attn_temporal.to_out.0.weight" in name:
                p_.add_((0.5 / math.sqrt(p_.shape[1])) * torch.randn(p_.shape, generator=g, device="cuda").to(p_))
    sd = {k: v.float() for k, v in unet.state_dict().items()}
    cfg = unet_ref.SD15_CONFIG
    F_ = 4
    gc = torch.Generator().manual_seed(3)
    x = torch.randn(3, 4, F_, 32, 32, generator=gc).half().cuda()
    ctx = torch.randn(1, 77, 768, generator=gc).half().cuda().expand(3, -1, -1).contiguous()
    pipe = types.SimpleNamespace(unet=unet)
    pnp_utils.register_spatial_attention_pnp(pipe)
    pnp_utils.register_time(pipe, 12)
    got = unet(x, 741, encoder_hidden_states=ctx).sample
    with torch.no_grad():
        ref, _ = unet_ref.unet_forward(sd, cfg, x.float(), 741, ctx.float(), pnp_idx=12, exact_temporal=True)
        plain, _ = unet_ref.unet_forward(sd, cfg, x.float(), 741, ctx.float(), pnp_idx=12, exact_temporal=False)
    mx, rms = errs(got, ref)
    record("trained_temporal_sd15_widths", dict(max_rel=mx, rms_rel=rms, temporal_effect_rms=errs(plain, ref)[1]))
    assert torch.isfinite(got.float()).all()
    assert mx < 5e-3 and rms < 3e-3, (mx, rms)
    assert errs(plain, ref)[1] > 3e-2, "the perturbed temporal layers must matter"

#!/usr/bin/env python
"""bench.py — stylized frames/sec of the UniVST SD-v1.5 three-branch denoising loop on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]            (N>1: one rank per GPU — launched by torch.distributed.run, or,
                                                                    when no launcher environment is present, re-executed under it)

Workload (BASELINE.json metric): SD-v1.5 geometry UNet (random-init synthetic weights of that architecture,
fp16), 16 frames x 512x512 (latents [1,4,16,64,64]), 50 DDIM steps of the three-branch transfer loop
(content-inv | style-inv | stylised) with AdaIN-guided attention injection on steps 0..25 and the latent AdaIN
on steps 41..45, all three branches computed on every step (reference-equivalent work, 46.8 TFLOP/step).
One "step" = one DDIM step of that loop (K = 50: the whole schedule; K < 50: K steps spread evenly over it, i = floor(50 j / K), so the
mix of in-window / out-of-window steps is the loop's; K > 50 wraps modulo 50);
value = F / (50 * mean step time) frames/s — with the default K = 50 that is exactly one full transfer.
Inputs are resident in HBM before the timed region.  Prints ONE JSON line (rank 0).

The default workload is the headline (BASELINE config 3 without the optional smoother: localized transfer, moving-disc
masks blended on steps 0..45).  --workload selects the other driver-timed lines (same JSON shape, own `roofline`):
  transfer_nomask   the same loop without masks
  inversion         BASELINE config 2: the single-branch DDIM inversion loop (one step = one single-branch UNet call + next_step)
  inversion_pair    the content + style inversions of one job as one batch-2 trajectory (value = frames of BOTH clips / s)
  vae_decode        the final decode of the clip (and each decode of the pixel smoother leg): the SVD temporal VAE decoder on the native library,
                    16 x 4 x 64 x 64 latents -> 16 x 3 x 512 x 512 frames (random-init weights of that architecture)
  maskprop          point-matching mask propagation of a 16-frame clip (64x64x640 features, 256 classes, 512^2 masks); HBM-bound
  warp              one sliding-window smoothing pass over 16 x 512^2 frames (58 occlusion + remap + blend warps in 16 launches, one per key frame); HBM-bound
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_FP16_TFLOPS = 2500.0   # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md (not the 2:1-sparse marketing figure)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--latent", type=int, default=None, help="latent side (default: 64 = 512 px; 128 = 1024 px for --workload sd3_transfer)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl == RCCL; gloo stages through the host, bring-up only)")
    ap.add_argument("--comm", default=None, choices=["ipc", "dist"],
                    help="frame-shard communicator for --gpus > 1: ipc = the library's own (HIP IPC peer writes + device flags, no host "
                         "callbacks; default, falls back to dist if the mapping fails), dist = torch.distributed callbacks (RCCL / gloo)")
    ap.add_argument("--emulate-wire", default=None, metavar="GBPS[,LAT_US]",
                    help="with --emulate-rank: model the wire too — every K/V exchange keeps the stream busy for pack bytes / GBPS (per-direction rate of one "
                         "xGMI link; rank 1's link from rank 0 carries two packs) and every GroupNorm all-reduce for LAT_US (default 3); serial, as csrc/comm.hip "
                         "issues them today")
    ap.add_argument("--comm-emulated", action="store_true",
                    help="with --emulate-rank and --emulate-wire: drive the rank through the library's OWN communicator in its emulated mode "
                         "(univst_comm_connect_emulated: production kernels, forked stream and flag waits, transfers replaced by delays) instead of "
                         "host callbacks that return at once; required for ranks > 0 of --workload sd3_transfer")
    ap.add_argument("--emulate-rank", default=None, metavar="R/W",
                    help="diagnostic: run the work of rank R of a W-GPU job alone on this GPU with no-op collectives (kernels, pack/unpack "
                         "and host callbacks of a frame shard, no wire time); prints the usual line with parallelism 'emulated R/W'")
    ap.add_argument("--workload", default="transfer", choices=["transfer", "transfer_nomask", "inversion", "inversion_pair", "maskprop", "warp", "sd3_transfer", "vae_decode"])
    ap.add_argument("--model", default="sd15", choices=["sd15", "sd21"], help="UNet configuration: SD-v1.5 (headline) or the SD-v2.1 layout "
                                                                            "(Linear projections, head_dim 64, 1024-wide text states; SURVEY §8f-3)")
    ap.add_argument("--full-cpu", action="store_true", help="cpu_baseline: time the two representative steps at the full frame count "
                                                             "(no extrapolation in F; ~2 min on 16 threads); default: one step at the full count")
    ap.add_argument("--quick-cpu", action="store_true", help="cpu_baseline: the F=2 sample of rounds 1-3 (14 s, extrapolated x8 in F)")
    ap.add_argument("--selftest-launch", action="store_true", help="multi-rank plumbing only (rendezvous, barriers, MAX over ranks, one JSON "
                                                                    "line); no GPU work, runs on a CPU box with --backend gloo")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-shard-check", action="store_true", help="multi-rank runs: skip the sharded-vs-unsharded forward check before timing")
    ap.add_argument("--no-skip-dead-branches-leg", action="store_true",
                    help="do not also time the variant that drops branches 0/1 after the PnP window (identical output; reported "
                         "separately in config, never as `value`)")
    return ap.parse_args()


class Pipe:
    def __init__(self, unet, scheduler):
        self.unet, self.scheduler = unet, scheduler


def make_step_fn(pipe, content, style, text3, mask_m, n=50):
    """returns step(i, latents) -> latents : one iteration of engine.transfer_loop (stable_diffusion.py:681-766)."""
    from univst_amd import _native, engine
    from univst_amd.backbones.video_diffusion_sd.pnp_utils import latent_adain, register_time
    ts = pipe.scheduler.timesteps

    def step(i, latents):
        i = i % n
        t = ts[i]
        c_t, s_t = content[n - i], style[n - i]
        if mask_m is not None and i <= 0.9 * n:
            latents = _native.mask_blend(latents, c_t, mask_m)
        if i > 0.8 * n and i <= 0.9 * n:
            latents = _native.mask_blend(latent_adain(latents, s_t), c_t, mask_m)
        register_time(pipe, i)
        x = torch.cat([c_t, s_t, latents])
        eps = pipe.unet(x, t, encoder_hidden_states=text3).sample[2:3]
        return engine.ddim_step(pipe.scheduler, eps, t, latents)
    return step


def usable_cores():
    """threads the container may actually run: min(affinity, cgroup cpu quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        pass
    return n


def cpu_baseline(frames_full, unet=None, full=False, single_branch=False, quick=False):
    """oracle ('port' of the reference algorithm, fp32 PyTorch CPU ops, reference plumbing incl. the dead temporal
    ops) on this box's host cores, weighted to the 26 in-window + 24 out-of-window steps of the loop.
    default: ONE three-branch UNet step inside the PnP window MEASURED at the full frame count (about a minute on 16 threads; no
             extrapolation in F: the F = 2 sample of rounds 1-3 over-estimated the CPU by 15-50 %, its activations being cache-resident),
             the out-of-window step taken as that time x the ratio of the two kinds of step measured at F = 2 (7 s each);
    full (--full-cpu): both steps measured at the full frame count;  quick (--quick-cpu): both at F = 2, extrapolated linearly in F.
    single_branch: the inversion step (no PnP)."""
    from oracle import unet_ref, synth_inputs as si
    cores = usable_cores()
    torch.set_num_threads(cores)
    cfg = unet_ref.SD15_CONFIG
    t0 = time.time()
    if unet is not None:      # the very weights the GPU run used (fp16 values, upcast), copied D2H: seconds instead of a minute
        sd = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}
    else:
        sd = unet_ref.synth_state_dict(cfg, seed=33)

    def inputs(F_s):
        if single_branch:
            return si.content_latent(40, F_s, 64, 64), si.text_embedding(768)
        return (torch.cat([si.content_latent(40, F_s, 64, 64), si.style_latent(40, F_s, 64, 64), si.content_latent(39, F_s, 64, 64)]),
                si.text_embedding(768).expand(3, -1, -1).contiguous())

    def step(x, ctx, inside):
        t = time.time()
        with torch.no_grad():
            unet_ref.unet_forward(sd, cfg, x, 781 if inside else 381, ctx, pnp_idx=None if single_branch else (10 if inside else 30), exact_temporal=True)
        return time.time() - t

    x2, ctx = inputs(2)
    step(x2[:, :, :1].contiguous(), ctx, True)      # untimed warm-up (thread pool, allocator, oneDNN primitive caches): the first CPU step used to be 20 % slow
    t1 = time.time()
    what = "single-branch UNet step (inversion)" if single_branch else "three-branch UNet steps (one inside the PnP window, one outside)"
    if full or quick:
        F_s = frames_full if full else 2
        x, ctx = inputs(F_s)
        t_in = step(x, ctx, True)
        t_out = t_in if single_branch else step(x, ctx, False)
        scale = frames_full / F_s
        how = (f"{1 if single_branch else 2} {what}: {t_in:.1f} s / {t_out:.1f} s at F={F_s} of {frames_full} frames"
               + ("" if full else f", EXTRAPOLATED x{frames_full // F_s} in F"))
        extrap = not full
    else:
        a_in = step(x2, ctx, True)
        ratio = 1.0
        if not single_branch:          # the two kinds of step twice each, best of each: a 6 s step on 16 threads is noisy (one draw gave 0.70)
            a_out = step(x2, ctx, False)
            a_in = min(a_in, step(x2, ctx, True))
            a_out = min(a_out, step(x2, ctx, False))
            ratio = a_out / a_in
        x, ctx = inputs(frames_full)
        t_in = step(x, ctx, True)
        t_out, scale = t_in * ratio, 1.0
        how = (f"one {'single-branch' if single_branch else 'three-branch'} UNet step inside the PnP window MEASURED at the full F={frames_full}: {t_in:.1f} s; "
               f"the out-of-window step taken as x{ratio:.3f} (ratio of the two kinds of step at F=2: {a_in:.1f} s in-window)")
        extrap = False if single_branch else "ratio"      # 24 of the 50 step times are DERIVED (in-window time at full F x the F = 2 ratio), not measured
    # the 50-step loop has 26 steps inside the window (i = 0..25) and 24 outside; per-step cost has no other data dependence
    loop_full = (26 * t_in + 24 * t_out) * scale
    method = "full" if full else ("quick_F2_linear" if quick else "inwindow_fullF_outwindow_ratio")
    return dict(value=frames_full / loop_full, unit="frames/s", cores=cores, kind="port", extrapolated=extrap, method=method, method_since_round=4 if method.startswith("inwindow") else 1,
                derived_out_of_window=(method == "inwindow_fullF_outwindow_ratio" and not single_branch),
                sample=f"{how}; fp32, all temporal ops, 64x64 latents, on {cores} threads (cgroup quota); weighted to 26 + 24 steps "
                       f"(weight copy {t1 - t0:.0f} s excluded)")


def hbm_roofline(kernel, algorithmic_bytes, ms, launches):
    gbs = algorithmic_bytes / (ms * 1e-3) / 1e9
    return {"kernel": kernel, "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None, "avg_launch_ms": round(ms / launches, 4),
            "algorithmic_mb_per_launch": round(algorithmic_bytes / launches / 1e6, 2)}


def vae_decode_flops(cfg, imgs, F_, h, w):
    """algorithmic matrix flops of AutoencoderKLTemporalDecoder.decode on imgs frames of h x w latents (2 x MACs of every conv / linear / attention)"""
    boc, L, lat = cfg["block_out_channels"], cfg["layers_per_block"], cfg["latent_channels"]
    fl = 0.0
    px = h * w

    def st(ci, co, px):            # ResnetBlock2D (3x3 convs, 1x1 shortcut) + TemporalResnetBlock (two Conv3d (3,1,1))
        return 2.0 * px * (9 * ci * co + 9 * co * co + (ci * co if ci != co else 0) + 2 * 3 * co * co)
    fl += 2.0 * px * 9 * lat * boc[3]
    fl += L * st(boc[3], boc[3], px) + 2.0 * px * 4 * boc[3] * boc[3] + 4.0 * px * px * boc[3]         # mid block: resnets, q/k/v/out, QK^T + PV
    ci = boc[3]
    for b in range(4):
        co = boc[3 - b]
        for l in range(L + 1):
            fl += st(ci, co, px)
            ci = co
        if b < 3:
            px *= 4
            fl += 2.0 * px * 9 * co * co
    fl += 2.0 * px * 9 * boc[0] * cfg["out_channels"] + 2.0 * px * 3 * cfg["out_channels"] ** 2
    return fl * imgs


def run_vae_workload(a, dev):
    """the clip's decode on the native temporal VAE: value = decoded frames / s (one step = one decode of the whole clip)"""
    from univst_amd import synth, vae
    F_, h = a.frames, (a.latent or 64)
    cfg = synth.SVD_VAE_CONFIG
    v = vae.NativeTemporalVAE(synth.vae_state_dict(cfg, device=dev), cfg, device=dev)
    z = torch.randn(F_, 4, h, h, device=dev, dtype=torch.float16)
    for _ in range(max(1, a.warmup)):
        v.decode(z, num_frames=F_)
    torch.cuda.synchronize()
    reps = max(1, a.steps // 5)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(reps):
        out = v.decode(z, num_frames=F_).sample
    ev1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    dev_ms = ev0.elapsed_time(ev1) / reps
    fl = vae_decode_flops(cfg, F_, F_, h, h)
    assert torch.isfinite(out.float()).all()
    return {"metric": f"decoded frames/sec, SVD temporal VAE decoder, {F_}x{h * 8}x{h * 8}", "value": round(F_ / wall, 3), "unit": "frames/s", "n_gpus": 1,
            "steps": reps, "warmup": max(1, a.warmup), "ms_per_step": round(wall * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"svd_temporal_vae_decode_{F_}x{h * 8}x{h * 8}", "frames": F_, "parallelism": "single",
                       "weights": "random-init AutoencoderKLTemporalDecoder architecture (128, 256, 512, 512), fp16",
                       "algorithmic_tflop_per_decode": round(fl / 1e12, 2)},
            "roofline": {"kernel": "gemm_kernel<4,1> (128 x 128 implicit-GEMM conv tile: the 128-wide convs of the 512 x 512 level, 44 % of the decode's kernel time; the 256 / 512-wide convs run on the ragged 256 x 320 tile, 34 %; GroupNorm passes 18 % — rocprofv3 trace of round 6)",
                         "bound": "mfma", "achieved": round(fl / (dev_ms * 1e-3) / 1e12, 1), "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(fl / (dev_ms * 1e-3) / 1e12 / PEAK_FP16_TFLOPS, 4), "traffic": None,
                         "note": "whole-decode figure (all launches of one decode); off the 50-step loop: the metric's timed region ends before the final decode"}}


def run_aux_workload(a, dev):
    """the two HBM-bound producers / post-processors of the path as driver-timed lines (SURVEY §8d byte counts)."""
    from univst_amd.src import mask_propagation as mp, cal_optica_flow as cf
    from univst_amd import synth
    import numpy as np
    F_ = a.frames
    out_cfg = {"frames": F_, "parallelism": "single"}
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if a.workload == "maskprop":
        feats, first = synth.synth_maskprop_inputs(F=F_, h=64, w=64, C=640, H=512, W=512, seed=11, device=dev)
        args = mp.build_parser().parse_args([])
        args.num_frames = F_

        def once():
            torch.manual_seed(33)
            return mp.propagate_masks(feats, first, args)
        unit_desc = "propagated masks"
        units = F_ - 1
        # per frame: affinity matrix [Nsrc<=~13.7k, 4096] fp32 written once + read 3x (top-k threshold, normalise, label GEMM)
        # + the [256, 512, 512] fp32 upsample field is never materialised (fused finalize): ~0.9 GB at the full queue
        alg_bytes = None
    else:
        H = W = 512
        rs = np.random.RandomState(3)
        frames = torch.from_numpy(rs.randint(0, 256, (1, 3, F_, H, W)).astype(np.uint8)).to(dev)
        mask = torch.from_numpy((rs.rand(F_, H, W) > 0.7).astype(np.uint8)).to(dev)
        flows = [synth.synth_flow(H, W, 3.3 * ((k % 3) - 1), -2.7 * ((k % 2) * 2 - 1), 100 + k, noise=0.6, device=dev) for k in range(8)]
        cnt = [0]

        def flow_fn(x, y):
            cnt[0] += 1
            return flows[cnt[0] % 8]

        def once():
            return cf.sliding_window_smooth(frames, flow_fn, mask)
        unit_desc = "smoothed frames"
        units = F_
        nwarp = sum(1 for k in range(F_) for b in (-2, -1, 1, 2) if 0 <= k + b < F_)
        alg_bytes = nwarp * H * W * 25.0          # 3 src + 3 key + 8 fwd + 8 bwd + 3 out bytes per pixel and warp
    for _ in range(max(1, a.warmup)):
        once()
    torch.cuda.synchronize()
    reps = max(1, a.steps // 10)
    ev0.record()
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    ev1.record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps
    dev_ms = ev0.elapsed_time(ev1) / reps
    out = {"metric": f"{unit_desc}/sec, {F_}-frame clip at 512x512 ({a.workload})", "value": round(units / wall, 3), "unit": "frames/s",
           "n_gpus": 1, "steps": reps, "warmup": max(1, a.warmup), "ms_per_step": round(wall * 1e3, 3), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32" if a.workload == "maskprop" else "u8", "data": "synthetic",
           "config": dict(out_cfg, workload=("maskprop_16x64x64x640_256cls_512sq" if a.workload == "maskprop" else "sliding_window_warp_blend_16x512x512_r2"))}
    if a.workload == "warp":
        out["roofline"] = hbm_roofline("warp_window_key_kernel (occlusion test + fixed-point remap of up to 4 neighbours + window mean; one launch per key frame, 16 per pass)",
                                       alg_bytes, dev_ms, F_)
        out["roofline"]["note"] = ("58 warps in 16 launches (round 3: 58 + 32 launches); 26 MB per launch: still latency- / gather-bound at 512^2, and "
                                   "sequential over key frames like the reference (Gauss-Seidel)")
    else:
        # traffic model per frame: |aff| = Nsrc x 4096 fp32, 1 write (affinity GEMM) + 2 reads (top-k scan, threshold + sum: the survivors
        # leave as compact lists, the normalised matrix is never written back); Nsrc grows 4096 -> ~13.7k as the queue fills
        nsrc = [4096 + min(k, 9) * 1070 for k in range(F_ - 1)]
        alg_bytes = sum(n * 4096 * 4 * 3.0 for n in nsrc)
        out["roofline"] = hbm_roofline("maskprop_frame (row-normalise, fp32-MFMA affinity GEMM + exp, top-15 threshold + survivor lists, sparse label product) + finalize",
                                       alg_bytes, dev_ms, F_ - 1)
        out["roofline"]["note"] = ("host randperm sub-sampling between frames (reference RNG stream; one scalar per frame comes back from the device) "
                                   "is inside the timed region; the affinity GEMM (up to 72 GFLOP fp32 per frame on v_mfma_f32_32x32x2_f32) is the largest kernel and is "
                                   "COMPUTE-bound: rocprofv3 (round 4) 0.62 ms per frame on average = 116 TFLOP/s = 0.74 of the 157 TFLOP/s fp32 matrix peak, a third of the "
                                   "clip's kernel time (top-k 26 %, sparse label product 23 %) - the HBM fraction above under-rates this line")
    return out


def sd3_step_flops(cfg, B, N, T, shift_window=False):
    """algorithmic matrix flops of one MM-DiT forward on B frames of N image + T text tokens with the cross-frame key set
    [first | prev | cur] ++ text (3N + T keys per query)."""
    D, L = cfg.num_attention_heads * cfg.attention_head_dim, cfg.num_layers
    nd = len(cfg.dual_attention_layers)
    img_lin = L * (8 + 16) * D * D + nd * 8 * D * D                    # q,k,v,out + FF (4x) per image token; attn2 of the dual blocks
    txt_lin = (L - 1) * (8 + 16) * D * D + 6 * D * D                   # last block: context_pre_only (no to_add_out, no ff_context)
    attn = L * 4 * (N + T) * (3 * N + T) * D + nd * 4 * N * 3 * N * D
    emb = 2 * N * (cfg.in_channels * cfg.patch_size ** 2) * D + 2 * N * D * cfg.patch_size ** 2 * cfg.out_channels + 2 * T * cfg.joint_attention_dim * D
    return B * (N * img_lin + T * txt_lin + attn + emb)


def run_sd3_workload(a, dev, rank=0, world=1, dist=None):
    """BASELINE config 5 (one GPU, or frame-sharded over `world` ranks: K/V of the clip's first and of the previous frame travel
    through the library's IPC communicator inside every joint attention): the three-branch transfer step of the SD3.5-medium MM-DiT (24 blocks, 1536 wide, 13 dual-attention
    blocks) at 16 x 1024 x 1024 (4096 image + 333 text tokens per frame, batch 48), AttentionShiftProcessor registered, random-init
    weights, synthetic latents / prompt embeddings.  A step = mask-free loop iteration: cat -> transformer -> eta-interpolated Euler."""
    import types
    from univst_amd import _native
    from univst_amd.backbones.video_diffusion_sd3 import pnp_utils
    from univst_amd.backbones.video_diffusion_sd3.models.transformer_3D_model import sd35_medium
    from univst_amd.backbones.video_diffusion_sd3.pipelines.custom_pipeline import CustomStableDiffusion3Pipeline
    from univst_amd.schedulers import FlowMatchEulerDiscreteScheduler
    from univst_amd.parallel import Sd3FrameShard
    F_all, hl = a.frames, a.latent
    emu = None
    if a.emulate_rank and world == 1:
        # one rank of a w-GPU job alone, no wire: its F/w frames of every branch as whole clips of that length — the same kernels and key
        # counts ([first | previous | current] + text) as the sharded rank runs, minus the K | V pack copies and the exchange
        er, ew = (int(v) for v in a.emulate_rank.split("/"))
        emu = (er, ew)
        if a.comm_emulated:
            # any rank, through the library's communicator in emulated mode: the real sharded op (pack, post on the forked stream, two-phase joint
            # attention, flag wait, unpack, barrier) with the transfers replaced by delays of pack bytes / link rate
            from univst_amd.parallel import EmulatedIpcComm
            assert a.emulate_wire, "--comm-emulated needs --emulate-wire GBPS[,LAT_US]"
            wire = [float(v) for v in a.emulate_wire.split(",")]
            shard = Sd3FrameShard(er, ew, F_all, comm=EmulatedIpcComm(er, ew, 1 << 17, *wire))
        else:
            # exact for rank 0 only: its 'first' and 'previous' frames are local.  A rank r > 0 reads both from halo blocks, so each of its frames
            # has three DISTINCT key sources (rank 0 with two frames: one and two after merging duplicates) — that needs the communicator's inboxes
            assert er == 0, "--workload sd3_transfer --emulate-rank: only rank 0 can be emulated without a communicator (ranks > 0: --comm-emulated)"
            shard = Sd3FrameShard(0, 1, F_all // ew)
    else:
        shard = Sd3FrameShard(rank, world, F_all)
    F_ = shard.local
    torch.manual_seed(33)
    with torch.device(dev):
        model = sd35_medium()
    model = model.half().requires_grad_(False)
    pipe = CustomStableDiffusion3Pipeline(transformer=model, scheduler=FlowMatchEulerDiscreteScheduler())
    pnp_utils.register_spatial_attention_pnp(pipe)
    if emu is not None and not a.comm_emulated:
        for proc in model.attn_processors.values():
            proc.clip_length = F_
    shard.attach(model, tokens=(hl // 2) ** 2)
    g = torch.Generator(device=dev).manual_seed(5 + rank)
    rn = lambda *sh: torch.randn(*sh, generator=g, device=dev, dtype=torch.float16)          # noqa: E731
    T = 77 + 256
    pe, pp = rn(1, T, 4096).repeat(3 * F_, 1, 1), rn(1, 2048).repeat(3 * F_, 1)
    content = [rn(F_, 16, hl, hl) for _ in range(4)]
    style = [rn(F_, 16, hl, hl) for _ in range(4)]
    target = rn(F_, 16, hl, hl)
    ts, tl, ds = pipe._schedule(50)
    eta = pipe.generate_eta_values(tl, 25, 39, 0.85, "constant")

    def step(i, lat):
        i = i % 50
        c_t, s_t = content[i % 4], style[i % 4]
        if i >= 40 and i <= 45:
            lat = pnp_utils.latent_adain(lat, s_t)
        x = torch.cat([c_t, s_t, lat])
        v = model(hidden_states=x, timestep=ts[i].expand(3 * F_), encoder_hidden_states=pe, pooled_projections=pp, return_dict=False,
                  joint_attention_kwargs={"idx": i})[0]
        return pipe._euler_eta(lat, v[2 * F_:].contiguous(), target, ds[i], eta[i], tl[i] / 1000.0)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    idx = [(j * 50) // a.steps if a.steps < 50 else j % 50 for j in range(a.steps)]
    lat = rn(F_, 16, hl, hl)
    for i in range(max(1, a.warmup)):
        lat = step(idx[i % len(idx)], lat)
    sync()
    if emu is not None and a.comm_emulated:
        shard.comm.wire_us()                   # (reading resets the counter)
    t0 = time.perf_counter()
    for i in idx:
        lat = step(i, lat)
    sync()
    ms = (time.perf_counter() - t0) * 1e3 / a.steps
    sd3_wire_ms = (shard.comm.wire_us() / a.steps / 1e3) if (emu is not None and a.comm_emulated) else None
    if dist is not None:                      # MAX over ranks
        outs = [None] * world
        dist.all_gather_object(outs, ms)
        ms = max(outs)
    assert torch.isfinite(lat.float()).all(), "non-finite latents"
    N = (hl // 2) ** 2
    F_ = F_all
    fl = sd3_step_flops(model.config, 3 * F_, N, T)
    out = {"metric": f"stylized frames/sec, SD-v3.5-medium {F_}x{hl * 8}x{hl * 8} @50 rectified-flow steps (three-branch transfer)",
           "value": round(F_ / (50 * ms / 1e3), 4),
           "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": max(1, a.warmup), "ms_per_step": round(ms, 2), "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f16", "data": "synthetic latents + prompt embeddings, random-init weights (2.2 B parameters)",
           "config": {"workload": f"sd35_medium_mmdit_three_branch_transfer_{F_}x{hl * 8}x{hl * 8}_50rf", "frames": F_, "tokens_per_frame": N,
                      "text_tokens": T, "batch": 3 * F_, "parallelism": ((f"rank {emu[0]} of {emu[1]} emulated through the library's communicator (wire modelled at {a.emulate_wire} GB/s per link, "
                                                                           f"{'exchange on the forked stream' if os.environ.get('UNIVST_KV_OVERLAP', '1') != '0' else 'serial'}): {F_all // emu[1]} frames per branch"
                                                                           if a.comm_emulated else f"rank {emu[0]} of {emu[1]} emulated (no wire): {F_all // emu[1]} frames per branch") if emu else
                                                                          "single" if world == 1 else f"frames{world} (IPC communicator)"),
                      **({"modelled_wire_ms_per_step": round(sd3_wire_ms, 3)} if sd3_wire_ms is not None else {}),
                      "note": "BASELINE config 5 names 8 GPUs and fp8 QKV; fp16 here (the reference's --weight_dtype default); --gpus N shards the frames"}}
    tf = fl / (ms * 1e-3) / 1e12 / (emu[1] if emu else world)
    out["roofline"] = {"bound": "mfma", "kernel": "whole step (linears + joint attention), per GPU", "achieved": round(tf, 1), "peak": PEAK_FP16_TFLOPS,
                       "unit": "TFLOP/s", "frac": round(tf / PEAK_FP16_TFLOPS, 4), "traffic": None, "algorithmic_tflop_per_step": round(fl / 1e12, 2)}
    if not a.no_profile and world == 1:
        _native.profile_enable(True)
        lat2 = step(idx[0], lat)
        torch.cuda.synchronize()
        prof = _native.profile_collect()
        _native.profile_enable(False)
        del lat2
        out["kernel_classes_ms"] = {k: round(v["ms"], 2) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:10]} if isinstance(prof, dict) else prof
    out["cpu_baseline"] = None
    return out


def self_launch(a):
    """`python bench.py --gpus N` with no launcher environment: re-execute this command under torch.distributed.run with one rank per
    GPU (exactly what the driver's `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N` does), so the
    line printed has n_gpus = N either way.  Rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this driver (RCCL / cross-process device memory)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def launch_selftest(a, rank, world):
    """--selftest-launch: everything of a multi-rank run except the GPU work — rendezvous, barrier-bracketed timing of K empty steps,
    MAX over ranks, one JSON line from rank 0 with n_gpus = world.  Runs on a CPU-only box (gloo); tests/test_bench_launch.py."""
    import datetime
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(a.backend if a.backend != "nccl" or torch.cuda.is_available() else "gloo", timeout=datetime.timedelta(seconds=120))
    for _ in range(a.warmup):
        pass
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        time.sleep(0.001 * (1 + rank))           # ranks differ: the reported time must be the slowest rank's
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "launch selftest (no GPU work)", "value": None, "unit": "frames/s", "n_gpus": world, "steps": a.steps,
                          "warmup": a.warmup, "ms_per_step": round(dt.item() / a.steps * 1e3, 3), "higher_is_better": True,
                          "scaling": "strong", "vs_baseline": None, "dtype": "f16", "data": "none",
                          "config": {"workload": "selftest_launch", "parallelism": f"frames{world}", "backend": dist.get_backend()}}))
    dist.barrier()
    dist.destroy_process_group()


def main():
    a = parse()
    if a.latent is None:
        a.latent = 128 if a.workload == "sd3_transfer" else 64
    if a.comm:
        os.environ["UNIVST_COMM"] = a.comm
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ and not a.emulate_rank:
        sys.exit(self_launch(a))
    if a.selftest_launch:
        return launch_selftest(a, int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)))
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local = min(local, torch.cuda.device_count() - 1)      # (bring-up: several ranks may share one GPU with --backend gloo)
        torch.cuda.set_device(local)
        if -(-world // torch.cuda.device_count()) > 4 and "UNIVST_KV_OVERLAP" not in os.environ:
            # more than four ranks on ONE device: their forward + forked queues oversubscribe its hardware queues and a spinning wait kernel can starve a
            # peer's unmapped queue until a bounded wait gives up — the exchange stays on the forward's stream there (a bring-up case: one rank per GPU forks)
            os.environ["UNIVST_KV_OVERLAP"] = "0"
        import datetime
        tmo = datetime.timedelta(seconds=300)        # a rank that died leaves its peers in a collective: fail in minutes, not in NCCL's default 10
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=tmo)
        else:
            dist.init_process_group(a.backend, timeout=tmo)
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local)

    if a.workload in ("maskprop", "warp", "sd3_transfer", "vae_decode"):
        assert world == 1 or a.workload == "sd3_transfer", "maskprop / warp / the VAE couple all frames of a clip: replicas only (DESIGN.md §5)"
        from univst_amd import _native
        _native.load()
        out = (run_sd3_workload(a, dev, rank, world, dist) if a.workload == "sd3_transfer" else
               (run_vae_workload(a, dev) if a.workload == "vae_decode" else run_aux_workload(a, dev)))
        if rank == 0:
            print(json.dumps(out))
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    from univst_amd import _native, synth
    from univst_amd.backbones.video_diffusion_sd import pnp_utils
    from univst_amd.schedulers import DDIMScheduler
    from univst_amd.parallel import FrameShard

    F_total, h = a.frames, a.latent
    emu = None
    if a.emulate_rank and world == 1:
        from univst_amd.parallel import NullComm
        er, ew = (int(v) for v in a.emulate_rank.split("/"))
        emu = (er, ew)
        wire = [float(v) for v in a.emulate_wire.split(",")] if a.emulate_wire else None
        if a.comm_emulated:
            from univst_amd.parallel import EmulatedIpcComm
            assert wire, "--comm-emulated needs --emulate-wire GBPS[,LAT_US]"
            shard = FrameShard(er, ew, F_total, comm=EmulatedIpcComm(er, ew, 1 << 17, *wire))
        else:
            shard = FrameShard(er, ew, F_total, comm=NullComm(er, ew, *(wire or []), kv_in_library=True))
    else:
        shard = FrameShard(rank, world, F_total)
    unet = synth.build_unet(config=synth.SD21_UNET_CONFIG if a.model == "sd21" else None, device=dev, seed=33)
    pipe = Pipe(unet, DDIMScheduler())
    pipe.scheduler.set_timesteps(50)
    if not a.workload.startswith("inversion"):          # the inversion UNet is the stock one (run_content_inversion_sd.py never registers PnP)
        pnp_utils.register_spatial_attention_pnp(pipe)
    masked = a.workload == "transfer"
    content, style, text3, mask = synth.synth_transfer_inputs(F=F_total, h=h, w=h, D=1024 if a.model == "sd21" else 768, device=dev, with_mask=masked, mask_hw=8 * h)
    content_full, style_full = content[50], style[50]          # (kept for the sharded self-check below)
    content = [shard.slice_frames(t) for t in content]
    style = [shard.slice_frames(t) for t in style]
    mask_m = None
    if masked:          # load_mask() output {0,1} [1,F,512,512] -> bilinear /8 once (stable_diffusion.py:688-691), this rank's frames
        mask_m = _native.mask_resize(mask.reshape(-1, 8 * h, 8 * h).contiguous(), h, h)
        mask_m = mask_m.reshape(-1, h, h)[shard.f0:shard.f0 + shard.local].contiguous()
    shard.attach(unet, max_tokens=h * h)
    if emu is not None and a.emulate_wire and not a.comm_emulated:
        # the K/V exchanges' wire time is modelled INSIDE the library, on the forked stream the real multicast runs on (csrc/unet.hip Fwd::kv_post), so the
        # emulation shows what the overlap hides; the GroupNorm all-reduces' flag round trips stay in NullComm on the forward's stream
        unet.set_native_option("emu_wire_gbps", int(wire[0]))
        if len(wire) > 1:
            unet.set_native_option("emu_wire_lat_us", int(wire[1]))
    shard_check = None
    if world > 1 and not a.no_shard_check:
        # First contact with real multi-GPU hardware happens in the driver's run: before anything is timed, every rank compares ONE
        # frame-sharded forward (inside the PnP window: K/V exchange, GroupNorm all-reduces) with the unsharded forward of a SEPARATELY BUILT
        # native handle of the same weights (FrameShard.fresh_reference) on the whole clip.  A communicator that fails the check — wrong numbers, a refused mapping, a bounded wait
        # that gave up — is replaced: library IPC -> torch.distributed callbacks; if that fails too the run stops with the evidence.
        # (the check lives in the product: parallel.FrameShard.self_check is what video_style_transfer runs under torchrun too;
        # its reference is a second native handle built from the same module, never attached to a communicator)
        shard_check = shard.self_check(pipe, content_full, style_full, text3)
    sharded = world > 1 or emu is not None
    inv = a.workload.startswith("inversion")
    pair = a.workload == "inversion_pair"
    if inv:
        from univst_amd import engine
        assert not sharded, "--workload inversion: single GPU line (config 2)"
        pipe.unet = unet
        text1 = text3[2:3].contiguous() if not pair else text3[1:3].contiguous()
        ts = [int(t) for t in pipe.scheduler.timesteps.tolist()]
        lat = content[0] if not pair else torch.cat([content[0], style[0]])

        def step(i, z):
            t = ts[len(ts) - (i % 50) - 1]
            eps = unet(z, t, encoder_hidden_states=text1).sample
            return engine.next_step(eps, t, z, pipe.scheduler)
    if not inv:
        lat = shard.latent_adain(content[50], style[50]) if sharded else pnp_utils.latent_adain(content[50], style[50])
        step = shard.make_step_fn(pipe, content, style, text3, mask_m) if sharded else make_step_fn(pipe, content, style, text3, mask_m)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # which steps of the 50-step schedule are timed: all of them for K = 50 (wrapping for K > 50); for K < 50 the K steps are spread
    # evenly over the schedule (i_j = floor(50 j / K)) so that the share of steps inside the PnP window (26 of 50), the mask blend
    # (0..45) and the latent AdaIN (41..45) is that of the full loop — steps 0..K-1 would all be the (more expensive) in-window kind
    idx = [(j * 50) // a.steps if a.steps < 50 else j % 50 for j in range(a.steps)]
    for i in range(a.warmup):
        lat = step(i % 50, lat)
    sync()
    if emu is not None and a.comm_emulated:
        shard.comm.wire_us()
    elif emu is not None:
        shard.comm.wire_us = 0.0
        _native.unet_query(unet._native_handle, "emu_wire_us")
    t0 = time.perf_counter()
    for i in idx:
        lat = step(i, lat)
    sync()
    dt = time.perf_counter() - t0
    if emu is not None and a.comm_emulated:
        wire_ms_per_step = shard.comm.wire_us() / a.steps / 1e3
    else:
        wire_ms_per_step = ((shard.comm.wire_us + _native.unet_query(unet._native_handle, "emu_wire_us")) / a.steps / 1e3) if emu is not None and a.emulate_wire else None
    kv_overlap = os.environ.get("UNIVST_KV_OVERLAP", "1") != "0"
    if dist is not None:
        tmax = torch.tensor([dt], device=dev if a.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    assert torch.isfinite(lat.float()).all(), "non-finite latents"
    ms_per_step = dt / a.steps * 1e3
    value = F_total * (2 if a.workload == "inversion_pair" else 1) / (50 * ms_per_step / 1e3)

    out = {
        "metric": ("inverted frames/sec, SD-v1.5 16x512x512 @50 DDIM inversion steps" if inv else
                   "stylized frames/sec, SD-v1.5 16x512x512 @50 DDIM steps").replace("SD-v1.5", "SD-v2.1 layout" if a.model == "sd21" else "SD-v1.5"), "value": round(value, 4), "unit": "frames/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": ((f"sd15_unet_content_plus_style_ddim_inversion_batch2_{F_total}x{h * 8}x{h * 8}_50ddim" if pair else
                                 f"sd15_unet_single_branch_ddim_inversion_{F_total}x{h * 8}x{h * 8}_50ddim") if inv else
                                f"{a.model}_unet_three_branch_pnp_transfer_{F_total}x{h * 8}x{h * 8}_50ddim"), "frames": F_total,
                   "latent": [1, 4, F_total, h, h], "branches": (2 if pair else 1) if inv else 3,
                   "schedule_steps": "all 50" if a.steps == 50 else (f"{a.steps} of 50, evenly spread" if a.steps < 50 else f"{a.steps} (wrapping modulo 50)"),
                   "masks": "moving disc, blended on steps 0..45" if masked else None, "parallelism": (f"emulated rank {emu[0]}/{emu[1]} ({'wire modelled at ' + a.emulate_wire + ' GB/s per link, ' + ('K/V exchange on the forked stream' if kv_overlap else 'serial') + (', through the library communicator in emulated mode' if a.comm_emulated else '') if a.emulate_wire else 'no wire'})" if emu else "single") if world == 1 else f"frames{world}",
                   "comm": None if world == 1 else type(shard.comm).__name__, "shard_check": shard_check,
                   **({"modelled_wire_ms_per_step": round(wire_ms_per_step, 3)} if wire_ms_per_step is not None else {}),
                   "weights": ("random-init SD-v2.1 layout (Linear projections, head_dim 64, text width 1024), fp16" if a.model == "sd21" else
                               "random-init SD-v1.5 architecture (859M + 201M temporal params), fp16")},
    }

    if rank == 0 and not a.no_profile:
        # roofline leg: per-kernel-class HIP-event timing over a second pass of the same steps (events bracket every
        # launch on its stream; kept out of the timed region so the event overhead does not perturb `value`)
        _native.profile_enable(True)
    if not a.no_profile:
        nprof = min(a.steps, 50)
        l2 = lat
        for i in idx[:nprof]:
            l2 = step(i, l2)
        sync()
    if rank == 0 and not a.no_profile:
        prof = _native.profile_collect()
        launched = _native.profile_symbols()
        _native.profile_enable(False)
        classes = {}
        for k, v in prof.items():
            if v["launches"]:
                classes[k] = dict(ms_per_step=round(v["ms"] / nprof, 3), launches_per_step=round(v["launches"] / nprof, 1),
                                  tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] else None,
                                  gbs=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1),
                                  **({"im2col_operand_gbs": round(v["expanded_bytes"] / (v["ms"] * 1e-3) / 1e9, 1)} if k.startswith(("conv", "gemm_kernel<*,1>", "gemm_big_kernel<1>")) and v.get("expanded_bytes") else {}))
        mfma = [k for k in prof if k.startswith(("gemm", "attn", "conv")) and prof[k]["launches"]]
        dom = max(mfma, key=lambda k: prof[k]["ms"])
        d = prof[dom]
        ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
        out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / PEAK_FP16_TFLOPS, 4), "traffic": None,
                           "avg_launch_ms": round(d["ms"] / d["launches"], 4),
                           "algorithmic_gflop_per_launch": round(d["flops"] / d["launches"] / 1e9, 2),
                           "classes": classes,
                           # the class times come from a SECOND pass with a hipEvent pair around every launch: each pair fences its kernel off from its
                           # neighbours (no overlap of one kernel's tail with the next one's ramp), so their sum exceeds ms_per_step of the timed pass
                           "classes_sum_ms": round(sum(v["ms_per_step"] for v in classes.values()), 3),
                           "classes_launches_per_step": round(sum(v["launches_per_step"] for v in classes.values()), 1),
                           "classes_note": "per-launch event pass, not the timed region: classes_sum_ms - ms_per_step = event fencing (a few us per launch) minus unprofiled kernels"}
        # HBM traffic per launch of that kernel from the committed PMC passes (bench.py cannot collect PMCs itself):
        # profiles/roundN_pmc_traffic.json = rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of this command
        # (tools/refresh_profiles.sh; re-collected whenever the kernel changes — the newest round's file wins)
        try:
            cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.startswith("round") and f.endswith("_pmc_traffic.json"))
            pm = json.load(open(os.path.join(ROOT, "profiles", cands[-1])))["kernels"]
            import re
            # class label -> the kernel symbols it times (template arguments as rocprofv3 prints them; tools/summarize_profiles.py uses the same map)
            pat = {"gemm_big_kernel<0>": r"gemm_big_kernel<0,|geglu_xres_kernel<", "gemm_big_kernel<1>": r"gemm_big_kernel<1,", "conv_patch_kernel": r"conv_patch_kernel<",
                   "attn_pp40_kernel<true>": r"attn_pp40_kernel<true,0[,>]"}.get(dom, re.escape(dom.replace(" ", "")))
            hit = [v for k, v in pm.items() if re.search(pat, k.replace(" ", ""))]
            # every symbol this run launched in the class must have its counters in that file (a kernel added since the PMC passes has none:
            # quoting the file's mean would then describe another set of kernels) — else traffic stays null and the line says what is missing
            missing = [sy for sy in launched.get(dom, []) if not any(sy in k.replace(" ", "") for k in pm)]
            out["roofline"]["symbols_launched"] = launched.get(dom, [])
            if missing:
                out["roofline"]["traffic_source"] = f"profiles/{cands[-1]} has no counters for {missing}: traffic not quoted (re-run tools/refresh_profiles.sh)"
                hit = []
            if hit and world == 1:      # a class may span several symbols (the 256- and 192-row tile): launch-weighted mean
                nl = sum(v["launches_sampled"] for v in hit)
                out["roofline"]["traffic"] = int(sum(v["hbm_bytes_per_launch_corrected"] * v["launches_sampled"] for v in hit) / max(nl, 1))
                out["roofline"]["traffic_source"] = f"profiles/{cands[-1]} (rocprofv3 PMC, gfx950-corrected, bytes/launch)"
        except Exception:
            pass
        if not inv:
            Fl = shard.local
            # duplicate key sources are merged exactly (frame 0: {0,0,0} -> one read with log2(3) added; frame 1: {0,1,0} / {0,0}):
            # executed attention work / algorithmic work, stock and PnP layers, for this rank's frames
            f0 = shard.f0
            stock = sum(len({max(f - 1, 0), f, 0}) for f in range(f0, f0 + Fl)) / (3.0 * Fl)
            pnp_l = sum(len({max(f - 1, 0), 0}) for f in range(f0, f0 + Fl)) / (2.0 * Fl)
            out["roofline"]["attn_executed_over_algorithmic"] = {"stock_layers": round(stock, 4), "pnp_layers": round(pnp_l, 4),
                                                                 "note": "TFLOP/s figures use the ALGORITHMIC flops (reference key multiplicities)"}
        tot_flops = sum(v["flops"] for v in prof.values()) / nprof
        out["config"]["algorithmic_tflop_per_step_executed"] = round(tot_flops / 1e12, 2)
        out["roofline"]["whole_step_tflops"] = round(tot_flops / (ms_per_step * 1e-3) / 1e12, 1)

    if not a.no_skip_dead_branches_leg and world == 1 and emu is None and a.steps >= 50 and not inv:
        from univst_amd import engine
        sync()
        t0 = time.perf_counter()
        engine.transfer_loop(pipe, pnp_utils.latent_adain(content[50], style[50]), text3, content, style, mask if masked else None, 50,
                             skip_dead_branches=True)
        sync()
        out["config"]["extra_skip_dead_branches_frames_per_s"] = round(F_total / (time.perf_counter() - t0), 4)
        out["config"]["extra_skip_dead_branches_note"] = ("same 50-step transfer, branches 0/1 dropped once the PnP window closes (i > 25): "
                                                          "same latents up to fp32 summation order (tests: >= 60 dB), 102 instead of 150 branch-steps; NOT the headline value")

    if rank == 0:
        if not a.no_cpu_baseline and world == 1 and emu is None and a.model == "sd15":
            out["cpu_baseline"] = cpu_baseline(F_total, unet, full=a.full_cpu, single_branch=inv, quick=a.quick_cpu)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

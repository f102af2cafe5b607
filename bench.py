#!/usr/bin/env python
"""bench.py — stylized frames/sec of the UniVST SD-v1.5 three-branch denoising loop on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]            (N>1: launched by torch.distributed.run)

Workload (BASELINE.json metric): SD-v1.5 geometry UNet (random-init synthetic weights of that architecture,
fp16), 16 frames x 512x512 (latents [1,4,16,64,64]), 50 DDIM steps of the three-branch transfer loop
(content-inv | style-inv | stylised) with AdaIN-guided attention injection on steps 0..25 and the latent AdaIN
on steps 41..45, all three branches computed on every step (reference-equivalent work, 46.8 TFLOP/step).
One "step" = one DDIM step of that loop (steps i = 0..K-1 of the 50-step schedule, wrapping modulo 50);
value = F / (50 * mean step time) frames/s — with the default K = 50 that is exactly one full transfer.
Inputs are resident in HBM before the timed region.  Prints ONE JSON line (rank 0).
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
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl == RCCL; gloo stages through the host, bring-up only)")
    ap.add_argument("--emulate-rank", default=None, metavar="R/W",
                    help="diagnostic: run the work of rank R of a W-GPU job alone on this GPU with no-op collectives (kernels, pack/unpack "
                         "and host callbacks of a frame shard, no wire time); prints the usual line with parallelism 'emulated R/W'")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
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


def cpu_baseline(frames_full, unet=None):
    """oracle ('port' of the reference algorithm, fp32 PyTorch CPU ops, reference plumbing incl. the dead temporal
    ops) on this box's host cores: TWO three-branch UNet steps at F=2 of the 16 frames (one inside the PnP window, one
    outside), extrapolated linearly in F (sparse-causal attention / convs / norms are all frame-linear) and to the
    26 + 24 steps of the loop."""
    from oracle import unet_ref, synth_inputs as si
    cores = usable_cores()
    torch.set_num_threads(cores)
    cfg = unet_ref.SD15_CONFIG
    t0 = time.time()
    if unet is not None:      # the very weights the GPU run used (fp16 values, upcast), copied D2H: seconds instead of a minute
        sd = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}
    else:
        sd = unet_ref.synth_state_dict(cfg, seed=33)
    F_s = 2
    x = torch.cat([si.content_latent(40, F_s, 64, 64), si.style_latent(40, F_s, 64, 64), si.content_latent(39, F_s, 64, 64)])
    ctx = si.text_embedding(768).expand(3, -1, -1).contiguous()
    t1 = time.time()
    with torch.no_grad():
        unet_ref.unet_forward(sd, cfg, x, 781, ctx, pnp_idx=10, exact_temporal=True)     # a step inside the PnP window (i <= 25)
        t2 = time.time()
        unet_ref.unet_forward(sd, cfg, x, 381, ctx, pnp_idx=30, exact_temporal=True)     # a step outside it
    t_in, t_out = t2 - t1, time.time() - t2
    # the 50-step loop has 26 steps inside the window (i = 0..25) and 24 outside; per-step cost has no other data dependence
    loop_full = (26 * t_in + 24 * t_out) * frames_full / F_s
    return dict(value=frames_full / loop_full, unit="frames/s", cores=cores, kind="port",
                sample=f"2 three-branch UNet steps (one inside the PnP window: {t_in:.1f} s, one outside: {t_out:.1f} s; fp32, all "
                       f"temporal ops) at F={F_s} of {frames_full} frames, 64x64 latents, on {cores} threads (cgroup quota); extrapolated "
                       f"x{frames_full // F_s} in F and to 26 + 24 steps (weight copy {t1 - t0:.0f} s excluded)")


def main():
    a = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        local = min(local, torch.cuda.device_count() - 1)      # (bring-up: several ranks may share one GPU with --backend gloo)
        torch.cuda.set_device(local)
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(a.backend)
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local)

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
        shard = FrameShard(er, ew, F_total, comm=NullComm(er, ew))
    else:
        shard = FrameShard(rank, world, F_total)
    unet = synth.build_unet(device=dev, seed=33)
    pipe = Pipe(unet, DDIMScheduler())
    pipe.scheduler.set_timesteps(50)
    pnp_utils.register_spatial_attention_pnp(pipe)
    content, style, text3, _ = synth.synth_transfer_inputs(F=F_total, h=h, w=h, device=dev)
    content = [shard.slice_frames(t) for t in content]
    style = [shard.slice_frames(t) for t in style]
    shard.attach(unet)
    sharded = world > 1 or emu is not None
    lat = shard.latent_adain(content[50], style[50]) if sharded else pnp_utils.latent_adain(content[50], style[50])
    step = shard.make_step_fn(pipe, content, style, text3) if sharded else make_step_fn(pipe, content, style, text3, None)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for i in range(a.warmup):
        lat = step(i, lat)
    sync()
    t0 = time.perf_counter()
    for i in range(a.steps):
        lat = step(i, lat)
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], device=dev if a.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    assert torch.isfinite(lat.float()).all(), "non-finite latents"
    ms_per_step = dt / a.steps * 1e3
    value = F_total / (50 * ms_per_step / 1e3)

    out = {
        "metric": "stylized frames/sec, SD-v1.5 16x512x512 @50 DDIM steps", "value": round(value, 4), "unit": "frames/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f16", "data": "synthetic",
        "config": {"workload": f"sd15_unet_three_branch_pnp_transfer_{F_total}x{h * 8}x{h * 8}_50ddim", "frames": F_total,
                   "latent": [1, 4, F_total, h, h], "branches": 3, "parallelism": (f"emulated rank {emu[0]}/{emu[1]} (no wire)" if emu else "single") if world == 1 else f"frames{world}",
                   "weights": "random-init SD-v1.5 architecture (859M + 201M temporal params), fp16"},
    }

    if rank == 0 and not a.no_profile:
        # roofline leg: per-kernel-class HIP-event timing over a second pass of the same steps (events bracket every
        # launch on its stream; kept out of the timed region so the event overhead does not perturb `value`)
        _native.profile_enable(True)
    if not a.no_profile:
        nprof = min(a.steps, 50)
        l2 = lat
        for i in range(nprof):
            l2 = step(i, l2)
        sync()
    if rank == 0 and not a.no_profile:
        prof = _native.profile_collect()
        _native.profile_enable(False)
        classes = {}
        for k, v in prof.items():
            if v["launches"]:
                classes[k] = dict(ms_per_step=round(v["ms"] / nprof, 3), launches_per_step=round(v["launches"] / nprof, 1),
                                  tflops=round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 1) if v["flops"] else None,
                                  gbs=round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1))
        mfma = [k for k in prof if k.startswith(("gemm", "attn")) and prof[k]["launches"]]
        dom = max(mfma, key=lambda k: prof[k]["ms"])
        d = prof[dom]
        ach = d["flops"] / (d["ms"] * 1e-3) / 1e12
        out["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_FP16_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ach / PEAK_FP16_TFLOPS, 4), "traffic": None,
                           "avg_launch_ms": round(d["ms"] / d["launches"], 4),
                           "algorithmic_gflop_per_launch": round(d["flops"] / d["launches"] / 1e9, 2),
                           "classes": classes}
        # HBM traffic per launch of that kernel from the committed PMC passes (bench.py cannot collect PMCs itself):
        # profiles/round1_pmc_traffic.json = rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of this command
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "round1_pmc_traffic.json")))["kernels"]
            key = dom.replace(" ", "")
            hit = [v for k, v in pm.items() if key.split("<")[0] in k and key.split("<")[1].rstrip(">") in k.replace(" ", "")]
            if hit and world == 1:
                out["roofline"]["traffic"] = hit[0]["hbm_bytes_per_launch_corrected"]
                out["roofline"]["traffic_source"] = "profiles/round1_pmc_traffic.json (rocprofv3 PMC, gfx950-corrected, bytes/launch)"
        except Exception:
            pass
        tot_flops = sum(v["flops"] for v in prof.values()) / nprof
        out["config"]["algorithmic_tflop_per_step_executed"] = round(tot_flops / 1e12, 2)
        out["roofline"]["whole_step_tflops"] = round(tot_flops / (ms_per_step * 1e-3) / 1e12, 1)

    if not a.no_skip_dead_branches_leg and world == 1 and emu is None and a.steps >= 50:
        from univst_amd import engine
        sync()
        t0 = time.perf_counter()
        engine.transfer_loop(pipe, pnp_utils.latent_adain(content[50], style[50]), text3, content, style, None, 50,
                             skip_dead_branches=True)
        sync()
        out["config"]["extra_skip_dead_branches_frames_per_s"] = round(F_total / (time.perf_counter() - t0), 4)
        out["config"]["extra_skip_dead_branches_note"] = ("same 50-step transfer, branches 0/1 dropped once the PnP window closes (i > 25): "
                                                          "same latents up to fp32 summation order (tests: >= 60 dB), 102 instead of 150 branch-steps; NOT the headline value")

    if rank == 0:
        if not a.no_cpu_baseline and world == 1 and emu is None:
            out["cpu_baseline"] = cpu_baseline(F_total, unet)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

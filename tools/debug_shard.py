"""Frame shard against the unsharded forward, rank by rank (host threads play the ranks on one GPU: ThreadLoopbackComm).
  python tools/debug_shard.py --world 8 --frames 16 --latent 32 [--idx 12] [--opt ln_fold=0] ...
prints max |got - want| / max |want| per rank and frame, inside (idx 12) or outside (idx 40) the PnP window."""
import argparse
import os
import sys
import threading
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--frames", type=int, default=16)
    ap.add_argument("--latent", type=int, default=32)
    ap.add_argument("--idx", type=int, default=12)
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--no-pnp", action="store_true")
    a = ap.parse_args()
    from univst_amd import synth
    from univst_amd.backbones.video_diffusion_sd import pnp_utils
    from univst_amd.parallel import FrameShard, ThreadLoopbackComm
    opts = [(o.split("=")[0], int(o.split("=")[1])) for o in a.opt]
    g = torch.Generator().manual_seed(12)
    x = torch.randn(3, 4, a.frames, a.latent, a.latent, generator=g).half().cuda()
    ctx = torch.randn(1, 77, 768, generator=g).half().cuda().expand(3, -1, -1).contiguous()

    def build():
        unet = synth.build_unet(device="cuda", seed=7)
        pipe = types.SimpleNamespace(unet=unet)
        if not a.no_pnp:
            pnp_utils.register_spatial_attention_pnp(pipe)
            pnp_utils.register_time(pipe, a.idx)
        for k, v in opts:
            unet.set_native_option(k, v)
        return unet

    ref = build()
    want = ref(x, 741, encoder_hidden_states=ctx).sample.float()
    del ref
    shared = ThreadLoopbackComm.Shared(a.world)
    results, errors = {}, []

    def run(rank):
        try:
            with torch.cuda.stream(torch.cuda.Stream()):
                unet = build()
                sh = FrameShard(rank, a.world, a.frames, comm=ThreadLoopbackComm(shared, rank))
                sh.attach(unet, max_tokens=a.latent * a.latent)
                results[rank] = unet(sh.slice_frames(x), 741, encoder_hidden_states=ctx).sample.float()
                torch.cuda.current_stream().synchronize()
        except Exception:
            import traceback
            errors.append(traceback.format_exc())
            shared.barrier.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(a.world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=900)
    if errors:
        print(errors[0])
        sys.exit(1)
    got = torch.cat([results[r] for r in range(a.world)], dim=2)
    den = float(want.abs().max())
    fl = a.frames // a.world
    for r in range(a.world):
        e = [(got[:, :, f] - want[:, :, f]).abs().amax(dim=(1, 2, 3)) / den for f in range(r * fl, (r + 1) * fl)]
        print(f"rank {r}: " + "  ".join("[" + " ".join(f"{float(v):.1e}" for v in ee) + "]" for ee in e))
    print("max", float((got - want).abs().max() / den))


if __name__ == "__main__":
    main()

"""full-size timing of the two off-loop stages: mask propagation (16 x 64x64x640 features, 512x512 masks) and the
sliding-window smoothing kernels (16 x 512x512 frames, synthetic flows instead of RAFT)."""
import sys, os, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from univst_amd.src import mask_propagation as mp, cal_optica_flow as cf

def main():
    g = torch.Generator().manual_seed(0)
    feats = torch.randn(16, 64, 64, 640, generator=g).half().cuda()
    yy, xx = np.mgrid[0:512, 0:512]
    first = (((xx - 200) ** 2 + (yy - 256) ** 2) < 128 ** 2).astype(np.uint8) * 255
    args = mp.build_parser().parse_args([])
    torch.manual_seed(33)
    mp.propagate_masks(feats, first, args); torch.cuda.synchronize()
    torch.manual_seed(33)
    t = time.time(); masks = mp.propagate_masks(feats, first, args); torch.cuda.synchronize()
    dt = time.time() - t
    print(f"mask propagation 16 frames 64x64x640 -> 512x512: {dt*1e3:.1f} ms total ({dt/15*1e3:.1f} ms/frame), fg px last frame {int((masks[-1]>0).sum())}")
    frames = torch.randint(0, 256, (1, 3, 16, 512, 512), dtype=torch.uint8, device="cuda")
    fl = [torch.randn(512, 512, 2, device="cuda") * 2 for _ in range(4)]
    cnt = [0]
    def flow_fn(a, b):
        cnt[0] += 1
        return fl[cnt[0] % 4]
    m = (torch.rand(16, 512, 512, device="cuda") > 0.5).to(torch.uint8)
    cf.sliding_window_smooth(frames, flow_fn, m); torch.cuda.synchronize()
    t = time.time(); cf.sliding_window_smooth(frames, flow_fn, m); torch.cuda.synchronize()
    dt = time.time() - t
    print(f"sliding window 16x512x512 (58 warps, flows precomputed): {dt*1e3:.2f} ms -> {58*512*512*25/dt/1e9:.1f} GB/s algorithmic")

main()

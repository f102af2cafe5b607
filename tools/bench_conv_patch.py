"""A/B of the LDS-patch 3x3 conv (conv_patch_kernel, weights [Co][Ci/32][9][32]) against the tap-inner im2col kernel
(gemm_big_kernel<1>, weights [Co][Ci/64][9][64]) at the SD-v1.5 conv shapes of one three-branch step (48 frames), interleaved
rounds in one process (guide rule 24), random operands."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from univst_amd import _native

SHAPES = [  # (Ci, C2, Co, H)   48 images each
    (320, 0, 320, 64), (640, 320, 320, 64), (320, 320, 320, 64), (640, 0, 640, 32), (320, 0, 640, 32), (1280, 640, 640, 32),
    (640, 640, 640, 32), (640, 320, 640, 32), (1280, 0, 1280, 16), (640, 0, 1280, 16), (1280, 1280, 1280, 16), (1280, 640, 1280, 16)]


def timeit(f, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main(rounds=5, iters=5):
    tot_a = tot_b = 0.0
    for Ci, C2, Co, H in SHAPES:
        imgs = 48
        x = torch.randn(imgs, H, H, Ci, device="cuda", dtype=torch.float16)
        x2 = torch.randn(imgs, H, H, C2, device="cuda", dtype=torch.float16) if C2 else None
        w = torch.randn(Co, Ci + C2, 3, 3, device="cuda", dtype=torch.float16) * 0.02
        wti = w.reshape(Co, (Ci + C2) // 64, 64, 9).permute(0, 1, 3, 2).contiguous()
        w32 = w.reshape(Co, (Ci + C2) // 32, 32, 9).permute(0, 1, 3, 2).contiguous()
        b = torch.randn(Co, device="cuda", dtype=torch.float16)
        fa = lambda: _native.conv_nhwc_tapinner(x, wti, bias=b, x2=x2)
        fb = lambda: _native.conv3x3_patch(x, w32, bias=b, x2=x2)
        ya, yb = fa(), fb()
        err = (ya.float() - yb.float()).abs().max().item() / ya.float().abs().max().item()
        ta, tb = [], []
        for _ in range(rounds):
            ta.append(timeit(fa, iters))
            tb.append(timeit(fb, iters))
        fl = 2.0 * imgs * H * H * Co * 9 * (Ci + C2)
        a, bb = min(ta), min(tb)
        tot_a += a
        tot_b += bb
        print(f"conv3x3 {Ci}+{C2}->{Co} @{H}x{H}: im2col {a:7.3f} ms {fl / a / 1e9:7.1f} TF | patch {bb:7.3f} ms {fl / bb / 1e9:7.1f} TF | x{a / bb:.3f} | rel diff {err:.1e}")
    print(f"sum: im2col {tot_a:.3f} ms, patch {tot_b:.3f} ms, x{tot_a / tot_b:.3f}")


if __name__ == "__main__":
    main()

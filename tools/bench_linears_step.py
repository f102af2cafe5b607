"""Every large linear of one three-branch UNet step (F=16, 64x64 latents), by shape, with its launch count: where the
gemm_big_kernel<0> class spends its time, and what torch's vendor GEMM does on the same problem (comparison only)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from univst_amd import _native


def t(f, it=8):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


vendor = "--vendor" in sys.argv
tot = 0.0
totv = 0.0
for lvl, (M, C, nblk) in enumerate([(196608, 320, 5), (49152, 640, 5), (12288, 1280, 5)]):
    for tag, N, K, res, geglu, bias, cnt in [("proj_in", C, C, False, False, True, 1), ("qkv", 3 * C, C, False, False, False, 1),
                                             ("to_out+res", C, C, True, False, True, 3), ("to_q", C, C, False, False, False, 1),
                                             ("ff1 geglu", 8 * C, C, False, True, True, 1), ("ff2+res", C, 4 * C, True, False, True, 1)]:
        x = torch.randn(M, K, device="cuda", dtype=torch.float16)
        w = torch.randn(N, K, device="cuda", dtype=torch.float16) * 0.02
        b = torch.randn(N, device="cuda", dtype=torch.float16) if bias else None
        r = torch.randn(M, N, device="cuda", dtype=torch.float16) if res else None
        out = torch.empty(M, N // 2 if geglu else N, device="cuda", dtype=torch.float16)
        ms = t(lambda: _native.linear(x, w, bias=b, residual=r, geglu=geglu, out=out))
        fl = 2.0 * M * N * K
        by = 2.0 * (M * K + N * K + M * (N // 2 if geglu else N) * (2 if res else 1))
        bound = max(fl / 1200e12, by / 5e12) * 1e3          # ms: what the shape costs at 1 200 TFLOP/s (the plateau of every MFMA kernel here) or 5 TB/s, whichever binds
        line = f"L{lvl} {tag:11s} M={M:6d} N={N:5d} K={K:4d} x{cnt * nblk:2d}: {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF {by / ms / 1e6:7.1f} GB/s | bound {bound:6.3f} ms ({'flops' if fl / 1200e12 > by / 5e12 else 'bytes'}) x{ms / bound:4.2f}"
        totb = globals().get("totb", 0.0) + bound * cnt * nblk
        globals()["totb"] = totb
        tot += ms * cnt * nblk
        if vendor:
            outv = torch.empty(M, N, device="cuda", dtype=torch.float16)
            if res:
                mv = t(lambda: torch.addmm(r, x, w.t(), out=outv))
            else:
                mv = t(lambda: torch.nn.functional.linear(x, w, b))
            totv += mv * cnt * nblk
            line += f" | vendor {mv:7.3f} ms {fl / mv / 1e9:7.1f} TF"
        print(line, flush=True)
        del x, w, b, r, out
print(f"sum of max(flops / 1 200 TF, bytes / 5 TB/s) over the same launches: {globals().get('totb', 0.0):.2f} ms")
print(f"sum over a step: {tot:.2f} ms" + (f" | vendor (no geglu / bias fusion) {totv:.2f} ms" if vendor else ""))

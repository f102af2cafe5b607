"""Every linear of ONE RANK of an 8-GPU frame shard (F = 16: 2 frames x 3 branches per rank) by shape: time, TFLOP/s, GB/s of compulsory traffic —
where the 119 small-tile launches of a rank's step (gemm_kernel<*,0>, 2.8 ms at 278 TF) lose their time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from univst_amd import _native


def t(f, it=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2
tot = 0.0
for lvl, (hw, C, nblk) in enumerate([(4096, 320, 5), (1024, 640, 5), (256, 1280, 5), (64, 1280, 1)]):
    M = 3 * frames * hw
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
        print(f"L{lvl} {tag:11s} M={M:6d} N={N:5d} K={K:4d} x{cnt * nblk:2d}: {ms * 1e3:7.1f} us {fl / ms / 1e9:7.1f} TF {by / ms / 1e6:7.1f} GB/s", flush=True)
        tot += ms * cnt * nblk
print(f"sum over a rank's step: {tot:.2f} ms")

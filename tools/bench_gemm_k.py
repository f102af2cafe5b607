"""per-tile cost model of the 256x320 linear: time(K) = a + b * (K / 64) per tile at fixed M, N — the fixed part `a` (prologue latency,
epilogue, stores) against the per-k-tile part `b` (80 MFMAs per wave).  python tools/bench_gemm_k.py [geglu|plain|res]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from univst_amd import _native as nat

kind = sys.argv[1] if len(sys.argv) > 1 else "geglu"
M = 196608
N = 2560 if kind == "geglu" else 320
nat.load()
g = torch.Generator().manual_seed(0)
res = []
for K in (64, 128, 320, 640, 1280, 2560):
    x = torch.randn(M, K, generator=g).half().cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).half().cuda()
    b = torch.randn(N, generator=g).half().cuda()
    r = torch.randn(M, N, generator=g).half().cuda() if kind == "res" else None
    f = lambda: nat.linear(x, w, bias=b, residual=r, geglu=(kind == "geglu"))
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    tiles_per_cu = (M / 256) * (N / 320) / 256
    res.append((K, ms, ms * 1e3 / tiles_per_cu))
    print(f"{kind} K={K:5d}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:7.1f} TF  per tile {ms * 1e3 / tiles_per_cu:6.2f} us  ({K // 64} k tiles)")
(k0, _, t0), (k1, _, t1) = res[2], res[-1]
b_ = (t1 - t0) / ((k1 - k0) / 64)
print(f"model: {t0 - b_ * k0 / 64:.2f} us fixed + {b_:.2f} us per k tile")

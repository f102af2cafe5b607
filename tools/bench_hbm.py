import torch, sys
sys.path.insert(0, "/root/repo")
from univst_amd import _native
def t(f, it=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for mb in (126, 377, 1000):
    n = mb * 1024 * 1024 // 2
    x = torch.randn(n, device="cuda", dtype=torch.float16); y = torch.empty_like(x)
    ms = t(lambda: y.copy_(x))
    print(f"copy {mb} MB: {ms:.3f} ms  {2 * n * 2 / ms / 1e6:.0f} GB/s")
    ms = t(lambda: torch.add(x, 1.0, out=y))
    print(f"add  {mb} MB: {ms:.3f} ms  {2 * n * 2 / ms / 1e6:.0f} GB/s")
    ms = t(lambda: x.sum())
    print(f"sum  {mb} MB: {ms:.3f} ms  {n * 2 / ms / 1e6:.0f} GB/s (read only)")
x = torch.randn(196608, 320, device="cuda", dtype=torch.float16)
g = torch.ones(320, device="cuda", dtype=torch.float16); b = torch.zeros(320, device="cuda", dtype=torch.float16)
ms = t(lambda: _native.layernorm(x, g, b)); print(f"my layernorm 196608x320: {ms:.3f} ms {2 * x.numel() * 2 / ms / 1e6:.0f} GB/s")
ms = t(lambda: torch.nn.functional.layer_norm(x, (320,), g, b)); print(f"torch layernorm: {ms:.3f} ms {2 * x.numel() * 2 / ms / 1e6:.0f} GB/s")

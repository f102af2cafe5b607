"""How do the vendor GEMMs (torch -> hipBLASLt / rocBLAS) do at the UNet's linear shapes? (comparison only; not used by the product)"""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univst_amd import _native
def t(f, it=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for tag, M, N, K, res in [("L0 proj", 196608, 320, 320, True), ("L0 qkv", 196608, 960, 320, False), ("L0 ff2", 196608, 320, 1280, True),
                          ("L1 qkv", 49152, 1920, 640, False), ("L1 ff2", 49152, 640, 2560, True), ("L2 qkv", 12288, 3840, 1280, False),
                          ("L2 ff2", 12288, 1280, 5120, True), ("L0 ff1 (no geglu)", 196608, 2560, 320, False), ("L2 ff1 (no geglu)", 12288, 10240, 1280, False)]:
    x = torch.randn(M, K, device="cuda", dtype=torch.float16)
    w = torch.randn(N, K, device="cuda", dtype=torch.float16) * 0.02
    b = torch.randn(N, device="cuda", dtype=torch.float16)
    r = torch.randn(M, N, device="cuda", dtype=torch.float16) if res else None
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    mine = t(lambda: _native.linear(x, w, bias=b, residual=r, out=out))
    if res:
        ven = t(lambda: torch.addmm(r, x, w.t(), out=out))          # (bias not included: favours the vendor path)
    else:
        ven = t(lambda: torch.nn.functional.linear(x, w, b))
    fl = 2.0 * M * N * K
    print(f"{tag:20s} M={M} N={N} K={K}: mine {mine:.3f} ms {fl / mine / 1e9:7.0f} TF | vendor {ven:.3f} ms {fl / ven / 1e9:7.0f} TF")

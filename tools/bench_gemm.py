"""micro-benchmark of the GEMM / implicit-GEMM conv kernel at SD-v1.5 layer shapes (through the C ABI)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from univst_amd import _native

def timeit(f, iters=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

def conv(Ci, Co, H, imgs=48, C2=0, tag=""):
    x = torch.randn(imgs, H, H, Ci, device="cuda", dtype=torch.float16)
    x2 = torch.randn(imgs, H, H, C2, device="cuda", dtype=torch.float16) if C2 else None
    w = torch.randn(Co, 9, Ci + C2, device="cuda", dtype=torch.float16) * 0.02
    b = torch.randn(Co, device="cuda", dtype=torch.float16)
    ms = timeit(lambda: _native.conv_nhwc(x, w, bias=b, x2=x2))
    fl = 2.0 * imgs * H * H * Co * 9 * (Ci + C2)
    print(f"conv3x3 {Ci}+{C2}->{Co} @{H}x{H} {tag}: {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF")

def lin(M, N, K, geglu=False, res=False, tag=""):
    x = torch.randn(M, K, device="cuda", dtype=torch.float16)
    w = torch.randn(N, K, device="cuda", dtype=torch.float16) * 0.02
    b = torch.randn(N, device="cuda", dtype=torch.float16)
    r = torch.randn(M, N, device="cuda", dtype=torch.float16) if res else None
    out = torch.empty(M, N // 2 if geglu else N, device="cuda", dtype=torch.float16)
    ms = timeit(lambda: _native.linear(x, w, bias=b, residual=r, geglu=geglu, out=out))
    fl = 2.0 * M * N * K
    by = 2.0 * (M * K + N * K + M * (N // 2 if geglu else N) * (2 if res else 1))
    print(f"linear M={M} N={N} K={K} geglu={int(geglu)} {tag}: {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF {by / ms / 1e6:7.1f} GB/s")

if __name__ == "__main__":
    conv(320, 320, 64); conv(640, 320, 64, C2=320); conv(640, 640, 32); conv(1280, 1280, 16); conv(1280, 1280, 8); conv(1280, 1280, 8, C2=1280)
    lin(196608, 320, 320, res=True, tag="L0 proj"); lin(196608, 960, 320, tag="L0 qkv"); lin(196608, 2560, 320, geglu=True, tag="L0 ff1")
    lin(196608, 320, 1280, res=True, tag="L0 ff2"); lin(49152, 5120, 640, geglu=True, tag="L1 ff1"); lin(49152, 640, 2560, res=True, tag="L1 ff2")
    lin(12288, 10240, 1280, geglu=True, tag="L2 ff1"); lin(12288, 1280, 5120, res=True, tag="L2 ff2"); lin(3072, 1280, 1280, tag="L3")

def conv_ti(Ci, Co, H, imgs=48, C2=0, tag=""):
    x = torch.randn(imgs, H, H, Ci, device="cuda", dtype=torch.float16)
    x2 = torch.randn(imgs, H, H, C2, device="cuda", dtype=torch.float16) if C2 else None
    w = torch.randn(Co, (Ci + C2) // 64, 9, 64, device="cuda", dtype=torch.float16) * 0.02
    b = torch.randn(Co, device="cuda", dtype=torch.float16)
    ms = timeit(lambda: _native.conv_nhwc_tapinner(x, w, bias=b, x2=x2))
    fl = 2.0 * imgs * H * H * Co * 9 * (Ci + C2)
    print(f"conv3x3 TAP-INNER {Ci}+{C2}->{Co} @{H}x{H} {tag}: {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF")

def conv_patch(Ci, Co, H, imgs=48, C2=0, tag=""):
    x = torch.randn(imgs, H, H, Ci, device="cuda", dtype=torch.float16)
    x2 = torch.randn(imgs, H, H, C2, device="cuda", dtype=torch.float16) if C2 else None
    w = torch.randn(Co, (Ci + C2) // 32, 9, 32, device="cuda", dtype=torch.float16) * 0.02
    b = torch.randn(Co, device="cuda", dtype=torch.float16)
    ms = timeit(lambda: _native.conv3x3_patch(x, w, bias=b, x2=x2))
    fl = 2.0 * imgs * H * H * Co * 9 * (Ci + C2)
    print(f"conv3x3 LDS-PATCH {Ci}+{C2}->{Co} @{H}x{H} {tag}: {ms:7.3f} ms {fl / ms / 1e9:7.1f} TF")

if __name__ == "__main__":
    conv_ti(320, 320, 64); conv_ti(640, 320, 64, C2=320); conv_ti(640, 640, 32); conv_ti(1280, 1280, 16); conv_ti(1280, 1280, 8); conv_ti(1280, 1280, 8, C2=1280)

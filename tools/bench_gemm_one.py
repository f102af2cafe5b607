import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_gemm import lin, conv_ti, conv_patch
which = sys.argv[1] if len(sys.argv) > 1 else "ff1"
if which == "ff1": lin(196608, 2560, 320, geglu=True, tag="L0 ff1")
elif which == "ff1x":
    import torch
    from univst_amd import _native
    from tools.bench_gemm import timeit
    x = torch.randn(196608, 320, device="cuda", dtype=torch.float16)
    w = _native.geglu_xres_permute(torch.randn(2560, 320, device="cuda", dtype=torch.float16) * 0.02)
    b = torch.randn(2560, device="cuda", dtype=torch.float16)
    out = torch.empty(196608, 1280, device="cuda", dtype=torch.float16)
    ms = timeit(lambda: _native.linear(x, w, bias=b, geglu=2, out=out))
    print(f"linear M=196608 N=2560 K=320 geglu=2 (X-resident) L0 ff1: {ms:7.3f} ms {2.0 * 196608 * 2560 * 320 / ms / 1e9:7.1f} TF")
elif which == "proj": lin(196608, 320, 320, res=True, tag="L0 proj")
elif which == "qkv": lin(196608, 960, 320, tag="L0 qkv")
elif which == "ff2": lin(196608, 320, 1280, res=True, tag="L0 ff2")
elif which == "l2qkv": lin(12288, 3840, 1280, tag="L2 qkv")
elif which == "conv": conv_ti(320, 320, 64)
elif which == "convp": conv_patch(320, 320, 64)

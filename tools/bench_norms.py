"""GroupNorm (+SiLU) at the step's shapes against a plain copy of the same bytes: how far the three-kernel GroupNorm is from what
HBM delivers to a streaming kernel on this chip (per-kernel split: rocprofv3 --kernel-trace --stats -- python tools/bench_norms.py)."""
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
    return e0.elapsed_time(e1) / it * 1e3


for rows, C1, C2, rps, silu, tag in ((196608, 320, 0, 65536, True, "L0 resnet norm"), (196608, 320, 0, 4096, False, "L0 transformer norm"),
                                     (196608, 640, 320, 65536, True, "L0 up-block concat norm1"), (49152, 640, 0, 16384, True, "L1 resnet norm"),
                                     (49152, 1280, 640, 16384, True, "L1 up-block concat norm1"), (12288, 1280, 0, 4096, True, "L2 resnet norm")):
    x1 = torch.randn(rows, C1, device="cuda", dtype=torch.float16)
    x2 = torch.randn(rows, C2, device="cuda", dtype=torch.float16) if C2 else None
    C = C1 + C2
    g = torch.randn(C, device="cuda", dtype=torch.float16)
    us = t(lambda: _native.groupnorm_nhwc(x1, g, g, 32, 1e-5, rps, silu=silu, x2=x2))
    src = torch.randn(rows, C, device="cuda", dtype=torch.float16)
    dst = torch.empty_like(src)
    cp = t(lambda: dst.copy_(src))
    by = rows * C * 2
    print(f"{tag:28s} rows={rows:6d} C={C:4d}: groupnorm {us:7.1f} us = {3 * by / us / 1e6:6.2f} TB/s over its 3 passes ({2 * by / us / 1e6:5.2f} on read+write) | copy {cp:6.1f} us = {2 * by / cp / 1e6:5.2f} TB/s")

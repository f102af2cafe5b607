"""What the LayerNorm fold costs the 256x320 linear: plain launch vs the same launch emitting row statistics (producer) vs the
same launch applying a folded LayerNorm (consumer), at the transformer-block shapes of a step — next to the LayerNorm launch it removes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from univst_amd import _native


def t(f, it=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


for M, C in ((196608, 320), (49152, 640), (12288, 1280)):
    x = torch.randn(M, C, device="cuda", dtype=torch.float16)
    gm = torch.randn(C, device="cuda", dtype=torch.float16)
    ln = t(lambda: _native.layernorm(x, gm, gm))
    st = torch.zeros(M, C // 160, 2, device="cuda", dtype=torch.float32)
    r = torch.randn(M, C, device="cuda", dtype=torch.float16)
    print(f"M={M} C={C}: layernorm {ln:7.1f} us")
    for tag, N, res, geglu in (("to_out+res (producer)", C, True, False), ("qkv (consumer)", 3 * C, False, False), ("to_q (consumer)", C, False, False),
                               ("ff1 geglu (consumer)", 8 * C, False, True)):
        w = torch.randn(N, C, device="cuda", dtype=torch.float16) * 0.02
        b = torch.randn(N, device="cuda", dtype=torch.float16)
        ws = torch.randn(N, device="cuda", dtype=torch.float32)
        out = torch.empty(M, N // 2 if geglu else N, device="cuda", dtype=torch.float16)
        rr = r if res else None
        plain = t(lambda: _native.linear(x, w, bias=b, residual=rr, geglu=geglu, out=out))
        same = t(lambda: _native.linear_ln(x, w, bias=b, residual=rr, geglu=geglu, out=out))
        if "producer" in tag:
            fold = t(lambda: _native.linear_ln(x, w, bias=b, residual=rr, geglu=geglu, out=out, stats_out=st))
        else:
            fold = t(lambda: _native.linear_ln(x, w, residual=rr, geglu=geglu, out=out, ln=(st, ws, ws)))
        print(f"    {tag:24s} N={N:5d}: plain {plain:7.1f} us | LNF kernel, nothing folded {same:7.1f} us | folded {fold:7.1f} us ({fold - plain:+.1f})")

"""Does the 256 MB Infinity Cache (MALL) pay for running the row-local linear chain of a transformer block band by band?
Times the K = N = 320 residual linear, the GEGLU projection and FF2 of the 64x64 level on the full 196608 rows (every
operand streams from HBM: 378 MB .. 755 MB per launch) and on row bands of 1/2, 1/3, 1/4, 1/6 of the rows launched back to back
in CHAIN order (band b: FF1 -> FF2, so the 4C-wide hidden band written by FF1 is re-read by FF2 while it is still cache-resident)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from univst_amd import _native


def t(f, it=6):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


M, C = 196608, 320
dev = "cuda"
x = torch.randn(M, C, device=dev, dtype=torch.float16)
h3 = torch.randn(M, C, device=dev, dtype=torch.float16)
w_o = torch.randn(C, C, device=dev, dtype=torch.float16) * 0.05
b_o = torch.randn(C, device=dev, dtype=torch.float16)
w1 = torch.randn(8 * C, C, device=dev, dtype=torch.float16) * 0.05
b1 = torch.randn(8 * C, device=dev, dtype=torch.float16)
w2 = torch.randn(C, 4 * C, device=dev, dtype=torch.float16) * 0.03
b2 = torch.randn(C, device=dev, dtype=torch.float16)
wq = torch.randn(C, C, device=dev, dtype=torch.float16) * 0.05
flush = torch.empty(1 << 29, device=dev, dtype=torch.float16)      # 1 GB: evicts the cache between experiments


def chain(nb, full_hidden):
    """to_out2+res -> (LN as a plain copy-sized op is skipped) -> FF1 geglu -> FF2+res -> proj_out+res, band by band"""
    rows = M // nb
    o1 = torch.empty(M, C, device=dev, dtype=torch.float16)
    o2 = torch.empty(M, C, device=dev, dtype=torch.float16)
    o3 = torch.empty(M, C, device=dev, dtype=torch.float16)
    hid = torch.empty(M if full_hidden else rows, 4 * C, device=dev, dtype=torch.float16)

    def run():
        for b in range(nb):
            sl = slice(b * rows, (b + 1) * rows)
            _native.linear(x[sl], w_o, bias=b_o, residual=h3[sl], out=o1[sl])
            hb = hid[sl] if full_hidden else hid
            _native.linear(o1[sl], w1, bias=b1, geglu=True, out=hb)
            _native.linear(hb, w2, bias=b2, residual=o1[sl], out=o2[sl])
            _native.linear(o2[sl], w_o, bias=b_o, residual=x[sl], out=o3[sl])
    return run


for nb in (1, 2, 3, 4, 6, 8, 12):
    for fh in (True, False):
        if nb == 1 and not fh:
            continue
        flush.zero_()
        ms = t(chain(nb, fh))
        print(f"bands {nb:2d} ({M // nb:6d} rows) hidden buffer {'full' if fh else 'band-sized, reused'}: chain {ms:7.3f} ms", flush=True)

# single ops per band size
for nb in (1, 2, 3, 4, 6):
    rows = M // nb
    xs, rs = x[:rows], h3[:rows]
    out = torch.empty(rows, C, device=dev, dtype=torch.float16)
    hid = torch.empty(rows, 4 * C, device=dev, dtype=torch.float16)
    a = t(lambda: _native.linear(xs, w_o, bias=b_o, residual=rs, out=out)) * nb
    b = t(lambda: _native.linear(xs, w1, bias=b1, geglu=True, out=hid)) * nb
    c = t(lambda: _native.linear(hid, w2, bias=b2, residual=rs, out=out)) * nb
    d = t(lambda: _native.linear(xs, wq, out=out)) * nb
    print(f"hot single ops x{nb} bands of {rows} rows: to_out+res {a:.3f}  ff1 geglu {b:.3f}  ff2+res {c:.3f}  to_q {d:.3f} ms (sum over bands)", flush=True)

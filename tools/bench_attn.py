"""micro-benchmark of the attention kernel at the SD-v1.5 level shapes (through the C ABI)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from univst_amd import _native

def run(heads, d, N, F, nsrc, B=3, iters=5, prescaled=False):
    C = heads * d
    qkv = torch.randn(B * F, N, 3 * C, device="cuda", dtype=torch.float16)
    rows = []
    for b in range(B):
        for f in range(F):
            prev, first = b * F + max(f - 1, 0), b * F
            rows.append([prev, b * F + f, first][:nsrc] if nsrc == 3 else [prev, first])
    src = torch.tensor(rows, dtype=torch.int32, device="cuda")
    q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
    f = lambda: _native.attention(q, k, v, src, heads, ldq=3 * C, ldkv=3 * C, Nkv=N, C_=C, q_prescaled=prescaled)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 4.0 * B * F * heads * N * (nsrc * N) * d
    print(f"heads={heads} d={d} N={N} F={F} nsrc={nsrc}: {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s")

if __name__ == "__main__":
    run(8, 40, 4096, 16, 3)
    run(8, 40, 4096, 16, 2)
    run(8, 80, 1024, 16, 3)
    run(8, 160, 256, 16, 3)
    run(8, 160, 64, 16, 3)

import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from univst_amd import _native
def run(heads, d, N, BF, pre):
    C = heads * d
    q = torch.randn(BF, N, C, device="cuda", dtype=torch.float16)
    kv = torch.randn(3, 77, 2 * C, device="cuda", dtype=torch.float16)
    k, v = kv[..., :C], kv[..., C:]
    src = torch.tensor([[i // (BF // 3)] for i in range(BF)], dtype=torch.int32, device="cuda")
    f = lambda: _native.attention(q, k, v, src, heads, ldkv=2 * C, Nkv=77, C_=C, q_prescaled=pre)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    byt = 2.0 * BF * N * C * 2
    print(f"cross d={d} N={N} BF={BF} pre={pre}: {ms:.3f} ms  ({byt / ms / 1e6:.0f} GB/s q+o)")
run(8, 40, 4096, 48, True); run(8, 40, 4096, 48, False); run(8, 80, 1024, 48, False); run(8, 160, 256, 48, False); run(8, 160, 64, 48, False)

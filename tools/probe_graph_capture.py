"""Is one UNet forward capturable in a hipGraph?  After a warm-up call at the same shape (arena, index tables) the forward is a pure
stream of kernels on torch's current stream: capture it with torch.cuda.CUDAGraph (= hipGraph on ROCm), replay it on new inputs written
in place, compare with the eager call, and time replay vs eager launches (SD-v1.5 widths, F frames: python tools/probe_graph_capture.py [F])."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from univst_amd import synth

F = int(sys.argv[1]) if len(sys.argv) > 1 else 2
unet = synth.build_unet()
g = torch.Generator().manual_seed(0)
x = torch.randn(3, 4, F, 64, 64, generator=g).half().cuda()
x2 = torch.randn(3, 4, F, 64, 64, generator=g).half().cuda()
ctx = torch.randn(3, 77, 768, generator=g).half().cuda()
t = 501
y_eager = unet(x, t, encoder_hidden_states=ctx).sample.clone()       # warm-up: arena and tables exist now
y2_eager = unet(x2, t, encoder_hidden_states=ctx).sample.clone()
xs = x.clone()
graph = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    unet(xs, t, encoder_hidden_states=ctx)
    with torch.cuda.graph(graph, stream=s):
        y_g = unet(xs, t, encoder_hidden_states=ctx).sample
torch.cuda.current_stream().wait_stream(s)
graph.replay(); torch.cuda.synchronize()
print("replay == eager (same input):", torch.equal(y_g, y_eager))
xs.copy_(x2); graph.replay(); torch.cuda.synchronize()
print("replay == eager (new input written in place):", torch.equal(y_g, y2_eager))


def timeit(f, n=20):
    f(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3


print(f"F = {F}: eager {timeit(lambda: unet(xs, t, encoder_hidden_states=ctx)):.3f} ms per forward, graph replay {timeit(graph.replay):.3f} ms")

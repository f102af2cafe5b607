"""a rank's linears (frame shard of 8 at F = 16: M = 24576 / 6144 / 1536 / 384 rows) on the 128-wide kernel: time against the k-tile count —
is the k loop bound by the DMA round trip?  (round 5 probe)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_gemm import lin
for K in (320, 640, 1280, 2560, 5120):
    lin(6144, 640, K, res=True, tag=f"L1 rank M=6144 N=640 K={K}")
for K in (640, 1280, 2560, 5120):
    lin(1536, 1280, K, res=True, tag=f"L2 rank M=1536 N=1280 K={K}")
lin(1536, 3840, 1280, tag="L2 rank qkv")
lin(384, 1280, 1280, res=True, tag="L3 rank proj")
lin(6144, 1920, 640, tag="L1 rank qkv")

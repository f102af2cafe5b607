import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_attn import run
run(8, 40, 4096, 16, 3, iters=2, prescaled=True)      # prescaled q: the kernel the UNet graph uses (attn_pp40_kernel<true, 0, 1>: LDS-DMA ring)

"""the 32x32-level self-attention (head_dim 80, 1024 queries per frame, three sources) with prescaled q: under rocprofv3 --pmc with UNIVST_ATTN_CF=0 / 1
the SQ counters show what the accumulator-folded reference removes from the generic body (tools/pmc_attn_d80.sh)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_attn import run
run(8, 80, 1024, 16, 3, iters=2, prescaled=True)

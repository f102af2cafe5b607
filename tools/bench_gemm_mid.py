"""mid-size linears / convs around the big-tile threshold (A/B aid: UNIVST_GEMM_BIGMIN, UNIVST_GEMM_SPLITK)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_gemm import lin, conv_ti
lin(49152, 640, 640, res=True, tag="L1 proj/out")
lin(49152, 640, 2560, res=True, tag="L1 ff2")
lin(12288, 1280, 1280, res=True, tag="L2 proj/out")
lin(12288, 1280, 5120, res=True, tag="L2 ff2")
lin(12288, 3840, 1280, tag="L2 qkv")
lin(3072, 1280, 1280, res=True, tag="L3 proj")
lin(3072, 3840, 1280, tag="L3 qkv")
lin(3072, 1280, 5120, res=True, tag="L3 ff2")
conv_ti(1280, 1280, 8); conv_ti(1280, 1280, 8, C2=1280); conv_ti(1280, 1280, 16); conv_ti(640, 640, 32)

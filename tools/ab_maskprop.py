"""A/B of the mask-propagation SGEMMs: fp32 MFMA (default) vs the fp32 VALU kernel, and a bitwise comparison of the masks
(v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain: the two must agree bit for bit).  Run on the GPU box."""
import os, subprocess, sys, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import sys, os, time, hashlib, torch, numpy as np
sys.path.insert(0, os.getcwd())
from univst_amd.src import mask_propagation as mp
from univst_amd import synth
feats, first = synth.synth_maskprop_inputs(F=16, h=64, w=64, C=640, H=512, W=512, seed=11, device="cuda")
args = mp.build_parser().parse_args([]); args.num_frames = 16
def once():
    torch.manual_seed(33); return mp.propagate_masks(feats, first, args)
once(); torch.cuda.synchronize(); t0 = time.time(); m = once(); torch.cuda.synchronize()
print("ms per clip %.1f  sha %s" % ((time.time() - t0) * 1e3, hashlib.sha1(np.stack(m).tobytes()).hexdigest()[:16]))
'''
for v in ("1", "0", "1", "0"):
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, UNIVST_MASKPROP_MFMA=v), capture_output=True, text=True)
    print("UNIVST_MASKPROP_MFMA=" + v, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-500:])

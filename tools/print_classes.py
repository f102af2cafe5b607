"""stdin: the JSON line of bench.py -> fps, ms/step and the per-kernel-class split."""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d["value"], "frames/s", d["ms_per_step"], "ms/step")
for k, v in d.get("roofline", {}).get("classes", {}).items():
    print(f"  {k:28s} {v['ms_per_step']:8.3f} ms  {v['tflops'] or 0:7.1f} TF  {v['gbs']:7.1f} GB/s")

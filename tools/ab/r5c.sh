cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5c
timeout 900 python -m pytest tests/test_gpu_fused.py -x -q > gpurun_out/r5c/test_fused.txt 2>&1; tail -5 gpurun_out/r5c/test_fused.txt
for i in 1 2; do
UNIVST_CONV_GN=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-skip-dead-branches-leg > gpurun_out/r5c/bench_off_$i.json 2>> gpurun_out/r5c/bench.err
UNIVST_CONV_GN=1 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-skip-dead-branches-leg > gpurun_out/r5c/bench_on_$i.json 2>> gpurun_out/r5c/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5c/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); c=d["roofline"]["classes"]
        print(f, d["ms_per_step"], {k:v["ms_per_step"] for k,v in c.items() if v["ms_per_step"]>0.3})
    except Exception as e: print(f, "ERR", e)
PY
tail -3 gpurun_out/r5c/bench.err
timeout 900 python -m pytest tests/test_gpu_baseline_size.py -x -q -k "forward or transfer" > gpurun_out/r5c/test_baseline.txt 2>&1; tail -5 gpurun_out/r5c/test_baseline.txt
timeout 900 python -m pytest tests/test_gpu_unet.py -x -q > gpurun_out/r5c/test_unet.txt 2>&1; tail -5 gpurun_out/r5c/test_unet.txt

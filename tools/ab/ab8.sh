cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_unet.py -x -q -m gpu -s -k "producer_epilogue" > gpurun_out/t8.log 2>&1; grep -E "GroupNorm statistics|passed|failed|Error|error" gpurun_out/t8.log | head
timeout 1500 python -m pytest tests/test_gpu_baseline_size.py -x -q -m gpu -k "forward_inside or sd21 or trained_temporal" > gpurun_out/t8b.log 2>&1; tail -3 gpurun_out/t8b.log
for i in 1 2; do
UNIVST_GN_PRODUCER=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab8_off_$i.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab8_on_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab8_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); c=d['roofline']['classes']
    print(f, d['ms_per_step'], {k:v['ms_per_step'] for k,v in c.items() if v['ms_per_step']>0.6})
PY

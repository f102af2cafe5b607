cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_unet.py tests/test_gpu_baseline_size.py -x -q -m gpu -k "producer or forward_inside or sd21_widths or frame_sharded_full_size or f32_world8" > gpurun_out/t17.log 2>&1; tail -3 gpurun_out/t17.log
for i in 1 2; do
UNIVST_GN_PRODUCER=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab17_off_$i.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab17_on_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab17_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); c=d['roofline']['classes']
    print(f, d['ms_per_step'], {k:v['ms_per_step'] for k,v in c.items() if v['ms_per_step']>1.0})
PY

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "groupnorm" > gpurun_out/t16.log 2>&1; tail -3 gpurun_out/t16.log
timeout 900 python -m pytest tests/test_gpu_unet.py -x -q -m gpu -k "golden or frame_sharded_two or determin" > gpurun_out/t16b.log 2>&1; tail -3 gpurun_out/t16b.log
for i in 1 2; do for sm in 0 4194304 16777216; do
UNIVST_GN_SMALL=$sm python bench.py --emulate-rank 1/8 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab16_emu_${sm}_$i.json 2>/dev/null
done; done
for i in 1 2; do for sm in 0 4194304; do
UNIVST_GN_SMALL=$sm python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab16_n1_${sm}_$i.json 2>/dev/null
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab16_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); c=d['roofline']['classes']
    print(f, d['ms_per_step'], c['groupnorm'])
PY

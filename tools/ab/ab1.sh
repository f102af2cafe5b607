set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
# quick correctness first
timeout 1500 python -m pytest tests/test_gpu_ops.py tests/test_gpu_unet.py -x -q -m gpu -k "geglu or linear or transfer_loop or warp or sliding or window or pipeline_method or frame_sharded_two or layernorm_fold or smooth" > gpurun_out/t1.log 2>&1; tail -5 gpurun_out/t1.log
# A/B on this box: r3 lib, new lib bands off, new lib auto
for i in 1 2; do
UNIVST_LIB=$PWD/build/ab/libunivst_r3.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_r3_$i.json 2>/dev/null
UNIVST_CHAIN_BANDS=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_new_nobands_$i.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab_new_$i.json 2>/dev/null
done
python tools/bench_linears_step.py > gpurun_out/linears_r4_gelu.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); c=d['roofline']['classes']
        print(f, d['ms_per_step'], {k:v['ms_per_step'] for k,v in c.items() if v['ms_per_step']>0.6})
    except Exception as e: print(f, 'ERR', e)
PY

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for bm in 150 120 90 60 40 24; do
UNIVST_GEMM_BIGMIN=$bm python bench.py --emulate-rank 1/8 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab11_bm$bm.json 2>/dev/null
done
UNIVST_GEMM_BIGMIN=60 python bench.py --emulate-rank 1/8 --frames 32 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab11_f32_bm60.json 2>/dev/null
python bench.py --emulate-rank 1/8 --frames 32 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab11_f32_bm150.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab11_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); c=d['roofline']['classes']
    print(f, d['ms_per_step'], {k:(v['ms_per_step'],v['launches_per_step']) for k,v in c.items() if v['ms_per_step']>0.3})
PY

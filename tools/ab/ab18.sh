cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
# LayerNorm fold on the 128-wide kernel (frame-shard levels): UNIVST_LN_FOLD_SMALL=0 vs default
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "layernorm_fold or linear" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_gpu_baseline_size.py -x -q -m gpu -k "forward_inside" 2>&1 | tail -3
for i in 1 2; do for e in 0 1 2; do
UNIVST_LN_FOLD_SMALL=$e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --emulate-rank 1/8 > gpurun_out/ab19_emu_fold${e}_$i.json 2>/dev/null
done; done
for e in 0 1 2; do
UNIVST_LN_FOLD_SMALL=$e python bench.py --frames 32 --steps 20 --warmup 5 --no-cpu-baseline --emulate-rank 1/8 > gpurun_out/ab19_emu32_fold${e}.json 2>/dev/null
UNIVST_LN_FOLD_SMALL=$e python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab19_n1_fold${e}.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab19_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); c=d['roofline']['classes']
    print(f, d['ms_per_step'], {k:v['ms_per_step'] for k,v in c.items() if v['ms_per_step']>0.4 or 'layernorm' in k})
PY

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for bm in 150 128 120 100 150; do
UNIVST_GEMM_BIGMIN=$bm python bench.py --emulate-rank 1/8 --frames 32 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab14_f32_bm$bm.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/ab14_f32_bm$bm.json').read().strip().splitlines()[-1]); c=d['roofline']['classes']
print('F32 rank1/8 bigmin $bm', d['ms_per_step'], {k:(v['ms_per_step'],v['launches_per_step']) for k,v in c.items() if v['ms_per_step']>0.3})
PY
done
python bench.py --frames 32 --steps 10 --warmup 3 --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('F32 n1', d['ms_per_step'])"

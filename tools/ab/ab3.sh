cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "geglu or linear" > gpurun_out/t3.log 2>&1; tail -3 gpurun_out/t3.log
for e in 4 1 4 1; do echo "== EPI $e"; UNIVST_GEMM_EPI=$e python tools/bench_gemm_k.py geglu 2>/dev/null | tail -5; done > gpurun_out/abl_geglu2.log 2>&1
cat gpurun_out/abl_geglu2.log
for i in 1 2; do
UNIVST_GEMM_EPI=4 UNIVST_CHAIN_BANDS=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab3_old_$i.json 2>/dev/null
UNIVST_CHAIN_BANDS=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab3_new_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab3_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); c=d['roofline']['classes']
    print(f, d['ms_per_step'], {k:v['ms_per_step'] for k,v in c.items() if v['ms_per_step']>0.6})
PY

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "geglu or linear" > gpurun_out/t10.log 2>&1; tail -3 gpurun_out/t10.log
for e in 0 1 0 1; do echo "== PERSIST $e"; UNIVST_GEMM_PERSIST=$e python tools/bench_gemm_k.py geglu 2>/dev/null | tail -5; done > gpurun_out/persist.log 2>&1
cat gpurun_out/persist.log
for i in 1 2; do
UNIVST_GEMM_PERSIST=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab10_off_$i.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab10_on_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab10_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); c=d['roofline']['classes']
    print(f, d['ms_per_step'], {k:v['ms_per_step'] for k,v in c.items() if v['ms_per_step']>0.6})
PY

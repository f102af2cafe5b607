cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "linear or geglu or fold" > gpurun_out/t13.log 2>&1; tail -2 gpurun_out/t13.log
for e in 0 1 0 1; do echo "== KROT $e"; UNIVST_GEMM_KROT=$e python tools/bench_linears_step.py 2>/dev/null | tail -19; done > gpurun_out/krot.log 2>&1
grep -E "KROT|sum over" gpurun_out/krot.log
for i in 1 2; do
UNIVST_GEMM_KROT=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab13_off_$i.json 2>/dev/null
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab13_on_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab13_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); c=d['roofline']['classes']
    print(f, d['ms_per_step'], {k:v['ms_per_step'] for k,v in c.items() if v['ms_per_step']>1.0})
PY

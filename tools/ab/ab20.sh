cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
# X-resident GEGLU projection at K = 320: UNIVST_GEGLU_XRES=0 (256x320 tile) vs default
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "geglu or layernorm_fold" 2>&1 | tail -3
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do for e in 0 1; do
UNIVST_GEGLU_XRES=$e python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab20_n1_xres${e}_$i.json 2>/dev/null
done; done
for e in 0 1; do
UNIVST_GEGLU_XRES=$e python bench.py --steps 20 --warmup 5 --no-cpu-baseline --emulate-rank 1/8 > gpurun_out/ab20_emu_xres${e}.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab20_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); c=d['roofline']['classes']
    print(f, d['ms_per_step'], {k:(v['ms_per_step'],v.get('tflops')) for k,v in c.items() if 'gemm' in k})
PY

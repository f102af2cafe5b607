cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
# GEGLU epilogue: the two Horner chains of a block interleaved (default) vs one after the other (build/alt/libunivst_x1.so, -DUV_GEGLU_X1)
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "geglu or layernorm_fold" 2>&1 | tail -2
for v in x1 x2 x1 x2; do
  if [ $v = x1 ]; then export UNIVST_LIB=$PWD/build/alt/libunivst_x1.so; else unset UNIVST_LIB; fi
  echo "== $v"; python tools/bench_gemm_k.py geglu 2>&1 | grep -v amdgpu
done
for i in 1 2; do for v in x1 x2; do
  if [ $v = x1 ]; then export UNIVST_LIB=$PWD/build/alt/libunivst_x1.so; else unset UNIVST_LIB; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab19_${v}_$i.json 2>/dev/null
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab19_x*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); c=d['roofline']['classes']
    print(f, d['ms_per_step'], {k:v['ms_per_step'] for k,v in c.items() if v['ms_per_step']>1.0})
PY

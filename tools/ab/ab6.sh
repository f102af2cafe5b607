cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for i in 1 2; do for l in hip nt nt2; do L=$PWD/univst_amd/lib/libunivst_hip.so; [ $l != hip ] && L=$PWD/build/ab/libunivst_$l.so
UNIVST_LIB=$L UNIVST_CHAIN_BANDS=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab6_${l}_$i.json 2>/dev/null; done; done
UNIVST_LIB=$PWD/build/ab/libunivst_nt2.so python tools/bench_linears_step.py > gpurun_out/linears_nt2.log 2>&1
python tools/bench_linears_step.py > gpurun_out/linears_hip.log 2>&1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab6_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); c=d['roofline']['classes']
    print(f, d['ms_per_step'], {k:v['ms_per_step'] for k,v in c.items() if v['ms_per_step']>0.6})
PY
paste gpurun_out/linears_hip.log gpurun_out/linears_nt2.log | cut -c1-60,100-175

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1700 python -m pytest tests/test_gpu_sd3.py -x -q -m gpu > gpurun_out/t9.log 2>&1; tail -3 gpurun_out/t9.log
for i in 1 2; do
UNIVST_SD3_FUSED_QKV=0 python bench.py --workload sd3_transfer --steps 4 --warmup 1 --no-profile > gpurun_out/ab9_sep_$i.json 2>/dev/null
python bench.py --workload sd3_transfer --steps 4 --warmup 1 --no-profile > gpurun_out/ab9_fused_$i.json 2>/dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab9_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['ms_per_step'], d['value'])
PY

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-profile > /tmp/probe.json 2>/dev/null
MS=$(python -c "import json;print(json.loads(open('/tmp/probe.json').read().strip().splitlines()[-1])['ms_per_step'])")
echo "probe: $MS ms per step" | tee gpurun_out/box_probe.txt
if python -c "import sys; sys.exit(0 if float('$MS') < ${LIMIT:-55.5} else 1)"; then
  bash tools/refresh_profiles.sh > gpurun_out/refresh.log 2>&1; tail -3 gpurun_out/refresh.log
else
  echo "slow box: skipping the refresh"
fi

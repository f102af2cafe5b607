cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
t0=$(date +%s)
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/final_gpu_tests.log 2>&1; echo "pytest rc=$? in $(( $(date +%s) - t0 )) s"; tail -3 gpurun_out/final_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
LIMIT=55.0 bash tools/ab/refresh_if_fast.sh

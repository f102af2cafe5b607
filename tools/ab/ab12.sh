cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python bench.py --workload sd3_transfer --emulate-rank 1/8 --steps 4 --warmup 1 --no-profile > gpurun_out/sd3_emu_1of8.json 2>gpurun_out/sd3_emu.err; head -c 700 gpurun_out/sd3_emu_1of8.json; echo
timeout 3300 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_full_r4a.log 2>&1; tail -6 gpurun_out/gpu_full_r4a.log

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1700 python -m pytest tests/test_gpu_sd3.py -x -q -m gpu -k "sd35_medium or joint_attention or univst_processors" > gpurun_out/t7.log 2>&1; tail -4 gpurun_out/t7.log
timeout 1700 python -m pytest tests/test_gpu_unet.py tests/test_gpu_baseline_size.py -x -q -m gpu -k "warp or sliding or window or smooth or pipeline_method or geglu" > gpurun_out/t7b.log 2>&1; tail -4 gpurun_out/t7b.log
python bench.py --workload warp > gpurun_out/warp_r4.json 2>gpurun_out/warp_r4.err; cat gpurun_out/warp_r4.json | head -c 900

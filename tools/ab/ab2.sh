cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_unet.py -x -q -m gpu -k "pipeline_method" > gpurun_out/t2.log 2>&1; tail -5 gpurun_out/t2.log
for a in 0 1 2 3; do
  L=$PWD/univst_amd/lib/libunivst_hip.so; [ $a != 0 ] && L=$PWD/build/ab/libunivst_abl$a.so
  echo "== ABL $a"; UNIVST_LIB=$L python tools/bench_gemm_k.py geglu 2>/dev/null | tail -5
done > gpurun_out/abl_geglu.log 2>&1
python bench.py --emulate-rank 1/8 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/emu_f16_r4a.json 2>gpurun_out/emu_f16_r4a.err
cat gpurun_out/abl_geglu.log

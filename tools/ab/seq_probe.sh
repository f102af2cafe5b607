cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5e
python - <<'PY' > gpurun_out/r5e/conv_micro.txt 2>&1
import os, sys, subprocess
for bm in ("0", "256", "192"):
    env = dict(os.environ, UNIVST_GEMM_BM=bm)
    out = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0,'.'); from tools.bench_gemm import conv_patch; conv_patch(320,320,64); conv_patch(640,320,64,C2=320); conv_patch(320,320,64,C2=320); conv_patch(640,640,32); conv_patch(1280,640,32,C2=640)"], env=env, capture_output=True, text=True).stdout
    print("UNIVST_GEMM_BM=" + bm); print(out)
PY
cat gpurun_out/r5e/conv_micro.txt | grep -v amdgpu.ids
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r5e/trace -- python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-profile --no-skip-dead-branches-leg > gpurun_out/r5e/bench.json 2> gpurun_out/r5e/err
python tools/step_shapes.py gpurun_out/r5e/trace seq 'conv_patch_kernel<3>' 1024 11 | tee gpurun_out/r5e/seq_conv_l0.txt
python tools/step_shapes.py gpurun_out/r5e/trace seq 'conv_patch_kernel<3>' 512 10 | tee gpurun_out/r5e/seq_conv_l1.txt
python tools/step_shapes.py gpurun_out/r5e/trace seq 'attn_pp40_kernel' 6144 5 | tee gpurun_out/r5e/seq_attn.txt
find gpurun_out/r5e/trace -type f -size +512k -delete

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for st in 0 2 3 4 6 0; do echo "== STAGGER $st"; UNIVST_GEMM_STAGGER=$st python tools/bench_gemm_k.py geglu 2>/dev/null | sed -n 3,3p; UNIVST_GEMM_STAGGER=$st python tools/bench_gemm_k.py res 2>/dev/null | sed -n 3,3p;  done > gpurun_out/stagger.log 2>&1
cat gpurun_out/stagger.log

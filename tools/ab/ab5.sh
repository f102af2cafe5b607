cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for l in hip nt hip nt; do L=$PWD/univst_amd/lib/libunivst_hip.so; [ $l = nt ] && L=$PWD/build/ab/libunivst_nt.so; echo "== $l"; UNIVST_LIB=$L python tools/bench_gemm_k.py geglu 2>/dev/null | tail -5; done > gpurun_out/nt.log 2>&1
cat gpurun_out/nt.log

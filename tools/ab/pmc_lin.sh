cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for w in ff1 ff1x l2qkv ff2; do
  echo "== $w"; bash tools/pmc_gemm.sh /tmp/pmc_$w $w 2>&1 | grep -E "^SQ_|linear" | tee gpurun_out/pmc_gemm_big_${w}_sq.txt
done

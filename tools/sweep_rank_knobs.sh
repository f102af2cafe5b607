cd $GRAFT_REPO_ROOT
O=gpurun_out/sweep1; mkdir -p $O
run() { # name, frames, env...
  n=$1; fr=$2; shift 2
  env "$@" python bench.py --frames $fr --emulate-rank 1/8 --steps 40 --warmup 3 --no-cpu-baseline --no-profile 2>>$O/err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$n', 'F=$fr', d['ms_per_step'])"
}
for fr in 32 16; do
run base $fr X=1
run bigmin100 $fr UNIVST_GEMM_BIGMIN=100
run bigmin250 $fr UNIVST_GEMM_BIGMIN=250
run bigmin400 $fr UNIVST_GEMM_BIGMIN=400
run bm192 $fr UNIVST_GEMM_BM=192
run bm256 $fr UNIVST_GEMM_BM=256
run nopatch $fr UNIVST_CONV_PATCH=0
run smallm0 $fr UNIVST_GEMM_SMALLM=0
run smallm1 $fr UNIVST_GEMM_SMALLM=1
run nosplitk $fr UNIVST_GEMM_SPLITK=0
run lnfold1 $fr UNIVST_LN_FOLD=1
run lnfold0 $fr UNIVST_LN_FOLD=0
run gnprod0 $fr UNIVST_GN_PRODUCER=0
run attn2pre0 $fr UNIVST_ATTN2_PRE=0
run base2 $fr X=1
done

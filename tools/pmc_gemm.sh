#!/bin/bash
# SQ counters of one GEMM-class launch (tools/bench_gemm_one.py $2: ff1 / ff2 / qkv / l2qkv = linears of the 256x320 kernel, convp = LDS-patch kernel
# (default), conv = tap-inner im2col kernel); counters only.
cd /tmp && export TMPDIR=/tmp
OUT=${1:-$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm}
WHICH=${2:-convp}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES \
    --output-format csv -d $OUT/p1 -- python tools/bench_gemm_one.py $WHICH > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES \
    --output-format csv -d $OUT/p2 -- python tools/bench_gemm_one.py $WHICH > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_MFMA SQ_INSTS_FLAT \
    --output-format csv -d $OUT/p3 -- python tools/bench_gemm_one.py $WHICH > $OUT/p3.log 2>&1
python - $OUT <<'PY'
import csv, sys, glob, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_" in r["Kernel_Name"] or "conv_patch" in r["Kernel_Name"] or "geglu_xres" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(d):
    print(f"{k:36s} {sum(d[k]) / len(d[k]):16.0f}  (n={len(d[k])})")
PY
tail -3 $OUT/p3.log

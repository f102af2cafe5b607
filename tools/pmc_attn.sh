#!/bin/bash
# SQ counters of the 64x64-level attention kernel (two passes; counters only, no other trace domains).
# usage (on the GPU box): bash tools/pmc_attn.sh [out_dir]
cd /tmp && export TMPDIR=/tmp
OUT=${1:-$GRAFT_REPO_ROOT/gpurun_out/pmc_attn}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES \
    --output-format csv -d $OUT/p1 -- python tools/bench_attn_one.py > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES \
    --output-format csv -d $OUT/p2 -- python tools/bench_attn_one.py > $OUT/p2.log 2>&1
python - $OUT <<'PY'
import csv, sys, glob, collections
d = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn" in r["Kernel_Name"]:
            d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(d):
    print(f"{k:36s} {sum(d[k]) / len(d[k]):16.0f}  (n={len(d[k])})")
PY

#!/bin/bash
# SQ counters of the 32x32-level attention (head_dim 80) with and without the accumulator-folded reference (attn_body CF); counters only, no other trace domains.
# usage (on the GPU box): bash tools/pmc_attn_d80.sh [out_dir]
cd /tmp && export TMPDIR=/tmp
OUT=${1:-$GRAFT_REPO_ROOT/gpurun_out/pmc_attn_d80}
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for cf in 0 1; do
  UNIVST_ATTN_CF=$cf rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES \
      --output-format csv -d $OUT/cf$cf -- python tools/bench_attn_d80.py > $OUT/cf$cf.log 2>&1
done
python - $OUT <<'PY'
import csv, sys, glob, collections
for cf in (0, 1):
    d = collections.defaultdict(list)
    for f in glob.glob(sys.argv[1] + f"/cf{cf}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "attn_kernel" in r["Kernel_Name"]:
                d[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"== UNIVST_ATTN_CF={cf}")
    for k in sorted(d):
        print(f"{k:36s} {sum(d[k]) / len(d[k]):16.0f}  (n={len(d[k])})")
PY

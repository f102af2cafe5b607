#!/bin/bash
# A/B of attention kernel variants inside ONE gpurun call (boxes of the pool differ by +-4 %).
# usage: bash tools/ab_attn.sh ENVVAR "v1 v2 ..." [rounds]      e.g.  bash tools/ab_attn.sh UNIVST_ATTN_STG "0 1" 3
cd ${GRAFT_REPO_ROOT:-.}
for r in $(seq 1 ${3:-2}); do
  for v in $2; do
    echo "== $1=$v"
    env $1=$v python tools/bench_attn_one.py 2>&1 | grep -v amdgpu.ids
  done
done

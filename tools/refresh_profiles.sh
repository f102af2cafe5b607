#!/bin/bash
# Regenerates everything under profiles/ on the GPU box (one gpurun call):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh'
# then copy gpurun_out/profiles_new/* over profiles/ and commit.  Counter passes carry no other trace domain.
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/profiles_new
rm -rf $O && mkdir -p $O/raw
cd $R
python bench.py > $O/round1_bench_n1.json 2> $O/raw/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw/stats -- python bench.py --no-cpu-baseline --no-skip-dead-branches-leg > $O/round1_bench_under_rocprof.json 2> $O/raw/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/raw/fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $O/raw/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/raw/write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $O/raw/write.log 2>&1
bash tools/pmc_attn.sh $O/raw/attn > $O/round1_pmc_attn_d40_sq.txt 2>&1
bash tools/pmc_gemm.sh $O/raw/gemm 2>&1 | grep -E "^SQ_" > $O/round1_pmc_conv_tile_sq.txt
python tools/summarize_profiles.py $O
rm -rf $O/raw/stats/*/*_agent_info.csv
ls -la $O

# the two HBM counter passes alone (tools/refresh_profiles.sh lines 14-15) + their summary -> gpurun_out/pmc2/round4_pmc_traffic.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc2; rm -rf $O; mkdir -p $O/raw; cd $R
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/raw/fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $O/raw/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/raw/write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $O/raw/write.log 2>&1
ROUND=4 python tools/summarize_profiles.py $O 2>&1 | tail -3
find $O/raw -type f -size +512k -delete

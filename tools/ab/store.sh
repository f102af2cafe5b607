cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/probes/store_probe 2>&1 | tee gpurun_out/store_probe.txt

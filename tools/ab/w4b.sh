cd $GRAFT_REPO_ROOT
for v in dma0 reads0 both0; do echo "=== ablation $v (all-asm column only is meaningful; results are wrong by construction)"; tools/probes/w4_probe_abl_$v 2>&1 | sed 's/| 4 waves builtin[^|]*|/|/; s/| + register double buffer[^|]*|/|/' | cut -c1-250; done

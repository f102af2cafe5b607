cd $GRAFT_REPO_ROOT
for cfg in "6 10" "8 6" "8 8"; do set -- $cfg
  echo "=== MJ=$1 (tile $((32*$1)) x 320), accumulator columns in AGPRs: $2"
  tools/probes/w4_probe_$1_$2 2>&1 | sed 's/| 4 waves builtin[^|]*|/|/' | cut -c1-330
done

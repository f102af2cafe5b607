#!/bin/bash
# ONE parameterised A/B driver (round 5; replaces the 28 one-off tools/ab/ab*.sh of round 4, which are in the git history).  Boxes of the pool differ by
# +-4 %, so every comparison alternates the variants INSIDE one gpurun call:
#   gpurun --timeout 1800 -- 'bash tools/ab.sh VAR=a,b[,c] [REPS=2] [TESTS="-k expr" | TESTS=all] [BENCH="--steps 20 --warmup 3 ..."] [EMU=1/8] [OUT=name]'
#     VAR=a,b     environment switch and its values (e.g. UNIVST_ATTN2_FUSED=0,1); "UNIVST_LIB=path1,path2" compares two builds of the library
#     TESTS       a pytest selection run first (correctness before timing); all = the whole GPU suite
#     EMU=r/w     additionally one emulated-rank line per variant
# Prints per run: ms per step and the per-class split (classes above 0.3 ms), then the mean per variant.
set -u
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/..}
SPEC="" REPS=2 TESTS="" BENCH="--steps 20 --warmup 3" EMU="" OUT="ab"
for a in "$@"; do
  case "$a" in
    REPS=*) REPS=${a#REPS=};; TESTS=*) TESTS=${a#TESTS=};; BENCH=*) BENCH=${a#BENCH=};; EMU=*) EMU=${a#EMU=};; OUT=*) OUT=${a#OUT=};;
    *=*) SPEC=$a;;
  esac
done
[ -z "$SPEC" ] && { echo "usage: tools/ab.sh VAR=a,b [REPS=n] [TESTS=..] [BENCH=..] [EMU=r/w] [OUT=name]"; exit 2; }
VAR=${SPEC%%=*}; IFS=, read -ra VALS <<< "${SPEC#*=}"
D=gpurun_out/$OUT; mkdir -p $D
if [ -n "$TESTS" ]; then
  if [ "$TESTS" = all ]; then timeout 2400 python -m pytest tests -x -q -m gpu > $D/tests.log 2>&1; else timeout 1500 python -m pytest tests -x -q -m gpu $TESTS > $D/tests.log 2>&1; fi
  echo "pytest rc=$?"; tail -3 $D/tests.log
fi
for i in $(seq 1 $REPS); do for v in "${VALS[@]}"; do
  env $VAR=$v python bench.py $BENCH --no-cpu-baseline --no-skip-dead-branches-leg > $D/${VAR}_${v//\//_}_$i.json 2>> $D/bench.err
done; done
if [ -n "$EMU" ]; then for v in "${VALS[@]}"; do
  env $VAR=$v python bench.py $BENCH --no-cpu-baseline --emulate-rank $EMU > $D/${VAR}_${v//\//_}_emu.json 2>> $D/bench.err
done; fi
python - $D <<'PY'
import collections, glob, json, os, sys
acc = collections.defaultdict(list)
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        c = d.get("roofline", {}).get("classes", {})
        print(os.path.basename(f), d["ms_per_step"], {k: v["ms_per_step"] for k, v in c.items() if v["ms_per_step"] > 0.3})
        acc[os.path.basename(f).rsplit("_", 1)[0]].append(d["ms_per_step"])
    except Exception as e:
        print(f, "ERR", e)
for k, v in acc.items():
    print(f"{k}: mean {sum(v) / len(v):.3f} ms per step over {len(v)} run(s)")
PY

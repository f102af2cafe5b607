cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5d
for w in "" "--emulate-wire 64" ; do for f in 16 32; do
python bench.py --frames $f --emulate-rank 1/8 $w --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r5d/emu_f${f}_$(echo $w | tr -d ' -').json 2>> gpurun_out/r5d/err
done; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5d/emu_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); c=d["roofline"]["classes"]
        print(f, d["ms_per_step"], d["config"].get("modelled_wire_ms_per_step"), {k:v["ms_per_step"] for k,v in c.items() if v["ms_per_step"]>0.15})
    except Exception as e: print(f, "ERR", e)
PY
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r5d/trace -- python bench.py --emulate-rank 1/8 --steps 10 --warmup 2 --no-cpu-baseline --no-profile > gpurun_out/r5d/emu_trace.json 2>> gpurun_out/r5d/err
python tools/step_shapes.py gpurun_out/r5d/trace 13 > gpurun_out/r5d/shapes_emu.txt 2>&1; find gpurun_out/r5d/trace -type f -size +512k -delete
head -60 gpurun_out/r5d/shapes_emu.txt
tail -3 gpurun_out/r5d/err

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
# board power and clocks while the headline step runs (rocm-smi sampled twice a second)
python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-profile > /tmp/b.json 2>/dev/null &
BP=$!
sleep 14
for i in $(seq 1 12); do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (edge|junction|hotspot)" | tr '\n' ';'; echo; sleep 0.5; done | tee gpurun_out/power_samples.txt
wait $BP
python -c "import json;d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]);print('ms_per_step',d['ms_per_step'])" | tee -a gpurun_out/power_samples.txt
echo "--- idle"; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ';' | tee -a gpurun_out/power_samples.txt; echo
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" | tee -a gpurun_out/power_samples.txt

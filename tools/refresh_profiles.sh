#!/bin/bash
# Regenerates everything under profiles/ on the GPU box (one gpurun call):
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh'
# then copy gpurun_out/profiles_new/* over profiles/ and commit.  Counter passes carry no other trace domain.
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=${ROUND:-6}
O=$R/gpurun_out/profiles_new
rm -rf $O && mkdir -p $O/raw
cd $R
python bench.py > $O/round${ROUND}_bench_n1.json 2> $O/raw/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw/stats -- python bench.py --no-cpu-baseline --no-skip-dead-branches-leg > $O/round${ROUND}_bench_under_rocprof.json 2> $O/raw/stats.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/raw/fetch -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $O/raw/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/raw/write -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile > $O/raw/write.log 2>&1
bash tools/pmc_attn.sh $O/raw/attn > $O/round${ROUND}_pmc_attn_d40_sq.txt 2>&1
bash tools/pmc_gemm.sh $O/raw/gemm convp 2>&1 | grep -E "^SQ_" > $O/round${ROUND}_pmc_conv_patch_sq.txt
# the dominant class by shape: the K = 320 GEGLU projection (fixed-cost bound) and a long-K linear of the same kernel
for w in ff1 l2qkv ff2; do echo "== $w"; bash tools/pmc_gemm.sh $O/raw/gemm_$w $w 2>&1 | grep -E "^SQ_"; done > $O/round${ROUND}_pmc_gemm_big_sq_raw.txt
for w in inversion inversion_pair transfer_nomask maskprop warp; do python bench.py --workload $w --no-cpu-baseline > $O/round${ROUND}_bench_$w.json 2>> $O/raw/bench.err; done
python bench.py --frames 32 --emulate-rank 0/8 --no-cpu-baseline > $O/round${ROUND}_bench_emulated_f32_rank0of8.json 2>> $O/raw/bench.err
python bench.py --frames 32 --emulate-rank 1/8 --no-cpu-baseline > $O/round${ROUND}_bench_emulated_f32_rank1of8.json 2>> $O/raw/bench.err
python bench.py --frames 32 --emulate-rank 7/8 --no-cpu-baseline > $O/round${ROUND}_bench_emulated_f32_rank7of8.json 2>> $O/raw/bench.err
python bench.py --emulate-rank 1/8 --no-cpu-baseline > $O/round${ROUND}_bench_emulated_f16_rank1of8.json 2>> $O/raw/bench.err
# the same ranks with the wire modelled at 64 GB/s per link.  Round 6: the delay sits on the FORKED stream the multicast really runs on (unet.hip kv_post) and the
# forward's stream spins on a self-raised flag; _serial = the same packs issued on the forward's own stream (UNIVST_KV_OVERLAP=0: round 5's order), _inf = a wire that
# costs nothing (what the fork / flag machinery itself costs)
python bench.py --emulate-rank 1/8 --emulate-wire 64 --no-cpu-baseline > $O/round${ROUND}_bench_emulated_f16_rank1of8_wire64.json 2>> $O/raw/bench.err
python bench.py --frames 32 --emulate-rank 1/8 --emulate-wire 64 --no-cpu-baseline > $O/round${ROUND}_bench_emulated_f32_rank1of8_wire64.json 2>> $O/raw/bench.err
python bench.py --frames 32 --emulate-rank 4/8 --emulate-wire 64 --no-cpu-baseline --no-profile > $O/round${ROUND}_bench_emulated_f32_rank4of8_wire64.json 2>> $O/raw/bench.err
UNIVST_KV_OVERLAP=0 python bench.py --emulate-rank 1/8 --emulate-wire 64 --no-cpu-baseline --no-profile > $O/round${ROUND}_bench_emulated_f16_rank1of8_wire64_serial.json 2>> $O/raw/bench.err
UNIVST_KV_OVERLAP=0 python bench.py --frames 32 --emulate-rank 1/8 --emulate-wire 64 --no-cpu-baseline --no-profile > $O/round${ROUND}_bench_emulated_f32_rank1of8_wire64_serial.json 2>> $O/raw/bench.err
python bench.py --frames 32 --emulate-rank 1/8 --emulate-wire 10000,0 --no-cpu-baseline --no-profile > $O/round${ROUND}_bench_emulated_f32_rank1of8_wire_inf.json 2>> $O/raw/bench.err
# the emulated F = 32 rank by kernel symbol x grid size (where a rank's step goes)
rocprofv3 --kernel-trace --output-format csv -d $O/raw/trace_rank -- python bench.py --frames 32 --emulate-rank 1/8 --emulate-wire 64 --steps 10 --warmup 2 --no-cpu-baseline --no-profile > $O/raw/trace_rank.log 2>&1
python tools/step_shapes.py $O/raw/trace_rank 12 > $O/round${ROUND}_emulated_f32_rank_shapes.txt 2>&1
python bench.py --workload vae_decode > $O/round${ROUND}_bench_vae_decode.json 2>> $O/raw/bench.err
# the linears of a step by shape, each next to max(flops / 1 200 TF, bytes / 5 TB/s); the fused text cross-attention against the three launches it replaces
python tools/bench_linears_step.py > $O/round${ROUND}_linears_by_shape.txt 2>> $O/raw/bench.err
python tools/bench_attn2.py > $O/round${ROUND}_attn2_fused_vs_three_launches.txt 2>> $O/raw/bench.err
tools/probes/attn2_probe > $O/round${ROUND}_attn2_probe.txt 2>&1
# per-shape table of the step in situ (kernel symbol x grid size)
rocprofv3 --kernel-trace --output-format csv -d $O/raw/trace -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-profile --no-skip-dead-branches-leg > $O/raw/trace.log 2>&1
python tools/step_shapes.py $O/raw/trace 12 > $O/round${ROUND}_step_shapes.txt 2>&1
# is the per-rank step launch-bound?  kernel-time sum (rocprofv3) against the wall time of the same run: no gaps -> a hipGraph has nothing to remove
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw/stats_emu -- python bench.py --emulate-rank 1/8 --steps 50 --warmup 2 --no-cpu-baseline --no-profile > $O/raw/emu_rocprof.json 2>> $O/raw/bench.err
python - <<PY > $O/round${ROUND}_emulated_rank_kernel_sum.txt
import csv, glob, json
rows = list(csv.DictReader(open(glob.glob("$O/raw/stats_emu/*/*kernel_stats.csv")[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows) / 1e6
calls = sum(int(r["Calls"]) for r in rows)
b = json.loads(open("$O/raw/emu_rocprof.json").read().strip().splitlines()[-1])
steps = b["steps"] + b["warmup"] + 1          # + the sharded latent_adain / set-up forward
print(f"rank 1 of 8 emulated (F = 16, no wire), under rocprofv3: {b['ms_per_step']:.2f} ms per step by the wall clock; "
      f"kernel-time sum {tot:.1f} ms over {calls} launches in ~{steps} steps = {tot / steps:.2f} ms and {calls / steps:.0f} launches per step")
PY
python bench.py --workload sd3_transfer --emulate-rank 0/8 --steps 5 --warmup 1 --no-profile > $O/round${ROUND}_bench_sd3_emulated_rank0of8.json 2>> $O/raw/bench.err
python bench.py --frames 32 --no-cpu-baseline --no-skip-dead-branches-leg > $O/round${ROUND}_bench_f32_n1.json 2>> $O/raw/bench.err
python bench.py --model sd21 --no-cpu-baseline > $O/round${ROUND}_bench_sd21_n1.json 2>> $O/raw/bench.err
[ "${FULLCPU:-1}" = 1 ] && python bench.py --full-cpu --no-profile --no-skip-dead-branches-leg > $O/round${ROUND}_bench_fullcpu.json 2>> $O/raw/bench.err
python bench.py --workload sd3_transfer --steps 5 --warmup 1 > $O/round${ROUND}_bench_sd3_transfer.json 2>> $O/raw/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw/stats_sd3 -- python bench.py --workload sd3_transfer --steps 3 --warmup 1 --no-profile > $O/raw/sd3_rocprof.log 2>&1
cp $(ls $O/raw/stats_sd3/*/*kernel_stats.csv | head -1) $O/round${ROUND}_kernel_stats_sd3_transfer.csv
# two ranks on this ONE GPU through the library's IPC communicator: evidence that the multi-rank path executes end to end (not a scaling number)
python bench.py --gpus 2 --backend gloo --steps 10 --warmup 2 --no-cpu-baseline --no-profile > $O/round${ROUND}_bench_two_ranks_one_gpu.json 2>> $O/raw/bench.err
python bench.py --gpus 8 --backend gloo --steps 4 --warmup 1 --no-cpu-baseline --no-profile > $O/round${ROUND}_bench_eight_ranks_one_gpu.json 2>> $O/raw/bench.err
tools/probes/coissue_probe > $O/round${ROUND}_coissue_probe.txt 2>&1
tools/probes/ipc_probe 1 > $O/round${ROUND}_ipc_probe.txt 2>&1
ROUND=$ROUND python tools/summarize_profiles.py $O
# gpurun merges at most 64 MiB back: the raw rocprofv3 traces (hundreds of MB of per-dispatch CSV) stay on the box, the summaries above are what is kept
find $O/raw -type f -size +512k -delete
ls -la $O

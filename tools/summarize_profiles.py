"""Turns the raw rocprofv3 output of tools/refresh_profiles.sh into the small files committed under profiles/."""
import collections, csv, glob, json, os, shutil, sys

O = sys.argv[1]
RND = os.environ.get("ROUND", "5")

def one(pattern):
    hits = glob.glob(os.path.join(O, "raw", pattern), recursive=True)
    return hits[0] if hits else None

# 1. kernel stats summary (rocprofv3 --stats) -> round1_kernel_stats.csv
ks = one("stats/**/*kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(O, f"round{RND}_kernel_stats.csv"))

# 2. HBM traffic per launch per kernel from the two counter passes
def per_kernel(pattern, counter):
    f = one(pattern)
    acc = collections.defaultdict(lambda: [0.0, 0])
    if f:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                a = acc[r["Kernel_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    return acc

fe, wr = per_kernel("fetch/**/*counter_collection.csv", "FETCH_SIZE"), per_kernel("write/**/*counter_collection.csv", "WRITE_SIZE")
kern = {}
for k in fe:
    if not any(t in k for t in ("gemm", "geglu", "attn", "groupnorm", "gn_", "layernorm", "adain", "conv")):
        continue
    f_kb = fe[k][0] / max(fe[k][1], 1)
    w_kb = wr[k][0] / max(wr[k][1], 1) if k in wr else 0.0
    kern[k] = {"launches_sampled": fe[k][1], "FETCH_SIZE_KB_per_launch": round(f_kb, 1), "WRITE_SIZE_KB_per_launch": round(w_kb, 1),
               "hbm_bytes_per_launch_corrected": int((2 * f_kb + w_kb) * 1024)}
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes over `python bench.py --steps 3 --warmup 1 "
                   "--no-cpu-baseline --no-profile` (4 steps incl. warm-up). Units: KB. Correction per MI355X_MICROARCH.md §HBM: on gfx950 "
                   "FETCH_SIZE reports 1/2 of the bytes of wide coalesced reads -> hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; WRITE_SIZE "
                   "and narrow/strided reads are uncalibrated, so treat the absolute as an upper-bound estimate.  Averages are per launch over "
                   "ALL launches of a kernel symbol (all layer shapes).",
           "kernels": kern}, open(os.path.join(O, f"round{RND}_pmc_traffic.json"), "w"), indent=1)

# 3. agreement between bench.py's HIP-event timing and rocprofv3 for the dominant kernel
try:
    b = json.loads(open(os.path.join(O, f"round{RND}_bench_under_rocprof.json")).read().strip().splitlines()[-1])
    dom = b["roofline"]["kernel"]
    import re
    # bench.py class label -> the kernel symbols it times (template arguments as rocprofv3 prints them)
    pat = {"gemm_big_kernel<0>": r"gemm_big_kernel<0,|geglu_xres_kernel<", "gemm_big_kernel<1>": r"gemm_big_kernel<1,", "conv_patch_kernel": r"conv_patch_kernel<",
           "attn_pp40_kernel<true>": r"attn_pp40_kernel<true,0[,>]", "attn_kernel_occ3<96,5,2>": r"attn_kernel_occ3<96,5,2>"}.get(dom, re.escape(dom.replace(" ", "")))
    rows = [r for r in csv.DictReader(open(ks)) if re.search(pat, r["Name"].replace(" ", ""))]
    tot_ns = sum(float(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
    open(os.path.join(O, f"round{RND}_agreement.txt"), "w").write(
        f"dominant kernel class {dom}: bench.py HIP-event average launch {b['roofline']['avg_launch_ms']:.4f} ms vs rocprofv3 "
        f"average {tot_ns / calls / 1e6:.4f} ms ({calls} calls, {len(rows)} kernel symbol(s))\n")
    print(open(os.path.join(O, f"round{RND}_agreement.txt")).read())
except Exception as e:
    print("agreement check failed:", repr(e))

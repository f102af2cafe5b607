"""Per-SHAPE table of one step from a rocprofv3 --kernel-trace CSV: launches of a kernel symbol are grouped by (symbol, grid size), i.e. by
layer shape, and reported as launches per step, average duration and ms per step — the table DESIGN.md §4 quotes next to
max(flops / 1 200 TF, bytes / 5 TB/s).

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-profile
    python tools/step_shapes.py gpurun_out/trace 12 [filter-regex]        # 12 = steps + warmup
    python tools/step_shapes.py gpurun_out/trace seq 'conv_patch_kernel<3>' 1024   # launch-by-launch durations of one (kernel, grid) in start order (last 2 steps)
"""
import collections, csv, glob, re, sys


def seq():
    root, pat, grid = sys.argv[1], re.compile(re.escape(sys.argv[3])), int(sys.argv[4])
    rows = []
    for f in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat.search(r["Kernel_Name"]) and int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1) == grid:
                rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
    rows.sort()
    per = int(sys.argv[5]) if len(sys.argv) > 5 else 11
    tail = rows[-2 * per:]
    print(f"{len(rows)} launches; the last {len(tail)} in start order (us):")
    print(" ".join(f"{d:.0f}" for _, d in tail))


def main():
    if sys.argv[2] == "seq":
        return seq()
    root, steps = sys.argv[1], float(sys.argv[2])
    flt = re.compile(sys.argv[3]) if len(sys.argv) > 3 else None
    files = glob.glob(root + "/**/*kernel_trace.csv", recursive=True)
    if not files:
        sys.exit("no *kernel_trace.csv under " + root)
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            name = re.sub(r"\(anonymous namespace\)::|void |_ZN12_GLOBAL__N_1\d+", "", name)
            name = re.sub(r"\(.*", "", name)
            if flt and not flt.search(name):
                continue
            grid = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)
            a = acc[(name, grid * int(r.get("Grid_Size_Y", 1) or 1))]
            a[0] += 1
            a[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    rows = sorted(acc.items(), key=lambda kv: -kv[1][1])
    tot = sum(v[1] for _, v in rows)
    print(f"{'kernel':58s} {'blocks':>8s} {'n/step':>7s} {'avg us':>9s} {'ms/step':>8s}")
    for (name, grid), (n, ns) in rows:
        if ns / tot < 0.0005:
            continue
        print(f"{name[:58]:58s} {grid:8d} {n / steps:7.2f} {ns / n / 1e3:9.1f} {ns / steps / 1e6:8.3f}")
    print(f"total kernel time per step: {tot / steps / 1e6:.2f} ms")


if __name__ == "__main__":
    main()

"""per-kernel register / spill / LDS table of one .hip source (hipcc -Rpass-analysis=kernel-resource-usage): python tools/resusage.py univst_amd/csrc/attention.hip [extra flags]"""
import re, subprocess, sys
src = sys.argv[1]
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Iinclude", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/ru.o"] + sys.argv[2:], capture_output=True, text=True).stderr
KEYS = {"VGPRs": "V", "AGPRs": "A", "VGPRs Spill": "spill", "ScratchSize [bytes/lane]": "scr", "Occupancy [waves/SIMD]": "occ", "LDS Size [bytes/block]": "lds"}
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
    for k, short in KEYS.items():
        m = re.search(r"remark:\s+" + re.escape(k) + r": (\d+)", line)
        if m and cur:
            rows[cur][short] = int(m.group(1))
for k, v in rows.items():
    name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
    print(f"{name[:100]:100s} " + " ".join(f"{s}{v.get(s)}" for s in KEYS.values()))

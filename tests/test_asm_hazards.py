"""The attention kernels prefetch K/V with inline-asm global loads whose destination registers are unprotected until the
explicit `s_waitcnt vmcnt(0)` in store_tile() (attention.hip).  The compiler does not know the loads are asynchronous, so
nothing stops it from copying or spilling such a register early.  This test compiles attention.hip to assembly and checks,
kernel by kernel, that no instruction between an asm load and the next vmcnt(0) wait touches a pending destination."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_async_load_destinations_untouched_until_wait(tmp_path):
    asm = tmp_path / "attention.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", "-S", "--cuda-device-only",
                    f"{ROOT}/univst_amd/csrc/attention.hip", "-o", str(asm)], check=True, capture_output=True, timeout=900)
    pending, in_asm, kernel, nloads, problems = set(), False, None, 0, []
    for line in asm.read_text().splitlines():
        s = line.strip()
        if s.startswith("_Z") and ":" in s.split(";")[0]:
            kernel, pending = s.split(":")[0], set()
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s.startswith((";", ".")):
            continue
        body = s.split(";")[0]
        if in_asm and body.startswith("global_load_dwordx4"):
            dst = body.split(",")[0]
            pending |= regs(dst)
            nloads += 1
            continue
        if body.startswith("s_waitcnt") and "vmcnt(0)" in body:
            pending = set()
            continue
        if body.startswith("s_endpgm"):
            pending = set()
            continue
        if pending and (regs(body) & pending):
            problems.append(f"{kernel}: `{body}` touches in-flight load destination(s) {sorted(regs(body) & pending)}")
    assert nloads >= 8, "no asm prefetch loads found: the check is not looking at the right code"
    kernels = sorted({q.split(":")[0] for q in problems})
    assert not problems, f"{len(problems)} hazards in {kernels}\n" + "\n".join(problems[:10])

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


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_mixed_mfma_families_are_fenced(tmp_path):
    """hipcc (ROCm 7.2) does not pad the dependency  v_mfma_f32_16x16x16_f16 D  ->  v_mfma_f32_16x16x32_f16 SrcC = D  with enough wait
    states: issued back to back the 32-wide MFMA reads stale accumulator registers (met in attn_text_kernel in round 2, and again in
    round 3 when an interleave hint let the scheduler pair the two in attn_pp40_kernel's 16-wide second k step: wrong scores, no
    fault) — and it is the scheduler, not the source order, that decides how close the two end up (it moved MFMAs across
    sched_barrier(0) by sinking them at IR level).  The kernel therefore puts a data-flow fence between the two families: an empty
    asm that READS every 16-wide result (hipcc does pad MFMA result -> VGPR read) and redefines an operand of every 32-wide MFMA.
    This test reads the ISA: between a 16-wide MFMA and any 32-wide MFMA that accumulates on its result there must be an
    ;;#ASMSTART marker."""
    asm = tmp_path / "attention.s"
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", "-S", "--cuda-device-only",
                    f"{ROOT}/univst_amd/csrc/attention.hip", "-o", str(asm)], check=True, capture_output=True, timeout=900)
    kernel, n_asm, writers, pairs, problems = None, 0, {}, 0, []
    for line in asm.read_text().splitlines():
        s = line.strip()
        if s.startswith("_Z") and ":" in s.split(";")[0]:
            kernel, n_asm, writers = s.split(":")[0], 0, {}
            continue
        if s.startswith(";;#ASMSTART"):
            n_asm += 1
            continue
        body = s.split(";")[0]
        if not body.startswith("v_mfma_f32_16x16x"):
            continue
        ops = [o.strip() for o in body.split(None, 1)[1].split(",")]
        dst = regs(ops[0])
        if body.startswith("v_mfma_f32_16x16x16_f16"):
            for r in dst:
                writers[r] = n_asm
        elif body.startswith("v_mfma_f32_16x16x32_f16"):
            srcc = regs(ops[3]) if len(ops) > 3 else set()
            hit = [writers[r] for r in srcc if r in writers]
            if hit:
                pairs += 1
                if n_asm == max(hit):
                    problems.append(f"{kernel}: `{body}` accumulates on a 16-wide MFMA result with no fence in between")
            for r in dst:
                writers.pop(r, None)
    assert pairs >= 8, "no 16-wide -> 32-wide accumulator hand-over found: the check is not looking at the kernel it is meant for"
    assert not problems, "\n".join(problems[:10])

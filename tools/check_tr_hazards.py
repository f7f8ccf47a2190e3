"""Static check of the kernels that use inline-asm LDS transpose reads (ds_read_b64_tr_b16).

hipcc neither counts nor waits for loads issued from inline asm, so between such a read and the `s_waitcnt lgkmcnt` that covers it the
destination registers are IN FLIGHT: any instruction the compiler places there that touches them (typically a v_mov phi copy at a branch
merge) reads stale data — a timing-dependent wrong result that no small test reproduces reliably (it happened once during development:
heads 4 and 12 of one shape, 1 run in 3).  The kernels therefore keep every read .. wait interval free of control flow and name the
destinations as in/out operands of the wait; this script compiles them to ISA and verifies both properties.

Usage: python tools/check_tr_hazards.py            (needs hipcc; no GPU)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "multi-task-transformer_amd", "csrc")
FILES = ["attn_fast.hip", "attn_bwd.hip", "gemm.hip"]


def _regs(text):
    out = set()
    for m in re.finditer(r"v\[(\d+):(\d+)\]", text):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        out.add(int(m.group(1)))
    return out


def check_asm(lines):
    """-> (blocks, problems): every run of transpose reads up to the lgkmcnt(0) wait that retires it."""
    blocks, problems, i = 0, [], 0
    func = "?"
    while i < len(lines):
        if lines[i].startswith("_Z") and lines[i].rstrip().endswith(":"):
            func = lines[i].split(":")[0]
        if "ds_read_b64_tr_b16" not in lines[i]:
            i += 1
            continue
        blocks += 1
        inflight, j, done = [], i, False
        while j < len(lines):
            t = lines[j].strip()
            if "ds_read_b64_tr_b16" in t:
                m = re.search(r"v\[(\d+):(\d+)\]", t)
                inflight.append(set(range(int(m.group(1)), int(m.group(2)) + 1)))
            elif t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
                done = True
                break
            elif t.startswith("s_waitcnt") and "lgkmcnt(" in t:
                n = int(re.search(r"lgkmcnt\((\d+)\)", t).group(1))
                inflight = inflight[len(inflight) - n:] if n < len(inflight) else inflight      # LDS returns in order
            elif t.startswith(".LBB") or t.startswith(("s_cbranch", "s_branch", "s_endpgm")):
                problems.append(f"{func}: control flow inside a transpose-read .. wait interval (line {j}: {t})")
            elif t and not t.startswith((";", ".")):
                cur = set().union(*inflight) if inflight else set()
                if _regs(t) & cur:
                    problems.append(f"{func}: in-flight register touched before its wait (line {j}: {t})")
            j += 1
        if not done:
            problems.append(f"{func}: transpose reads at line {i} are never retired by an lgkmcnt(0) wait")
        i = j + 1
    return blocks, problems


def main():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    total, problems = 0, []
    with tempfile.TemporaryDirectory() as tmp:
        for f in FILES:
            out = os.path.join(tmp, f + ".s")
            subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-S",
                            "--cuda-device-only", os.path.join(CSRC, f), "-o", out], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            b, p = check_asm(open(out).read().split("\n"))
            total += b
            problems += [f"{f}: {x}" for x in p]
    print(f"{total} transpose-read blocks checked, {len(problems)} problem(s)")
    for p in problems:
        print("  " + p)
    return total, problems


if __name__ == "__main__":
    sys.exit(1 if main()[1] else 0)

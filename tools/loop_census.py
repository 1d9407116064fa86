#!/usr/bin/env python3
"""Static instruction census of the innermost MFMA loops of a HIP translation unit (no GPU needed).

    python tools/loop_census.py vllm_omni_amd/csrc/gemm.hip [--filter gemm_bf16_pp_kernel] [--flags '-DOMNI_DEV ...']
    python tools/loop_census.py --objdump loop.s              # a disassembled loop body (llvm-objdump -d), e.g. the vendor kernel's

Per loop: MFMAs, SALU, VALU (non-MFMA), LDS reads / writes, LDS-DMA pieces, other VMEM, barriers, waits, branches, and the
ratios the GEMM log quotes (scalar instructions per MFMA).  What it cannot see is time; it answers "what
does a wave have to issue per K-tile" and shows at once when a change adds address arithmetic or a spill to a hot loop."""
from __future__ import annotations

import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def classify(ins: str) -> str:
    op = ins.split()[0]
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if "lds" in ins.split(";")[0].split("//")[0] and (op.startswith("global_load_lds") or op.startswith("buffer_load")):
        return "dma"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "lds_read"
    if op.startswith("ds_"):
        return "lds_write"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op == "s_barrier":
        return "barrier"
    if op == "s_waitcnt":
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op in ("s_nop", "s_setprio", "s_sleep"):
        return "nop/prio"
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_memtime"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("v_"):
        return "valu"
    return "other"


def census(lines) -> dict[str, int]:
    out: dict[str, int] = {}
    for ln in lines:
        code = ln.split(";")[0].split("//")[0].strip()
        if not code or code.endswith(":") or code.startswith((".", "#", "<")) or re.match(r"^[0-9a-f]+ <", code):
            continue
        k = classify(code)
        out[k] = out.get(k, 0) + 1
    return out


def show(name: str, c: dict[str, int]) -> None:
    m = max(1, c.get("mfma", 0))
    keys = ("mfma", "salu", "valu", "lds_read", "lds_write", "dma", "vmem", "smem", "barrier", "wait", "branch", "nop/prio")
    print(f"{name}\n    " + "  ".join(f"{k} {c.get(k, 0)}" for k in keys)
          + f"\n    per MFMA: salu {c.get('salu', 0) / m:.2f}  salu+branch+wait+nop {(c.get('salu', 0) + c.get('branch', 0) + c.get('wait', 0) + c.get('nop/prio', 0)) / m:.2f}"
            f"  lds_read {c.get('lds_read', 0) / m:.3f}  dma {c.get('dma', 0) / m:.3f}")


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("--filter", default="")
    ap.add_argument("--objdump", action="store_true", help="src is a disassembled loop body, not a .hip file")
    ap.add_argument("--flags", default="", help="extra hipcc flags, one string (e.g. '-DOMNI_DEV -DOMNI_PP_BALANCED=1')")
    a = ap.parse_args()
    if a.objdump:
        show(os.path.basename(a.src), census(open(a.src).read().splitlines()))
        return
    from vllm_omni_amd.csrc import build as B

    extra = a.flags.split()
    if extra:
        B.FLAGS = B.FLAGS + extra
    asm = B.device_asm(a.src)
    # function bodies, to slice the loops out of
    fn, body, bodies = None, [], {}
    for line in asm:
        m = re.match(r"^([A-Za-z_][\w$.]*):", line)
        if m and not m.group(1).startswith(".L"):
            fn, body = m.group(1), []
            continue
        if fn is not None:
            if line.startswith(".Lfunc_end"):
                bodies[fn], fn = body, None
            else:
                body.append(line)
    for sym, loops in B.mfma_loops(asm).items():
        name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
        name = name.split("(omni_")[0].split("(unsigned")[0]
        if a.filter and a.filter not in name:
            continue
        for lp in loops:
            if lp["innermost"]:
                show(f"{name}  [lines {lp['start']}..{lp['end']} of the function]", census(bodies[sym][lp["start"]:lp["end"] + 1]))


if __name__ == "__main__":
    main()

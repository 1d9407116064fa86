#!/usr/bin/env python3
"""Builds libomni_cdna4.so (the C-ABI HIP library) for gfx950, in-tree, with plain hipcc.

    python vllm_omni_amd/csrc/build.py [--force]

One `hipcc -c` per .hip translation unit (run concurrently), then one `hipcc -shared` link.  hipcc
cross-compiles without a GPU, so this also runs in the GPU-less authoring container.  The built .so is
git-ignored but travels with gpurun snapshots.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["gemm.hip", "attention.hip", "attention_w64.hip", "elementwise.hip", "vae.hip", "dit_forward.hip"]
LIB = os.path.join(os.path.dirname(HERE), "libomni_cdna4.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-result"]


def _stale(target: str, deps: list[str]) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


_AGPR = re.compile(r"\ba(\d+|\[\d+(:\d+)?\])")


def agpr_violations(asm_lines) -> dict[str, list[tuple[int, str]]]:
    """{kernel: [(line, instruction), ...]} for every function of a device-assembly listing that carries the OMNI_OWNS_AGPRS
    marker and yet has compiler-generated instructions (outside #ASMSTART/#ASMEND) naming an AGPR."""
    out: dict[str, list[tuple[int, str]]] = {}
    fn, owned, inasm, hits = None, False, False, []
    for n, line in enumerate(asm_lines, 1):
        m = re.match(r"^([A-Za-z_][\w$.]*):", line)
        if m and not m.group(1).startswith(".L"):
            fn, owned, inasm, hits = m.group(1), False, False, []
            continue
        if "#ASMSTART" in line:
            inasm = True
        elif "#ASMEND" in line:
            inasm = False
        elif inasm:
            owned = owned or "omni: AGPRs owned by asm" in line
        else:
            code = line.split(";")[0].strip()
            if code and not code.startswith(".") and _AGPR.search(code):
                hits.append((n, code))
        if line.startswith(".Lfunc_end") and fn is not None:
            if owned and hits:
                out[fn] = hits
            fn, hits = None, []
    return out


def check_agpr_ownership(src: str, verbose: bool = True) -> None:
    """A kernel that keeps its accumulators in literal `a[...]` registers inside asm statements (marker: OMNI_OWNS_AGPRS, the
    all-AGPR clobber statement of common.h) is only correct while hipcc itself never allocates an AGPR in it: under register
    pressure the allocator parks spilled VGPRs and constants there, silently overwriting the kernel's data (seen in round 3:
    one O^T register of the attention kernel corrupted only when a late rescale branch was taken).  There is no way to
    reserve them, so the build CHECKS: the device assembly of such a kernel must not name an AGPR outside the kernel's own
    asm statements.  Raises RuntimeError otherwise."""
    if "OMNI_OWNS_AGPRS" not in open(src).read():
        return
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run([HIPCC, *FLAGS, "--cuda-device-only", "-S", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
        with open(out) as f:
            bad = agpr_violations(f)
    for fn, hits in bad.items():
        raise RuntimeError(f"{os.path.basename(src)}: hipcc allocated AGPRs in {fn}, which owns a[0:255] through asm "
                           f"({len(hits)} instructions, first: line {hits[0][0]}: {hits[0][1]}) - lower the VGPR pressure")
    if verbose:
        print(f"agpr ownership check: {os.path.basename(src)} ok", flush=True)


def build(force: bool = False, verbose: bool = True) -> str:
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(HERE, "common.h"), os.path.join(HERE, "..", "..", "include", "omni_cdna4.h")]
    srcs = [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    objs = [os.path.join(objdir, os.path.basename(s)[:-4] + ".o") for s in srcs]

    def compile_one(src, obj):
        if not force and not _stale(obj, [src] + headers):
            return
        cmd = [HIPCC, *FLAGS, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        check_agpr_ownership(src, verbose)

    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        list(ex.map(compile_one, srcs, objs))
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))

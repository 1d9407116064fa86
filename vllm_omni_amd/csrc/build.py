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
import subprocess
import sys

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

import os
import sys

import pytest

# The fp32 ORACLE side of the VAE parity tests runs torch convolutions (MIOpen).  On a fresh box MIOpen's default exhaustive
# "find" benchmarks every solver for every new conv shape: 167 s for the 2048^2 decode oracle alone, 218 s over the three VAE image
# tests (profiles/r06i_pytest_gpu_full.log) — a third of the GPU suite spent tuning the CHECKER.  FAST find picks a solver from the
# heuristics instead (the same three tests: 7 s, gpurun_out/r06k_vae_tests_fast_find.log); results stay fp32 convolutions.  The
# product path never calls MIOpen.  A caller's own setting wins.
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)

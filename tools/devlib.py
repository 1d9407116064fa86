"""Dev-only: point the ctypes binding at a variant library for same-box A/B runs (tools/build_variants.sh).

The product loader (vllm_omni_amd/_native.py) reads no environment variable; a dev script imports this module FIRST, and it
assigns `_native.LIB_PATH` from OMNI_DEV_LIB before the library is loaded for the first time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from vllm_omni_amd import _native  # noqa: E402

if os.environ.get("OMNI_DEV_LIB"):
    assert _native._lib is None, "tools.devlib must be imported before the first native call"
    _native.LIB_PATH = os.path.abspath(os.environ["OMNI_DEV_LIB"])

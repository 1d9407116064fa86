"""Shared helpers for the test-suite (oracle access, golden loading, error metrics)."""
import json
import os

import numpy as np
import torch

import qwen_image_oracle as O  # oracle/ is on sys.path via conftest

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def cosine(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().float().cpu().flatten(), b.detach().float().cpu().flatten()
    return float(torch.dot(a, b) / (a.norm() * b.norm()))


def load_golden(name: str):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    return z, meta, meta["case"]


def load_golden_raw(name: str):
    """Fixtures whose meta has no 'case' entry (pipe_helpers)."""
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    return z, meta, meta.get("case")


def golden_params(case, dtype=torch.float32):
    P = O.make_dit_params(case["layers"], seed=1234, bias_std=case["bias_std"], norm_jitter=case["jitter"],
                          num_heads=case["heads"], joint_dim=case["joint"])
    return {k: v.to(dtype) for k, v in P.items()}


def bf16_round(t: torch.Tensor) -> torch.Tensor:
    return t.to(torch.bfloat16).to(torch.float32)

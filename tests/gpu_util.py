"""Helpers for the -m gpu parity tests (HIP path vs the CPU oracle)."""
import numpy as np
import torch


def rel_err(a, b):
    """max |a-b| / max|b|  (scale-relative, the form the 1e-4 fp32 tolerance is stated in)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def to_bf16_f32(t):
    return t.to(torch.bfloat16).to(torch.float32)

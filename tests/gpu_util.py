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


def close_fraction(a, b, rtol):
    """Element-relative reading of a tolerance: the share of elements with |a - b| <= rtol * (|b| + max|b|), i.e.
    torch.allclose(rtol=rtol, atol=rtol * max|ref|) per element -- small-magnitude outputs (weakly expressed genes)
    are held to the same absolute floor as the largest ones are held to relatively.  (`rel_err` is the max-norm
    form max|a-b| / max|b|.)"""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    ok = np.abs(a - b) <= rtol * (np.abs(b) + np.abs(b).max())
    return float(ok.mean())


def assert_allclose_rel(a, b, rtol, what=""):
    frac = close_fraction(a, b, rtol)
    assert frac == 1.0, f"{what}: only {frac:.6f} of the elements within rtol={rtol} (allclose form, atol = rtol * max|ref|)"

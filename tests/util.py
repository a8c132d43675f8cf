"""Shared helpers for the parity tests."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name))


def sd_from_npz(d, prefix="sd."):
    out = {}
    for k in d.files:
        if k.startswith(prefix):
            v = torch.from_numpy(d[k])
            out[k[len(prefix):]] = v.float() if v.dtype.is_floating_point else v
    return out


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """||a-b|| / ||b|| in fp64."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_err(a: torch.Tensor, b: torch.Tensor) -> float:
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())


def report(name, got, ref):
    r, m = rel_err(got, ref), max_err(got, ref)
    print(f"[parity] {name}: rel={r:.3e} max={m:.3e} ref_absmax={float(ref.detach().abs().max()):.3e}")
    return r, m

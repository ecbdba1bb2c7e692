"""Helpers to read the committed golden fixtures (tests/golden/*.npz)."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with np.load(os.path.join(GOLDEN_DIR, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def sub(d, prefix):
    """entries of d under 'prefix.' with the prefix stripped"""
    p = prefix + "."
    return {k[len(p):]: v for k, v in d.items() if k.startswith(p)}


def tensors(d, prefix, dtype=torch.float32):
    """'prefix.name' arrays -> {name: tensor}"""
    return {k: torch.from_numpy(np.asarray(v)).to(dtype) if np.asarray(v).dtype.kind == "f" else torch.from_numpy(np.asarray(v))
            for k, v in sub(d, prefix).items()}


def rel_err(a, b):
    a = torch.as_tensor(a, dtype=torch.float64).detach()
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_err(a, b):
    a = torch.as_tensor(a, dtype=torch.float64).detach()
    b = torch.as_tensor(b, dtype=torch.float64)
    return float((a - b).abs().max())

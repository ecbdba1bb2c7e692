"""deterministic tensor contents shared by tools/gen_golden_growth.py (reference side) and tests/test_growth.py"""
import zlib

import torch


def det_values(name, shape, tag=0):
    n = 1
    for d in shape:
        n *= d
    phase = (zlib.crc32(name.encode()) % 1000) / 7.0 + 1.7 * tag
    i = torch.arange(n, dtype=torch.float64)
    return (torch.sin(0.37 * (i + 1.0) + phase) * 0.5 + 0.01 * tag).to(torch.float32).reshape(shape)


def fill_state_dict(sd, tag):
    return {k: (det_values(k, tuple(v.shape), tag) if v.dtype.is_floating_point else v.clone()) for k, v in sd.items()}


def fingerprint(t, k=12):
    f = t.detach().to(torch.float64).reshape(-1)
    idx = (torch.arange(k, dtype=torch.int64) * 2654435761 % max(f.numel(), 1)) if f.numel() else torch.zeros(0, dtype=torch.int64)
    return [float(f.sum()), float(f.abs().sum())] + [float(v) for v in f[idx]]

"""Weight initialisers of the reference (gcn/inits.py:5-27) as GPU tensors.

TensorFlow's random stream cannot be reproduced without TensorFlow; values come from a
torch generator seeded by ``graph.set_random_seed`` and the variable's name, so a model is
reproducible run to run for a given seed."""
import zlib

import numpy as np
import torch

from . import graph


def _generator(name, device):
    g = torch.Generator(device=device)
    g.manual_seed((graph.get_random_seed() * 1000003 + zlib.crc32((name or '').encode())) % (2 ** 63))
    return g


def uniform(shape, scale=0.05, name=None, device='cuda'):
    """U(-scale, scale) (gcn/inits.py:5-8)."""
    g = _generator(name, device)
    t = (torch.rand(tuple(shape), generator=g, device=device, dtype=torch.float32) * 2 - 1) * scale
    return t.requires_grad_(True)


def glorot(shape, name=None, device='cuda'):
    """Glorot & Bengio: U(-r, r), r = sqrt(6 / (fan_in + fan_out)) (gcn/inits.py:11-15)."""
    init_range = float(np.sqrt(6.0 / (shape[0] + shape[1])))
    return uniform(shape, scale=init_range, name=name, device=device)


def zeros(shape, name=None, device='cuda'):
    return torch.zeros(tuple(shape), dtype=torch.float32, device=device).requires_grad_(True)


def ones(shape, name=None, device='cuda'):
    return torch.ones(tuple(shape), dtype=torch.float32, device=device).requires_grad_(True)

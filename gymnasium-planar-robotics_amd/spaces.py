"""Minimal stand-ins for gymnasium.spaces.Box / Dict (gymnasium is not a dependency of the engine). They carry what the
reference's spaces carry (PL:242-259): low, high, shape, dtype -- plus contains() and sample() on NumPy arrays."""
from __future__ import annotations

import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float64):
        if shape is None:
            shape = np.shape(low)
        self.shape = tuple(int(s) for s in shape)
        self.dtype = np.dtype(dtype)
        self.low = np.broadcast_to(np.asarray(low, dtype=self.dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=self.dtype), self.shape).copy()

    def contains(self, x) -> bool:
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))

    def sample(self, rng: np.random.Generator | None = None):
        rng = rng or np.random.default_rng()
        lo = np.where(np.isfinite(self.low), self.low, -1.0)
        hi = np.where(np.isfinite(self.high), self.high, 1.0)
        return rng.uniform(lo, hi).astype(self.dtype)

    def __repr__(self):
        return f'Box(shape={self.shape}, dtype={self.dtype})'


class Dict(dict):
    @property
    def spaces(self):
        return self

    def __repr__(self):
        return 'Dict(' + ', '.join(f'{k}: {v!r}' for k, v in self.items()) + ')'


def batch_box(space: Box, n: int) -> Box:
    return Box(np.broadcast_to(space.low, (n,) + space.shape), np.broadcast_to(space.high, (n,) + space.shape), dtype=space.dtype)

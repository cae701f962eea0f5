"""Observation / action spaces and the optional Gymnasium / PettingZoo binding.

Gymnasium is NOT a dependency of the engine (the build image has none). When it is importable the package binds to it:
``Box`` / ``Dict`` below ARE ``gymnasium.spaces.Box`` / ``Dict``, the env classes derive from ``gymnasium.vector.VectorEnv``
(``VectorEnvBase``), the parallel env from ``pettingzoo.ParallelEnv`` (``ParallelEnvBase``), and ``registration.py`` registers the
reference's ids (gymnasium_planar_robotics/__init__.py:21-41). Without it, minimal stand-ins carry what the reference's spaces
carry (PL:242-259): low, high, shape, dtype -- plus contains() and sample() on NumPy arrays.
"""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - exercised by tests/test_host_cpu.py with a stand-in package on sys.path
    import gymnasium as _gym
    from gymnasium.spaces import Box, Dict
    from gymnasium.vector import VectorEnv as VectorEnvBase
    HAVE_GYMNASIUM = True
except ImportError:
    _gym = None
    HAVE_GYMNASIUM = False
    VectorEnvBase = object

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

try:  # pragma: no cover
    from pettingzoo import ParallelEnv as ParallelEnvBase
    HAVE_PETTINGZOO = True
except ImportError:
    HAVE_PETTINGZOO = False
    ParallelEnvBase = object


def batch_box(space, n: int):
    """the space of ``n`` stacked samples of a Box (gymnasium.vector.utils.batch_space for a Box)"""
    return Box(np.broadcast_to(space.low, (n,) + tuple(space.shape)).copy(), np.broadcast_to(space.high, (n,) + tuple(space.shape)).copy(),
               dtype=space.dtype)


def autoreset_metadata(mode: str):
    """``metadata['autoreset_mode']`` as Gymnasium's VectorEnv wants it (``gymnasium.vector.AutoresetMode``), else the plain string"""
    if HAVE_GYMNASIUM:
        try:
            from gymnasium.vector import AutoresetMode
            return {'same_step': AutoresetMode.SAME_STEP, 'next_step': AutoresetMode.NEXT_STEP, 'disabled': AutoresetMode.DISABLED}[mode]
        except (ImportError, AttributeError):
            pass
    return mode

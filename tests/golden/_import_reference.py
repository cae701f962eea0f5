"""Import helper used ONLY by tests/golden/make_golden.py (never by the product, never at test time).

The reference (ubi-coro/gymnasium-planar-robotics, mounted read-only at /root/reference in the build
container) is pure Python + NumPy, but its modules `import mujoco`, `gymnasium` and `pettingzoo` at the
top of the file. None of the three is installed here (no network). The functions the golden vectors come
from -- qpos_is_valid, check_mover_collision, geometry_2D_utils.*, ensure_max_dyn_val, compute_reward,
compute_terminated, the ctor-time tables -- are plain NumPy and never call into MuJoCo. So this helper puts
*empty placeholder modules* into sys.modules that satisfy the import statements and the env constructors
(the MuJoCo model object is a dummy that only remembers its XML string). No MuJoCo arithmetic is emulated:
nothing that depends on mj_step results is ever used as a golden value.

The reference never travels to the GPU box; only the data files written by make_golden.py do.
"""
from __future__ import annotations

import re
import sys
import types

import numpy as np

REFERENCE_ROOT = '/root/reference'


def _module(name: str) -> types.ModuleType:
    mod = types.ModuleType(name)
    sys.modules[name] = mod
    return mod


class _Opt:
    timestep = 0.001  # <option timestep="0.001"> in the reference's MJCF (basic_envs.py:1132)


class _DummyModel:
    def __init__(self, xml: str) -> None:
        self.xml = xml
        self.opt = _Opt()

    @classmethod
    def from_xml_string(cls, xml: str) -> '_DummyModel':
        return cls(xml)

    # name -> id lookups made by MoverImpedanceController.__init__ (utils/impedance_control.py:37-46) when BenchmarkPushingEnv is
    # constructed: plain indices. The mass only enters the controller's damping matrix, which no fixture evaluates
    # (its update() needs mj_jacBody and is never called here).
    def joint(self, name):
        return types.SimpleNamespace(bodyid=[0])

    def body(self, idx):
        return types.SimpleNamespace(dofadr=[0], dofnum=[6], mass=[1.0])


class _DummyData:
    def __init__(self, model: _DummyModel) -> None:
        self.model = model


class _Box:
    def __init__(self, low, high, shape=None, dtype=np.float64):
        if shape is None:
            shape = np.shape(low)
        self.shape = tuple(shape)
        self.low = np.broadcast_to(np.asarray(low, dtype=dtype), self.shape).copy()
        self.high = np.broadcast_to(np.asarray(high, dtype=dtype), self.shape).copy()
        self.dtype = np.dtype(dtype)

    def contains(self, x) -> bool:
        x = np.asarray(x)
        return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


class _Dict(dict):
    def __init__(self, spaces):
        super().__init__(spaces)
        self.spaces = spaces


class _Env:
    """gymnasium.Env placeholder: reset(seed) only (re)creates np_random, like Gymnasium's Env.reset."""

    _np_random = None

    @property
    def np_random(self):
        if self._np_random is None:
            self._np_random = np.random.default_rng()
        return self._np_random

    @np_random.setter
    def np_random(self, value):
        self._np_random = value

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            self._np_random = np.random.default_rng(seed)


def install_placeholders() -> None:
    if 'mujoco' in sys.modules and getattr(sys.modules['mujoco'], '_gpr_placeholder', False):
        return
    mj = _module('mujoco')
    mj._gpr_placeholder = True
    mj.MjModel = _DummyModel
    mj.MjData = _DummyData
    mj.mj_step = lambda *a, **k: None
    mj.mj_forward = lambda *a, **k: None
    mj.mjtObj = types.SimpleNamespace()
    mj.mjtJoint = types.SimpleNamespace()
    mj.mjtGeom = types.SimpleNamespace()
    viewer = _module('mujoco.viewer')
    mj.viewer = viewer

    gym = _module('gymnasium')
    gym.Env = _Env
    gym.logger = types.SimpleNamespace(warn=lambda *a, **k: None)
    sys.modules['gymnasium.logger'] = gym.logger
    spaces = _module('gymnasium.spaces')
    spaces.Box = _Box
    spaces.Dict = _Dict
    gym.spaces = spaces
    envs = _module('gymnasium.envs')
    reg = _module('gymnasium.envs.registration')
    reg.register = lambda *a, **k: None
    envs.registration = reg
    gmj = _module('gymnasium.envs.mujoco')
    rend = _module('gymnasium.envs.mujoco.mujoco_rendering')
    for cls_name in ('MujocoRenderer', 'BaseRender', 'OffScreenViewer', 'WindowViewer'):
        setattr(rend, cls_name, type(cls_name, (), {}))
    gmj.mujoco_rendering = rend
    envs.mujoco = gmj
    gym.envs = envs

    pz = _module('pettingzoo')
    pz.ParallelEnv = type('ParallelEnv', (), {})


def import_reference():
    """Return the reference package with placeholders installed and name lookup patched to parse the XML."""
    sys.dont_write_bytecode = True  # /root/reference is read-only; never drop __pycache__ there
    install_placeholders()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import gymnasium_planar_robotics  # noqa: F401
    from gymnasium_planar_robotics.utils import mujoco_utils

    def names_from_xml(model, obj_type: str, name_pattern: str = ''):
        tag = {'body': 'body', 'joint': 'joint', 'site': 'site', 'actuator': 'general', 'geom': 'geom'}[obj_type]
        names = re.findall(r'<%s\b[^>]*?\bname="([^"]+)"' % tag, model.xml)
        return [n for n in names if name_pattern in n]

    mujoco_utils.get_mujoco_type_names = names_from_xml
    return gymnasium_planar_robotics

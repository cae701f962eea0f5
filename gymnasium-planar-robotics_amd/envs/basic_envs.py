"""Host-side mirror of the part of ``BasicPlanarRoboticsEnv`` that the hot path needs
(reference: gymnasium_planar_robotics/envs/basic_envs.py, cited as BE:line).

What lives here: tile / mover / collision-parameter parsing with the reference's defaults and validation rules
(BE:194-264, BE:1469-1604), the ctor-time tile tables (BE:1292-1339) and the batched collision operators
``qpos_is_valid`` / ``check_mover_collision`` / ``check_wall_collision`` (BE:355-788) -- executed by the HIP kernels of
libgpr_hip.so on device tensors. What does not: MuJoCo XML generation, meshes, rendering (no MuJoCo model exists in this
engine; see DESIGN.md, "out of scope").
"""
from __future__ import annotations

import ctypes as C
import warnings
from typing import Any

import numpy as np
import torch

from .. import _native, spaces


class VecBasicPlanarRoboticsEnv(spaces.VectorEnvBase):
    """Batched counterpart of ``BasicPlanarRoboticsEnv`` (BE:21-1626): ``num_envs`` independent copies of one tile /
    mover configuration living on one MI355X.

    :param layout_tiles: (num_tiles_x, num_tiles_y) array of 0/1, x = axis 0 (BE:36-38)
    :param num_movers: movers per env
    :param num_envs: envs owned by this object (this rank's shard)
    :param tile_params / mover_params / collision_params / std_noise / initial_mover_zpos: as in the reference (BE:39-133)
    :param device: a CUDA (ROCm) device; the engine has no CPU path
    """

    metadata = {'render_modes': [], 'autoreset_mode': 'same_step'}
    render_mode = None
    spec = None
    closed = False

    def __init__(
        self,
        layout_tiles: np.ndarray,
        num_movers: int,
        num_envs: int = 1,
        tile_params: dict[str, Any] | None = None,
        mover_params: dict[str, Any] | None = None,
        initial_mover_zpos: float = 0.005,
        std_noise: np.ndarray | float = 1e-5,
        render_mode: str | None = None,
        collision_params: dict[str, Any] | None = None,
        device: str | torch.device = 'cuda',
        _host_only: bool = False,
    ) -> None:
        if render_mode is not None:
            raise ValueError('rendering is out of scope of the GPU engine: use render_mode=None')
        # noise (BE:184-192)
        if isinstance(std_noise, float):
            self.std_noise = np.array([std_noise, std_noise, std_noise])
        else:
            assert isinstance(std_noise, np.ndarray) and std_noise.shape == (3,), (
                'noise standard deviation has to be a float or a numpy array of shape (3,)'
            )
            self.std_noise = std_noise.astype(np.float64)

        # tiles (BE:194-204, 1469-1485)
        layout_tiles = np.asarray(layout_tiles)
        assert layout_tiles.ndim == 2, 'Unexpected tile layout shape. Expected: (num_tiles_x,num_tiles_y)'
        assert np.bitwise_or(layout_tiles == 0, layout_tiles == 1).all(), 'Use a numpy array of only 0 and 1 to specify the tile layout.'
        self.layout_tiles = np.ascontiguousarray(layout_tiles.astype(np.int8))
        self.num_tiles = int(self.layout_tiles.sum())
        assert self.num_tiles > 0, 'Number of tiles must be >0.'
        self.num_tiles_x, self.num_tiles_y = self.layout_tiles.shape
        tile_params = tile_params or {}
        self.tile_size = np.asarray(tile_params.get('size', np.array([0.24 / 2, 0.24 / 2, 0.0352 / 2])), dtype=np.float64)
        self.tile_mass = tile_params.get('mass', 5.6)
        assert self.tile_size.shape == (3,), 'Specify the size of a tile using a numpy array of shape (3,)'
        assert (self.tile_size > 0).all(), 'Tile size must be >0.'
        assert self.tile_mass > 0, 'Tile mass must be >0.'
        self.x_pos_tiles, self.y_pos_tiles = self.get_tile_xy_pos()
        self.layout_tiles_wc = np.pad(self.layout_tiles, ((0, 1), (0, 1)), mode='constant', constant_values=0)

        # movers (BE:223-255, 1487-1546). Only box movers: meshes need MuJoCo's mesh compiler.
        self.num_movers = int(num_movers)
        assert self.num_movers > 0, 'Number of movers must be >0.'
        assert self.num_movers <= _native.MAX_MOVERS, f'the engine supports at most {_native.MAX_MOVERS} movers per env'
        mover_params = mover_params or {}
        self.mover_mass = mover_params.get('mass', 1.24)
        self.mover_shape = mover_params.get('shape', 'box')
        if self.mover_shape not in ('box', 'cylinder'):
            raise ValueError("mover shape must be 'box' or 'cylinder' in the GPU engine (mesh movers need MuJoCo)")
        size = np.asarray(mover_params.get('size', np.array([0.155 / 2, 0.155 / 2, 0.012 / 2])), dtype=np.float64)
        assert size.shape == (3,) or size.shape == (self.num_movers, 3), (
            'Unexpected mover size. Use a numpy array of shape (3,) for equally sized movers '
            'and a numpy array of shape (num_movers, 3) to specify an individual size for each mover.'
        )
        self.mover_size = np.tile(size, (self.num_movers, 1)) if size.shape == (3,) else size
        assert (self.mover_size > 0).all(), 'Mover size must be >0.'
        if isinstance(self.mover_mass, np.ndarray):
            assert self.mover_mass.shape == (self.num_movers,) and (self.mover_mass > 0).all(), 'Mover mass must be >0.'
        else:
            assert self.mover_mass > 0, 'Mover mass must be >0.'
        self.initial_mover_zpos = initial_mover_zpos
        assert self.initial_mover_zpos >= 0, 'Initial mover z position must be >= 0.'
        self.mover_names = [f'mover_{i}' for i in range(self.num_movers)]  # body names of BE:878

        # collision parameters (BE:257-264, 1548-1604)
        collision_params = collision_params or {}
        self.c_shape = collision_params.get('shape', 'circle')
        self.c_size = collision_params.get('size', 0.11)
        self.c_size_offset = collision_params.get('offset', 0.0)
        self.c_size_offset_wall = collision_params.get('offset_wall', 0.0)
        self._check_collision_params()

        self.num_envs = int(num_envs)
        assert self.num_envs >= 1
        self.cycle_time = 0.001  # MuJoCo option/timestep of the reference's MJCF (BE:1132, BE:283)
        self._handle = None
        self._host_only = bool(_host_only)  # configuration / table logic only (CPU unit tests); nothing can be computed
        if self._host_only:
            self.device, self._lib = None, None
            return
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise RuntimeError('the planar-robotics engine runs on MI355X only (device must be a CUDA/ROCm device); there is no CPU path')
        if self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self._lib = _native.load()

    # ------------------------------------------------------------------------------------------------------------
    # ctor-time tables
    # ------------------------------------------------------------------------------------------------------------
    def get_tile_xy_pos(self) -> tuple[np.ndarray, np.ndarray]:
        """(x,y) centres of all cells, each (num_tiles_x, num_tiles_y); same construction as BE:1292-1310."""
        def pos_1d(n: int, width: float) -> np.ndarray:
            return np.linspace(start=width / 2, stop=(n - 1) * width + width / 2, num=n, endpoint=True)

        return np.meshgrid(pos_1d(self.num_tiles_x, self.tile_size[0] * 2), pos_1d(self.num_tiles_y, self.tile_size[1] * 2), indexing='ij')

    def get_tile_indices_mask(self, mask: np.ndarray) -> tuple[np.ndarray, np.ndarray]:
        """Indices of the cells at which ``mask`` (0/1 pattern) matches the layout; 3x3 masks are anchored at their centre,
        2x2 masks at their upper-left cell (BE:1312-1339)."""
        mask = np.asarray(mask)
        assert mask.ndim == 2 and np.bitwise_or(mask == 0, mask == 1).all()
        mx, my = mask.shape
        ax, ay = (mx // 2 if mx % 2 else mx // 2 - 1), (my // 2 if my % 2 else my // 2 - 1)
        xs, ys = [], []
        for ix in range(ax, self.num_tiles_x - (mx - 1 - ax)):
            for iy in range(ay, self.num_tiles_y - (my - 1 - ay)):
                if (self.layout_tiles[ix - ax:ix - ax + mx, iy - ay:iy - ay + my] == mask).all():
                    xs.append(ix)
                    ys.append(iy)
        return np.array(xs, dtype=np.int64), np.array(ys, dtype=np.int64)

    def get_c_size_arr(self, c_size: float | np.ndarray, num_reps: int) -> np.ndarray:
        """(num_reps,1) for circles, (num_reps,2) for boxes (BE:1209-1242)."""
        if isinstance(c_size, float):
            assert self.c_shape == 'circle', 'Use a float value or a numpy array of shape (num_reps,) to specify the size parameter.'
            return np.full((num_reps, 1), c_size)
        c_size = np.asarray(c_size, dtype=np.float64)
        if self.c_shape == 'circle':
            return c_size.reshape((num_reps, 1))
        if c_size.shape == (2,):
            return np.tile(c_size, (num_reps, 1))
        return c_size.copy()

    def _c_size_pairs(self, c_size, n: int) -> np.ndarray:
        """[n,2] float64 array in the layout the C ABI takes (column 0 = radius for circles)."""
        arr = self.get_c_size_arr(c_size, n)
        out = np.zeros((n, 2))
        out[:, : arr.shape[1]] = arr
        return out

    def _check_collision_params(self) -> None:
        assert self.c_shape in ('circle', 'box'), 'Unexpected collision shape. You can choose between circle and box.'
        if self.c_shape == 'circle' and isinstance(self.c_size, np.ndarray):
            assert self.c_size.shape == (self.num_movers,), (
                'Use a single float value (radius) or a numpy array of shape (num_movers,) to specify the size parameter.'
            )
        elif self.c_shape == 'box':
            assert not isinstance(self.c_size, float), 'Use a numpy array of shape (2,) or (num_movers,2) to specify the size parameter.'
            assert self.c_size.shape == (2,) or self.c_size.shape == (self.num_movers, 2), (
                'The shape of the size array (collision_params["size"]) has to be (2,) or (num_movers,2).'
            )
        sizes = self._c_size_pairs(self.c_size, self.num_movers)
        for i in range(self.num_movers):  # warning of BE:1577-1598
            sx, sy = self.mover_size[i, 0], (self.mover_size[i, 1] if self.mover_shape == 'box' else self.mover_size[i, 0])
            inside = sizes[i, 0] >= np.hypot(sx, sy) if self.c_shape == 'circle' else (sizes[i] >= np.array([sx, sy])).all()
            if not inside:
                warnings.warn(f'Mover {i} is not completely included in collision shape. Choose a larger collision_params["size"].')
        assert isinstance(self.c_size_offset, float), 'Use a single float value to specify the size offset.'
        assert isinstance(self.c_size_offset_wall, float), 'Use a single float value to specify the wall offset.'
        assert self.c_size_offset >= 0, 'collision_params["offset"] must be >= 0.'
        assert self.c_size_offset_wall >= 0, 'collision_params["offset_wall"] must be >= 0.'

    # ------------------------------------------------------------------------------------------------------------
    # native handle
    # ------------------------------------------------------------------------------------------------------------
    def _make_config(self, **kw) -> _native.GprConfig:
        cfg = _native.GprConfig()
        cfg.num_envs = self.num_envs
        cfg.env_id_offset = kw.get('env_id_offset', 0)
        cfg.num_movers = self.num_movers
        cfg.num_tiles_x, cfg.num_tiles_y = self.num_tiles_x, self.num_tiles_y
        self._keep = dict(
            layout=self.layout_tiles,
            xc=np.ascontiguousarray(self.x_pos_tiles[:, 0], dtype=np.float64),
            yc=np.ascontiguousarray(self.y_pos_tiles[0, :], dtype=np.float64),
            cs=np.ascontiguousarray(self._c_size_pairs(self.c_size, self.num_movers).reshape(-1)),
        )
        cfg.layout_tiles = self._keep['layout'].ctypes.data_as(C.POINTER(C.c_int8))
        cfg.tile_half_size[0], cfg.tile_half_size[1] = float(self.tile_size[0]), float(self.tile_size[1])
        cfg.tile_x_centres = self._keep['xc'].ctypes.data_as(C.POINTER(C.c_double))
        cfg.tile_y_centres = self._keep['yc'].ctypes.data_as(C.POINTER(C.c_double))
        cfg.c_shape = _native.SHAPE[self.c_shape]
        cfg.c_size = self._keep['cs'].ctypes.data_as(C.POINTER(C.c_double))
        cfg.c_offset, cfg.c_offset_wall = float(self.c_size_offset), float(self.c_size_offset_wall)
        cfg.cycle_time = self.cycle_time
        cfg.num_cycles = kw.get('num_cycles', 40)
        cfg.v_max, cfg.a_max, cfg.j_max = kw.get('v_max', 2.0), kw.get('a_max', 10.0), kw.get('j_max', 100.0)
        cfg.learn_jerk = int(kw.get('learn_jerk', False))
        cfg.threshold_pos = kw.get('threshold_pos', 0.1)
        for i in range(3):
            cfg.std_noise[i] = float(self.std_noise[i])
        cfg.max_episode_steps = kw.get('max_episode_steps', 0) or 0
        cfg.autoreset_mode = _native.AUTORESET[kw.get('autoreset_mode', 'disabled')]
        mn, mx = kw.get('min_xy_pos', np.zeros(2)), kw.get('max_xy_pos', np.ones(2))
        cfg.min_xy_pos[0], cfg.min_xy_pos[1], cfg.max_xy_pos[0], cfg.max_xy_pos[1] = float(mn[0]), float(mn[1]), float(mx[0]), float(mx[1])
        cfg.min_goal_dist = float(kw.get('min_goal_dist', 0.0))
        cfg.max_reset_trials = int(kw.get('max_reset_trials', 10000))
        cfg.precision = _native.PRECISION[kw.get('precision', 'f32')]
        return cfg

    def _create_handle(self, **kw) -> None:
        cfg = self._make_config(**kw)
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _native.check(self._lib.gpr_create(C.byref(cfg), C.byref(handle)))
        self._handle = handle

    def close(self, **kwargs) -> None:
        if getattr(self, '_handle', None) is not None:
            try:
                self.warn_diagnostics()  # what the reference would have printed while it ran (reset searches without a valid sample, clipped actions)
            except Exception:
                pass
            self._lib.gpr_destroy(self._handle)
            self._handle = None
        self.closed = True

    def render(self):
        """rendering is out of scope of the GPU engine (render_mode is always None)"""
        return None

    @property
    def unwrapped(self):
        return self

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self) -> int:
        return torch.cuda.current_stream(self.device).cuda_stream

    # ------------------------------------------------------------------------------------------------------------
    # the template's extension hooks (BE:1952-1986)
    # ------------------------------------------------------------------------------------------------------------
    def _step_callback(self, action: torch.Tensor) -> None:
        """Called by ``step()`` with the batched action right before the step kernel is launched (BE:1876, 1960-1966): a subclass may
        use it for host-side bookkeeping per call. The reference's other two hooks run INSIDE the 40-cycle loop of one env
        (``_mujoco_step_callback`` BE:1881, ``_check_for_other_collisions_callback`` BE:1903) and have no Python counterpart here:
        the loop lives in the kernel. What they are used for in the two benchmark envs is in the kernel (the per-cycle control,
        PL:420-450 / PU:419-455); both envs leave the collision hook at its default (no other collision). Static obstacles are
        expressed as missing tiles of ``layout_tiles`` (tile granularity): a mover reaching over one is a wall collision."""
        return None

    # ------------------------------------------------------------------------------------------------------------
    # engine additions with no reference counterpart: resumable random streams, checkpoints, episode counters
    # ------------------------------------------------------------------------------------------------------------
    def get_rng_state(self) -> tuple[int, int]:
        """(seed, tick) of the handle's counter-based random streams (RNG spec gpr-rng-v2, include/gpr.h)."""
        seed, tick = C.c_uint64(), C.c_uint64()
        _native.check(self._lib.gpr_get_rng_state(self._handle, C.byref(seed), C.byref(tick)), self._handle)
        return int(seed.value), int(tick.value)

    def set_rng_state(self, seed: int, tick: int) -> None:
        _native.check(self._lib.gpr_set_rng_state(self._handle, int(seed), int(tick)), self._handle)
        self._needs_seed = False

    def _state_tensors(self) -> dict[str, torch.Tensor]:
        raise NotImplementedError

    def state_dict(self) -> dict[str, Any]:
        """Everything a run consists of: the state tensors (cloned) and the position of the random streams. ``torch.save`` it;
        ``load_state_dict`` on an env of the same configuration continues the run bit for bit."""
        seed, tick = self.get_rng_state()
        return {'rng_seed': seed, 'rng_tick': tick, **{k: v.detach().clone() for k, v in self._state_tensors().items()}}

    def load_state_dict(self, sd: dict[str, Any]) -> None:
        for k, v in self._state_tensors().items():
            v.copy_(torch.as_tensor(sd[k]).to(device=v.device, dtype=v.dtype))
        self.set_rng_state(sd['rng_seed'], sd['rng_tick'])

    def set_wave_budget(self, max_wavefronts: int = -1) -> None:
        """Thin wavefronts (``include/gpr.h: gpr_set_wave_budget``): the most wavefronts a ``step()`` launch of this env is spread over -- a
        small batch is cut into more wavefronts holding fewer envs each, which shortens its slowest wavefront; results do not depend on it.
        0 turns it off, a negative value restores the default. (No reference counterpart: the batch axis is the engine's.)"""
        _native.check(self._lib.gpr_set_wave_budget(self._handle, int(max_wavefronts)), self._handle)

    def enable_episode_counters(self, enable: bool = True) -> None:
        """Let the step kernels count episode ends on the device (per-wavefront popcounts of the done masks; no host sync)."""
        with torch.cuda.device(self.device):
            _native.check(self._lib.gpr_enable_counters(self._handle, int(enable)), self._handle)
        if not hasattr(self, '_counter_totals'):
            self._counter_totals = torch.zeros(len(_native.COUNTERS), dtype=torch.int64, device=self.device)
        self._counters_on = bool(enable)

    def read_episode_counters(self) -> torch.Tensor:
        """int64 device tensor [8] = running totals in the order of ``_native.COUNTERS`` (episodes, terminated, truncated, success,
        wall_collision, mover_collision, reset_exhausted, actions_clipped) since ``enable_episode_counters``. One tiny reduce kernel, no
        host sync."""
        assert getattr(self, '_counters_on', False), 'call enable_episode_counters() first'
        with torch.cuda.device(self.device):
            _native.check(self._lib.gpr_read_counters(self._handle, self._counter_totals.data_ptr(), self._stream()), self._handle)
        return self._counter_totals

    def all_reduce_episode_counters(self) -> torch.Tensor:
        """Job-wide totals: ONE all_reduce (SUM) of the eight counters over the default process group -- the only collective a
        rollout flush needs when every rank's learner consumes its own shard."""
        from .. import distributed
        if not getattr(self, '_counters_on', False):
            self.enable_episode_counters()
        return distributed.all_reduce_sum(self.read_episode_counters().clone())

    # ------------------------------------------------------------------------------------------------------------
    # the reference's warnings, batched: counted on the device, reported on the host
    # ------------------------------------------------------------------------------------------------------------
    def count_clipped_actions(self, enable: bool = True) -> None:
        """Let the step kernels count the env-steps whose action lies outside the action space. The reference prints 'action is not within
        the action space' and clips (BE:1871-1873); the engine always clips, and -- after this call -- counts, so that
        ``diagnostics()`` / ``warn_diagnostics()`` can say how often."""
        _native.check(self._lib.gpr_count_clipped_actions(self._handle, int(enable)), self._handle)

    def diagnostics(self) -> dict[str, int]:
        """{'reset_exhausted': rejection searches of resets (starts, goals, the pushing env's object) that used up ``max_reset_trials`` -- the
        last candidate then stands although it is INVALID; 'actions_clipped': env-steps with an out-of-range action (after
        ``count_clipped_actions()``)} since the env was created. Waits for the env's stream (a diagnostic read)."""
        tot = (C.c_uint64 * 2)()
        with torch.cuda.device(self.device):
            _native.check(self._lib.gpr_read_diagnostics(self._handle, tot, self._stream()), self._handle)
        return {'reset_exhausted': int(tot[0]), 'actions_clipped': int(tot[1])}

    def warn_diagnostics(self) -> dict[str, int]:
        """``diagnostics()``, and a ``warnings.warn`` for every count that grew since the last call: what the reference logs while it runs
        (PL:371-375, 397-401, PU:398-402: 'Trying to find a start / goal configuration ...'; BE:1871-1873). Called by ``reset()`` and
        ``close()``; call it from a training loop's logging step to hear about auto-resets in time."""
        d = self.diagnostics()
        seen = getattr(self, '_diag_seen', {'reset_exhausted': 0, 'actions_clipped': 0})
        if d['reset_exhausted'] > seen['reset_exhausted']:
            warnings.warn(f"{d['reset_exhausted'] - seen['reset_exhausted']} reset search(es) found no valid start / goal configuration within "
                          f"max_reset_trials = {getattr(self, 'max_reset_trials', '?')} trials: the last (invalid) sample was kept. Consider choosing "
                          'more tiles or fewer movers (the reference loops on and warns every 100 trials).')
        if d['actions_clipped'] > seen['actions_clipped']:
            warnings.warn(f"{d['actions_clipped'] - seen['actions_clipped']} env-step(s) had an action outside the action space; it was clipped "
                          '(BE:1871-1873).')
        self._diag_seen = d
        return d

    # ------------------------------------------------------------------------------------------------------------
    # noisy state getters (BE:801-855, 1244-1290), batched over the envs
    # ------------------------------------------------------------------------------------------------------------
    def _mover_state(self, which: int, mover_names, add_noise: bool) -> torch.Tensor:
        idx = [self.mover_names.index(n) for n in mover_names]  # (a name that is no mover raises ValueError, like BE:813)
        sel = torch.tensor(idx, dtype=torch.int32, device=self.device)
        out = torch.empty(self.num_envs, len(idx), 7 if which == 0 else 6, dtype=self.dtype, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(self._lib.gpr_get_mover_state(self._handle, which, len(idx), sel.data_ptr(), float(self.initial_mover_zpos), int(add_noise),
                                                        out.data_ptr(), self._stream()), self._handle)
        return out

    def get_mover_qpos(self, mover_name: str, add_noise: bool = False) -> torch.Tensor:
        """``get_mover_qpos`` (BE:801-828) of every env: ``[num_envs, 7]`` = (x_p, y_p, z_p, w_o, x_o, y_o, z_o). As in the reference, z is
        the distance between the bottom of the mover and the top of a tile (BE:821-824) -- ``initial_mover_zpos`` throughout: the engine's
        movers are planar. ``add_noise``: Gaussian noise of ``std_noise[0]`` on every component."""
        return self._mover_state(0, [mover_name], add_noise)[:, 0]

    def get_mover_qvel(self, mover_name: str, add_noise: bool = False) -> torch.Tensor:
        """``get_mover_qvel`` (BE:830-841): ``[num_envs, 6]`` = (x, y, z, a, b, c) velocities; noise ``std_noise[1]``."""
        return self._mover_state(1, [mover_name], add_noise)[:, 0]

    def get_mover_qacc(self, mover_name: str, add_noise: bool = False) -> torch.Tensor:
        """``get_mover_qacc`` (BE:843-855): ``[num_envs, 6]`` accelerations; noise ``std_noise[2]``."""
        return self._mover_state(2, [mover_name], add_noise)[:, 0]

    def get_mover_qpos_arr(self, mover_names: list[str], add_noise: bool = False) -> torch.Tensor:
        """``get_mover_qpos_arr`` (BE:1244-1258): ``[num_envs, len(mover_names), 7]``, rows in the order of the names."""
        return self._mover_state(0, mover_names, add_noise)

    def get_mover_qvel_arr(self, mover_names: list[str], add_noise: bool = False) -> torch.Tensor:
        """``get_mover_qvel_arr`` (BE:1260-1274): ``[num_envs, len(mover_names), 6]``."""
        return self._mover_state(1, mover_names, add_noise)

    def get_mover_qacc_arr(self, mover_names: list[str], add_noise: bool = False) -> torch.Tensor:
        """``get_mover_qacc_arr`` (BE:1276-1290): ``[num_envs, len(mover_names), 6]``."""
        return self._mover_state(2, mover_names, add_noise)

    def _preprocess_info(self, info) -> tuple[torch.Tensor, torch.Tensor, bool]:
        """``_preprocess_info_dict`` (PL:666-688, PU:670-689): the collision flags of a batch as uint8 device tensors. Accepts what the
        reference accepts -- ONE info dict (batch of 1; the caller then returns a scalar like the reference's ``reward[0]``) or a NumPy
        array / list / tuple of info dicts (what e.g. SB3's HER replay buffer passes) -- and, batched engine style, a dict whose values
        are (batch,) tensors / arrays (what ``step()`` returns). -> (mover_collision, wall_collision, scalar)"""
        if isinstance(info, (np.ndarray, list, tuple)):
            items = list(np.asarray(info, dtype=object).reshape(-1)) if isinstance(info, np.ndarray) else list(info)
            for it in items:  # the reference indexes both keys (PL:678-679): a dict without one of them is an error there too
                assert isinstance(it, dict) and 'wall_collision' in it and 'mover_collision' in it, (
                    "every info dict needs the keys 'mover_collision' and 'wall_collision'")
            flag = lambda it, k: bool(np.asarray(torch.as_tensor(it[k]).cpu()).reshape(-1)[0])  # noqa: E731
            mc = torch.tensor([flag(it, 'mover_collision') for it in items], dtype=torch.uint8)
            wc = torch.tensor([flag(it, 'wall_collision') for it in items], dtype=torch.uint8)
            # `reward if batch_size > 1 else reward[0]` (PL:534): an array holding ONE info dict gives a scalar, like one dict
            return mc.to(self.device), wc.to(self.device), len(items) == 1
        assert isinstance(info, dict), 'info must be an info dict or an array / list of info dicts'
        wc = torch.as_tensor(info['wall_collision'], device=self.device)
        mc_in = info.get('mover_collision')
        mc = torch.as_tensor(mc_in, device=self.device) if mc_in is not None else torch.zeros_like(wc)
        scalar = wc.ndim == 0
        assert not scalar or mc_in is not None, "an info dict needs the keys 'mover_collision' and 'wall_collision'"  # PL:685-686
        return mc.reshape(-1).to(torch.uint8).contiguous(), wc.reshape(-1).to(torch.uint8).contiguous(), scalar

    def _dev_f64(self, a) -> torch.Tensor:
        if isinstance(a, torch.Tensor):
            return a.to(device=self.device, dtype=torch.float64).contiguous()
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=self.device)

    # ------------------------------------------------------------------------------------------------------------
    # collision / validity operators (BE:355-788) on device
    # ------------------------------------------------------------------------------------------------------------
    def qpos_is_valid(self, qpos, c_size: float | np.ndarray, add_safety_offset: bool = False) -> torch.Tensor:
        """``qpos_is_valid`` (BE:459-788): qpos (n,7) -> int32 (n,), 1 = valid. Runs the f64 HIP kernel."""
        q = self._dev_f64(qpos)
        assert q.ndim == 2 and q.shape[1] == 7
        n = q.shape[0]
        cs = self._dev_f64(self._c_size_pairs(c_size, n))
        out = torch.empty(n, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(self._lib.gpr_qpos_is_valid(self._handle, n, q.data_ptr(), cs.data_ptr(), int(add_safety_offset), out.data_ptr(),
                                                      self._stream()), self._handle)
        return out

    def check_wall_collision(self, mover_qpos, c_size: float | np.ndarray, add_safety_offset: bool = False) -> torch.Tensor:
        """``check_wall_collision`` (BE:426-457) for explicitly given qpos: 1 where the qpos collides with a wall."""
        return 1 - self.qpos_is_valid(mover_qpos, c_size, add_safety_offset)

    def check_mover_collision(self, mover_qpos, c_size: float | np.ndarray, add_safety_offset: bool = False) -> torch.Tensor:
        """``check_mover_collision`` (BE:355-424) for explicitly given qpos: (num_movers,7) -> bool scalar tensor, or a batch
        (batch,num_movers,7) -> bool (batch,)."""
        q = self._dev_f64(mover_qpos)
        single = q.ndim == 2
        if single:
            q = q.unsqueeze(0)
        assert q.shape[1:] == (self.num_movers, 7)
        cs = self._dev_f64(self._c_size_pairs(c_size, self.num_movers))
        out = torch.empty(q.shape[0], dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(self._lib.gpr_check_mover_collision(self._handle, q.shape[0], q.data_ptr(), cs.data_ptr(), int(add_safety_offset),
                                                              out.data_ptr(), self._stream()), self._handle)
        return out.bool()[0] if single else out.bool()

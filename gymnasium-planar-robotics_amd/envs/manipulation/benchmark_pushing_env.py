"""``VecBenchmarkPushingEnv``: ``num_envs`` copies of the reference's ``BenchmarkPushingEnv`` (one mover pushes one box over 3x3
tiles; ``num_movers=2`` is the engine's own two-mover variant, BASELINE.json's config 3), one fused HIP kernel launch per ``step()``.

Reference: gymnasium_planar_robotics/envs/manipulation/benchmark_pushing_env.py (PU:line). Same constructor keywords, observation
dict (``observation`` = mover position and velocity (and acceleration with ``learn_jerk``), ``achieved_goal`` = object position,
``desired_goal`` = object goal), reward (-50 wall collision / 0 at the goal / -1), termination (wall collision only), info keys
and TimeLimit (50) as the registered ``BenchmarkPushingEnv-v0``.

**Parity status.** Pinned by the reference: the mover's actuation and clamping (its 24 actuator tests), the observation / reward /
termination / info rules, the reset sampling rules, the wall check. **Not pinned**: contact. The reference hands mover-object and
object-tile contact, friction and the soft impedance hold of the mover's orientation to MuJoCo; MuJoCo is not available to this
build and no reference test exercises contact. The kernels therefore integrate the engine's own planar rigid-body model
``gpr-push-v3`` (include/gpr.h): 3-DoF boxes, separating-axis test, a clipped two-point manifold with a 0.1 mm
contact margin, the pair's two normal rows solved exactly + one friction row, two block Gauss-Seidel visits per cycle, a 20 ms
penetration-recovery time constant, Coulomb + torsional friction against the tiles, the reference's joint damping and impedance gains on
the yaw. It is MuJoCo-like in its parameters, not MuJoCo-equal in its
trajectories.

**Two movers** (``num_movers=2``; the reference hard-codes one, PU:196): both movers are actuated (action ``(num_envs, 4)``) and
push the same object; the observation keeps the reference's ``concatenate((pos, vel[, acc]), axis=0).flatten()`` = all positions,
all velocities[, all accelerations] (PU:540-545); the movers do not touch each other -- their collision circles meeting ends the
episode like a wall collision (reward -50, ``info['mover_collision']``), the planning env's rule (BE:355-424); a reset re-draws the
two starts together until the circles (with the safety offset) are apart, and the object keeps ``min_mo_dist`` from both.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Any

import numpy as np
import torch

from ... import _native, spaces
from ..basic_envs import VecBasicPlanarRoboticsEnv

_TORCH_DTYPE = {'f32': torch.float32, 'f32_f64acc': torch.float32, 'f64': torch.float64}


class VecBenchmarkPushingEnv(VecBasicPlanarRoboticsEnv):
    """Vectorised ``BenchmarkPushingEnv`` (PU:101-689). Reference keywords (PU:154-169) + the engine keywords of
    ``VecBenchmarkPlanningEnv`` (num_envs, device, max_episode_steps, autoreset_mode, precision, env_id_offset)."""

    reward_wall_collision = -50  # PU:219

    def __init__(
        self,
        mover_params: dict[str, Any] | None = None,
        initial_mover_zpos: float = 0.003,
        std_noise: np.ndarray | float = 1e-5,
        render_mode: str | None = None,
        render_every_cycle: bool = False,
        num_cycles: int = 40,
        collision_params: dict[str, Any] | None = None,
        v_max: float = 2.0,
        a_max: float = 10.0,
        j_max: float = 100.0,
        learn_jerk: bool = False,
        threshold_pos: float = 0.05,
        use_mj_passive_viewer: bool = False,
        *,
        num_envs: int = 1,
        device: str | torch.device = 'cuda',
        max_episode_steps: int | None = 50,
        autoreset_mode: str = 'same_step',
        precision: str = 'f32',
        env_id_offset: int = 0,
        max_reset_trials: int = 10000,
        num_movers: int = 1,
        warn_on_reset: bool = True,
        _host_only: bool = False,
        _push_overrides: dict[str, float] | None = None,
    ) -> None:
        if render_every_cycle or use_mj_passive_viewer:
            raise ValueError('rendering is out of scope of the GPU engine')
        assert precision in _native.PRECISION and autoreset_mode in _native.AUTORESET
        # object parameters (PU:172-178)
        self.object_length_xy = 0.07 / 2
        self.object_height = 0.04 / 2
        self.object_mass = 0.01
        self.object_noise_xy_pos = 1e-5
        self.idx_mover = 0
        assert num_movers in (1, 2), 'the pushing env has one mover (PU:196) or, as an engine extension, two'
        super().__init__(layout_tiles=np.ones((3, 3)), num_movers=num_movers, num_envs=num_envs, tile_params=None, mover_params=mover_params,
                         initial_mover_zpos=initial_mover_zpos, std_noise=std_noise, render_mode=render_mode,
                         collision_params=collision_params, device=device, _host_only=_host_only)  # PU:195-196
        assert self.c_shape == 'circle', "the GPU pushing env supports the default 'circle' collision shape"
        self.learn_jerk, self.num_cycles = bool(learn_jerk), int(num_cycles)
        self.v_max, self.a_max, self.j_max, self.threshold_pos = float(v_max), float(a_max), float(j_max), float(threshold_pos)
        self.max_episode_steps, self.autoreset_mode, self.precision = max_episode_steps, autoreset_mode, precision
        self.env_id_offset, self.dtype = int(env_id_offset), _TORCH_DTYPE[precision]
        self.warn_on_reset, self.max_reset_trials = bool(warn_on_reset), int(max_reset_trials)
        self.metadata = dict(self.metadata, autoreset_mode=spaces.autoreset_metadata(autoreset_mode))

        # spaces (PU:232-253)
        low_goals = np.zeros(2)
        high_goals = np.array([np.max(self.x_pos_tiles) + (self.tile_size[0] / 2), np.max(self.y_pos_tiles) + (self.tile_size[1] / 2)])
        NM = self.num_movers
        self.obs_dim = (2 + int(self.learn_jerk)) * 2 * NM
        self.single_observation_space = spaces.Dict({
            'observation': spaces.Box(low=-np.inf, high=np.inf, shape=(self.obs_dim,), dtype=np.float64),
            'achieved_goal': spaces.Box(low=low_goals, high=high_goals, dtype=np.float64),
            'desired_goal': spaces.Box(low=low_goals, high=high_goals, dtype=np.float64),
        })
        lim = self.j_max if self.learn_jerk else self.a_max
        self.single_action_space = spaces.Box(low=-lim, high=lim, shape=(2 * NM,), dtype=np.float64)
        self.observation_space = spaces.Dict({k: spaces.batch_box(v, self.num_envs) for k, v in self.single_observation_space.items()})
        self.action_space = spaces.batch_box(self.single_action_space, self.num_envs)

        # sampling ranges (PU:255-258) and the minimum mover-object distance after a reset (PU:279-288)
        safety_margin = self.c_size + self.c_size_offset_wall + self.c_size_offset
        self.min_xy_pos = np.zeros(2) + safety_margin
        self.max_xy_pos = high_goals - safety_margin
        self.object_min_xy_pos = self.min_xy_pos + safety_margin
        self.object_max_xy_pos = self.max_xy_pos - safety_margin
        self.min_mo_dist = max(float(np.linalg.norm(self.object_length_xy + self.mover_size.flatten()[:2], ord=2)),
                               float(self.c_size + self.c_size_offset))
        if self._host_only:
            return

        self._create_handle(env_id_offset=self.env_id_offset, num_cycles=self.num_cycles, v_max=self.v_max, a_max=self.a_max,
                            j_max=self.j_max, learn_jerk=self.learn_jerk, threshold_pos=self.threshold_pos,
                            max_episode_steps=max_episode_steps, autoreset_mode=autoreset_mode, min_xy_pos=self.min_xy_pos,
                            max_xy_pos=self.max_xy_pos, min_goal_dist=0.0, max_reset_trials=max_reset_trials, precision=precision)
        E, dev, dt = self.num_envs, self.device, self.dtype
        z = lambda *shape, dtype=dt: torch.zeros(*shape, dtype=dtype, device=dev)  # noqa: E731
        self.pos, self.vel, self.acc, self.goals = z(E, 2 * NM), z(E, 2 * NM), z(E, 2 * NM), z(E, 2)  # mover x,y state; object goal
        self.mover_rot = z(E, 2 * NM)   # (yaw, yaw rate) per mover
        self.object = z(E, 6)      # object x, y, yaw, vx, vy, yaw rate
        self.elapsed_steps = z(E, dtype=torch.int32)
        self._obs = OrderedDict(observation=z(E, self.obs_dim), achieved_goal=z(E, 2), desired_goal=z(E, 2))
        self._final = OrderedDict(observation=z(E, self.obs_dim), achieved_goal=z(E, 2), desired_goal=z(E, 2))
        self._reward = z(E)
        u8 = lambda: z(E, dtype=torch.uint8)  # noqa: E731
        self._terminated, self._truncated, self._is_success, self._mover_collision, self._wall_collision = u8(), u8(), u8(), u8(), u8()
        state = _native.GprState(self.pos.data_ptr(), self.vel.data_ptr(), self.acc.data_ptr(), self.goals.data_ptr(), self.elapsed_steps.data_ptr())
        _native.check(self._lib.gpr_bind_state(self._handle, C.byref(state)), self._handle)
        mass = float(self.mover_mass) if not isinstance(self.mover_mass, np.ndarray) else float(self.mover_mass[0])
        self._push_params = _native.GprPushParams(
            mass, (C.c_double * 2)(float(self.mover_size[0, 0]), float(self.mover_size[0, 1])), self.object_length_xy, self.object_mass,
            0.01,            # free-joint damping (PU:335)
            1.0, 1.0, 0.005,  # MuJoCo default sliding / torsional friction
            9.81, 0.1,        # gravity; impedance rotational stiffness (PU:266)
            0.02,             # MuJoCo default solref time constant
            self.object_noise_xy_pos, self.min_mo_dist, (C.c_double * 2)(*self.object_min_xy_pos), (C.c_double * 2)(*self.object_max_xy_pos))
        for k, v in (_push_overrides or {}).items():  # (tests of the contact model's invariants switch friction / damping off)
            setattr(self._push_params, k, v)
        pstate = _native.GprPushState(self.mover_rot.data_ptr(), self.object.data_ptr())
        _native.check(self._lib.gpr_push_configure(self._handle, C.byref(self._push_params), C.byref(pstate)), self._handle)
        use_final = autoreset_mode == 'same_step'
        self._out = _native.GprStepOut(
            self._obs['observation'].data_ptr(), self._obs['achieved_goal'].data_ptr(), self._obs['desired_goal'].data_ptr(),
            self._reward.data_ptr(), self._terminated.data_ptr(), self._truncated.data_ptr(), self._is_success.data_ptr(),
            self._mover_collision.data_ptr(), self._wall_collision.data_ptr(),
            self._final['observation'].data_ptr() if use_final else None, self._final['achieved_goal'].data_ptr() if use_final else None,
            self._final['desired_goal'].data_ptr() if use_final else None)
        self._needs_seed = True

    def _info_dict(self) -> dict[str, torch.Tensor]:
        return {'is_success': self._is_success.view(torch.bool), 'mover_collision': self._mover_collision.view(torch.bool),
                'wall_collision': self._wall_collision.view(torch.bool)}

    def _state_tensors(self) -> dict[str, torch.Tensor]:
        return {'pos': self.pos, 'vel': self.vel, 'acc': self.acc, 'goals': self.goals, 'mover_rot': self.mover_rot, 'object': self.object,
                'elapsed_steps': self.elapsed_steps}

    supports_two_movers = True

    def reset(self, seed: int | None = None, options: dict[str, Any] | None = None):
        """PU:373-417 + BE:1770-1833. ``options``: ``'mask'`` (bool tensor) resets a subset; the reference's
        ``'mover_start_xy_pos'`` ((num_movers, 2), or batched (num_envs, num_movers, 2)) fixes the mover start(s) -- the object's start
        is still rejection-sampled to keep ``min_mo_dist`` from it (PU:396-407) -- and ``'object_goal_xy_pos'`` ((2,) or
        (num_envs, 2)) fixes the goal (PU:389-391, 412-414). Engine option ``'object_candidates'`` ((n_cand, num_envs, 2)): the object's start
        candidates of PU:403-405 from the caller instead of the engine's draws, tried in order (``info['candidates_used']``)."""
        if seed is not None or self._needs_seed:
            _native.check(self._lib.gpr_seed(self._handle, int(seed if seed is not None else np.random.SeedSequence().entropy % (1 << 63))),
                          self._handle)
            self._needs_seed = False
        options = options or {}
        mask = None
        if options.get('mask') is not None:
            mask = torch.as_tensor(options['mask'], device=self.device).to(torch.uint8).contiguous()

        def fixed(key, width):
            if options.get(key) is None:
                return None
            t = torch.as_tensor(np.asarray(options[key], dtype=np.float64) if not isinstance(options[key], torch.Tensor) else options[key],
                                device=self.device).to(self.dtype).reshape(-1, width)
            assert t.shape[0] in (1, self.num_envs), f'options[{key!r}] must hold one value or one per env'
            return t.expand(self.num_envs, width).contiguous()

        start, goal = fixed('mover_start_xy_pos', 2 * self.num_movers), fixed('object_goal_xy_pos', 2)
        if options.get('object_candidates') is not None:
            cand = torch.as_tensor(options['object_candidates'], device=self.device).to(self.dtype).reshape(-1, self.num_envs, 2).contiguous()
            used = torch.zeros(self.num_envs, dtype=torch.int32, device=self.device)
            with torch.cuda.device(self.device):
                _native.check(self._lib.gpr_push_reset_candidates(self._handle, mask.data_ptr() if mask is not None else None,
                                                                  start.data_ptr() if start is not None else None,
                                                                  goal.data_ptr() if goal is not None else None, cand.shape[0], cand.data_ptr(),
                                                                  used.data_ptr(), C.byref(self._out), self._stream()), self._handle)
            return OrderedDict(self._obs), dict(self._info_dict(), candidates_used=used)
        with torch.cuda.device(self.device):
            _native.check(self._lib.gpr_push_reset_opts(self._handle, mask.data_ptr() if mask is not None else None,
                                                        start.data_ptr() if start is not None else None,
                                                        goal.data_ptr() if goal is not None else None, C.byref(self._out), self._stream()),
                          self._handle)
        if self.warn_on_reset:
            self.warn_diagnostics()  # PU:398-402: an object search that found no valid sample is reported (one host synchronisation)
        return OrderedDict(self._obs), self._info_dict()

    def step(self, action: torch.Tensor):
        """BE:1835-1950 with the callbacks of PU:419-607; action (num_envs, 2 * num_movers): mover acceleration(s) (or jerk(s))."""
        if not isinstance(action, torch.Tensor):
            action = torch.as_tensor(np.asarray(action), device=self.device)
        assert tuple(action.shape) == (self.num_envs, 2 * self.num_movers), 'action dim != action_space dim'
        if action.dtype != self.dtype or action.device != self.device or not action.is_contiguous():
            action = action.to(device=self.device, dtype=self.dtype).contiguous()
        self._step_callback(action)  # BE:1876
        with torch.cuda.device(self.device):
            _native.check(self._lib.gpr_push_step(self._handle, action.data_ptr(), C.byref(self._out), self._stream()), self._handle)
        info = self._info_dict()
        if self.autoreset_mode == 'same_step':
            info['final_obs'] = self._final
        return OrderedDict(self._obs), self._reward, self._terminated.view(torch.bool), self._truncated.view(torch.bool), info

    def compute_reward(self, achieved_goal, desired_goal, info) -> torch.Tensor:
        """PU:502-527 on batches (plain tensor algebra: three elementwise operations, no kernel of its own)."""
        ag = torch.as_tensor(achieved_goal, device=self.device).reshape(-1, 2).double()
        dg = torch.as_tensor(desired_goal, device=self.device).reshape(-1, 2).double()
        mc, wc, scalar = self._preprocess_info(info)  # one info dict, an array / list of info dicts (PU:670-689), or a dict of tensors
        wc = wc.bool()
        if self.num_movers > 1:  # (two movers, engine-defined: their collision circles meeting ends the episode like a wall)
            wc = wc | mc.bool()
        assert wc.shape == (ag.shape[0],), 'achieved_goal and info disagree on the batch size'
        reached = (ag - dg).norm(dim=1) <= self.threshold_pos
        r = torch.where(wc, torch.full_like(reached, -50.0, dtype=torch.float64), torch.full_like(reached, -1.0, dtype=torch.float64))
        r = torch.where(reached & ~wc, torch.zeros_like(r), r).to(self.dtype)
        return r[0] if scalar else r

    def compute_terminated(self, achieved_goal, desired_goal, info) -> torch.Tensor:
        return self.compute_reward(achieved_goal, desired_goal, info) == self.reward_wall_collision  # PU:475

    def compute_truncated(self, achieved_goal, desired_goal, info=None) -> torch.Tensor:
        batch = achieved_goal.shape[0] if achieved_goal.ndim > 1 else 1
        return torch.zeros(batch, dtype=torch.bool, device=self.device)

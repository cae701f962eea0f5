"""``VecBenchmarkPlanningEnv``: ``num_envs`` copies of the reference's ``BenchmarkPlanningEnv`` stepped by ONE fused HIP kernel
launch per ``step()`` on an MI355X.

Reference: gymnasium_planar_robotics/envs/planning/benchmark_planning_env.py (PL:line) and the step()/reset() template of
gymnasium_planar_robotics/envs/basic_envs.py (BE:1770-1950). Same constructor keywords, observation-dict keys and orderings,
reward constants, termination rule, info keys and TimeLimit (50 steps) as the registered ``BenchmarkPlanningEnv-v0`` --
batched: every returned array is a torch tensor on the device with a leading ``num_envs`` axis.

There is no MuJoCo here. For this env the reference's MJCF has no contact that can fire before termination, no gravity
effect and no damping, so ``mj_step`` is the clamped semi-implicit Euler integrator that the reference's own tests assert
(tests/test_benchmark_planning_env.py:86-121,199-231); the kernel integrates exactly that.
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


class VecBenchmarkPlanningEnv(VecBasicPlanarRoboticsEnv):
    """Vectorised ``BenchmarkPlanningEnv`` (PL:98-688).

    Reference keywords (PL:165-185): layout_tiles, num_movers, show_2D_plot, mover_colors_2D_plot, tile_params, mover_params,
    initial_mover_zpos, std_noise, render_mode, render_every_cycle, num_cycles, collision_params, v_max, a_max, j_max,
    learn_jerk, threshold_pos, use_mj_passive_viewer. Rendering keywords are accepted for signature compatibility but must be
    off (``render_mode=None``, ``show_2D_plot=False``).

    Engine keywords:

    :param num_envs: envs on this device (this rank's shard)
    :param device: CUDA/ROCm device
    :param max_episode_steps: TimeLimit of the registered id (gymnasium_planar_robotics/__init__.py:28); ``None`` disables it
    :param autoreset_mode: ``'same_step'`` (done envs are re-sampled inside ``step()``; the returned observation is the first
        of the new episode and ``info['final_obs']`` holds the terminal one), ``'next_step'`` (Gymnasium's VectorEnv default:
        the step after a done ignores the action and returns the reset observation, reward 0) or ``'disabled'`` (like the
        single reference env: the caller resets)
    :param precision: ``'f32'`` (float storage and arithmetic; the throughput configuration), ``'f32_f64acc'`` (float
        storage, double arithmetic inside a step) or ``'f64'`` (the reference's dtype; decisions identical to the CPU oracle)
    :param env_id_offset: global index of local env 0; random streams are keyed by global index, so a sharded run reproduces
        the single-device run
    :param alias_desired_goal: return the goal state tensor itself as ``desired_goal`` (no copy per step)
    :param max_reset_trials: bound of the rejection loops of a reset (the reference has none: PL:369, 395 loop until a sample is valid and
        warn every 100 trials); a search that runs into it keeps its last, invalid sample and is COUNTED (``diagnostics()``)
    :param warn_on_reset: ``reset()`` ends with ``warn_diagnostics()`` -- one host synchronisation per explicit reset -- so that such
        searches (also those of the auto-resets since the last call) are reported the way the reference reports them
    """

    reward_success = 50  # PL:220

    def __init__(
        self,
        layout_tiles: np.ndarray,
        num_movers: int,
        show_2D_plot: bool = False,
        mover_colors_2D_plot: list[str] | None = None,
        tile_params: dict[str, Any] | None = None,
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
        threshold_pos: float = 0.1,
        use_mj_passive_viewer: bool = False,
        *,
        num_envs: int = 1,
        device: str | torch.device = 'cuda',
        max_episode_steps: int | None = 50,
        autoreset_mode: str = 'same_step',
        precision: str = 'f32',
        env_id_offset: int = 0,
        max_reset_trials: int = 10000,
        alias_desired_goal: bool = False,
        warn_on_reset: bool = True,
        _host_only: bool = False,
    ) -> None:
        if show_2D_plot and mover_colors_2D_plot is None:
            raise ValueError('Please specify the colors of the movers for the 2D plot.')  # PL:223-224
        if show_2D_plot or render_every_cycle or use_mj_passive_viewer:
            raise ValueError('rendering is out of scope of the GPU engine (show_2D_plot / render_every_cycle / use_mj_passive_viewer)')
        assert precision in _native.PRECISION, f'precision must be one of {list(_native.PRECISION)}'
        assert autoreset_mode in _native.AUTORESET, f'autoreset_mode must be one of {list(_native.AUTORESET)}'
        super().__init__(layout_tiles=layout_tiles, num_movers=num_movers, num_envs=num_envs, tile_params=tile_params,
                         mover_params=mover_params, initial_mover_zpos=initial_mover_zpos, std_noise=std_noise, render_mode=render_mode,
                         collision_params=collision_params, device=device, _host_only=_host_only)
        self.learn_jerk = bool(learn_jerk)
        self.num_cycles = int(num_cycles)
        self.v_max, self.a_max, self.j_max = float(v_max), float(a_max), float(j_max)
        self.threshold_pos = float(threshold_pos)
        self.show_2D_plot = False
        self.max_episode_steps = max_episode_steps
        self.autoreset_mode = autoreset_mode
        self.precision = precision
        self.env_id_offset = int(env_id_offset)
        self.warn_on_reset = bool(warn_on_reset)
        self.max_reset_trials = int(max_reset_trials)
        self.dtype = _TORCH_DTYPE[precision]
        self.metadata = dict(self.metadata, autoreset_mode=spaces.autoreset_metadata(autoreset_mode))

        # spaces (PL:242-259); float64 like the reference, whatever the storage precision
        M = self.num_movers
        low_goals = np.zeros((M * 2,))
        high_goals = np.array([np.max(self.x_pos_tiles) + (self.tile_size[0] / 2), np.max(self.y_pos_tiles) + (self.tile_size[1] / 2)] * M)
        self.obs_dim = M * (1 + int(self.learn_jerk)) * 2
        self.single_observation_space = spaces.Dict({
            'observation': spaces.Box(low=-np.inf, high=np.inf, shape=(self.obs_dim,), dtype=np.float64),
            'achieved_goal': spaces.Box(low=low_goals, high=high_goals, dtype=np.float64),
            'desired_goal': spaces.Box(low=low_goals, high=high_goals, dtype=np.float64),
        })
        lim = self.j_max if self.learn_jerk else self.a_max
        self.single_action_space = spaces.Box(low=-lim, high=lim, shape=(M * 2,), dtype=np.float64)
        self.observation_space = spaces.Dict({k: spaces.batch_box(v, self.num_envs) for k, v in self.single_observation_space.items()})
        self.action_space = spaces.batch_box(self.single_action_space, self.num_envs)

        # minimum and maximum possible mover (x,y)-positions (PL:262-267; tile_size is already a half size there too)
        safety_margin = self.c_size + self.c_size_offset_wall + self.c_size_offset
        self.min_xy_pos = np.zeros(2) + safety_margin
        self.max_xy_pos = (
            np.array([np.max(self.x_pos_tiles) + (self.tile_size[0] / 2), np.max(self.y_pos_tiles) + (self.tile_size[1] / 2)]) - safety_margin
        )
        # minimum distance between any two goals (PL:270-274)
        if self.c_shape == 'circle':
            self.min_goal_dist = 2 * (self.c_size + self.c_size_offset)
            assert np.ndim(self.min_goal_dist) == 0, 'the planning env needs one collision radius for all movers (PL:263,271)'
        else:
            assert np.shape(self.c_size) == (2,), 'the planning env samples starts / goals from one (2,) box size for all movers'
            self.min_goal_dist = 2 * np.linalg.norm(self.c_size + self.c_size_offset, ord=2)

        if self._host_only:
            return
        self._create_handle(env_id_offset=self.env_id_offset, num_cycles=self.num_cycles, v_max=self.v_max, a_max=self.a_max,
                            j_max=self.j_max, learn_jerk=self.learn_jerk, threshold_pos=self.threshold_pos,
                            max_episode_steps=max_episode_steps, autoreset_mode=autoreset_mode, min_xy_pos=self.min_xy_pos,
                            max_xy_pos=self.max_xy_pos, min_goal_dist=float(self.min_goal_dist), max_reset_trials=max_reset_trials,
                            precision=precision)

        # state + output tensors: owned here (PyTorch), handed to the library as raw device pointers
        E, dev, dt = self.num_envs, self.device, self.dtype
        z = lambda *shape, dtype=dt: torch.zeros(*shape, dtype=dtype, device=dev)  # noqa: E731
        self.pos, self.vel, self.acc, self.goals = z(E, 2 * M), z(E, 2 * M), z(E, 2 * M), z(E, 2 * M)
        self.elapsed_steps = z(E, dtype=torch.int32)
        self._obs = OrderedDict(observation=z(E, self.obs_dim), achieved_goal=z(E, 2 * M),
                                desired_goal=self.goals if alias_desired_goal else z(E, 2 * M))
        self._final = OrderedDict(observation=z(E, self.obs_dim), achieved_goal=z(E, 2 * M), desired_goal=z(E, 2 * M))
        self._reward = z(E)
        u8 = lambda: z(E, dtype=torch.uint8)  # noqa: E731
        self._terminated, self._truncated, self._is_success, self._mover_collision, self._wall_collision = u8(), u8(), u8(), u8(), u8()
        state = _native.GprState(self.pos.data_ptr(), self.vel.data_ptr(), self.acc.data_ptr(), self.goals.data_ptr(),
                                 self.elapsed_steps.data_ptr())
        _native.check(self._lib.gpr_bind_state(self._handle, C.byref(state)), self._handle)
        use_final = autoreset_mode == 'same_step'
        self._out = _native.GprStepOut(
            self._obs['observation'].data_ptr(), self._obs['achieved_goal'].data_ptr(), self._obs['desired_goal'].data_ptr(),
            self._reward.data_ptr(), self._terminated.data_ptr(), self._truncated.data_ptr(), self._is_success.data_ptr(),
            self._mover_collision.data_ptr(), self._wall_collision.data_ptr(),
            self._final['observation'].data_ptr() if use_final else None, self._final['achieved_goal'].data_ptr() if use_final else None,
            self._final['desired_goal'].data_ptr() if use_final else None)
        self._needs_seed = True

    def _state_tensors(self) -> dict[str, torch.Tensor]:
        return {'pos': self.pos, 'vel': self.vel, 'acc': self.acc, 'goals': self.goals, 'elapsed_steps': self.elapsed_steps}

    # ------------------------------------------------------------------------------------------------------------
    # Gymnasium-style API, batched
    # ------------------------------------------------------------------------------------------------------------
    def _obs_dict(self) -> OrderedDict:
        return OrderedDict((k, v) for k, v in self._obs.items())

    def _info_dict(self) -> dict[str, torch.Tensor]:
        # uint8 0/1 written by the kernel, re-typed in place (no copy, no launch)
        return {'is_success': self._is_success.view(torch.bool), 'mover_collision': self._mover_collision.view(torch.bool),
                'wall_collision': self._wall_collision.view(torch.bool)}

    def reset(self, seed: int | None = None, options: dict[str, Any] | None = None) -> tuple[OrderedDict, dict[str, torch.Tensor]]:
        """Reset envs (BE:1770-1833 + PL:355-418): rejection-sample collision-free starts and valid goals, zero the velocities,
        return the initial observation and info. ``options={'mask': bool tensor (num_envs,)}`` resets a subset.

        ``options={'candidates': tensor (n_cand, num_envs, num_movers, 2)}`` puts the generator of PL:377 / PL:402 into the caller's
        hands (scripted / curriculum resets): set ``k`` is the ``k``-th sample ``np_random.uniform`` would hand out -- per env the start
        loop consumes sets until one passes its checks, the goal loop goes on from the next one, and a loop that reaches the end of the
        stream keeps the last set. ``info['candidates_used']`` (int32, (num_envs,)) tells how many sets each env consumed."""
        if seed is not None or self._needs_seed:
            _native.check(self._lib.gpr_seed(self._handle, int(seed if seed is not None else np.random.SeedSequence().entropy % (1 << 63))),
                          self._handle)
            self._needs_seed = False
        mask = None
        if options is not None and options.get('mask') is not None:
            mask = torch.as_tensor(options['mask'], device=self.device).to(torch.uint8).contiguous()
            assert mask.shape == (self.num_envs,)
        cand = options.get('candidates') if options is not None else None
        if cand is not None:
            cand = torch.as_tensor(cand, device=self.device).to(self.dtype).reshape(-1, self.num_envs, 2 * self.num_movers).contiguous()
            assert cand.shape[0] >= 2, 'the candidate stream needs at least one start set and one goal set'
            used = torch.zeros(self.num_envs, dtype=torch.int32, device=self.device)
            with torch.cuda.device(self.device):
                _native.check(self._lib.gpr_reset_candidates(self._handle, mask.data_ptr() if mask is not None else None, cand.shape[0],
                                                             cand.data_ptr(), used.data_ptr(), C.byref(self._out), self._stream()), self._handle)
            return self._obs_dict(), dict(self._info_dict(), candidates_used=used)
        with torch.cuda.device(self.device):
            _native.check(self._lib.gpr_reset(self._handle, mask.data_ptr() if mask is not None else None, C.byref(self._out),
                                              self._stream()), self._handle)
        if self.warn_on_reset:
            self.warn_diagnostics()  # PL:371-375, 397-401: a search that found no valid sample is reported (one host synchronisation)
        return self._obs_dict(), self._info_dict()

    def step(self, action: torch.Tensor) -> tuple[OrderedDict, torch.Tensor, torch.Tensor, torch.Tensor, dict[str, torch.Tensor]]:
        """One env-step of every env (BE:1835-1950): the action (num_envs, 2*num_movers) -- accelerations, or jerks when
        ``learn_jerk`` -- is clipped to the action space and applied for ``num_cycles`` control cycles, stopping early at the
        first wall / mover collision. Returns (obs dict, reward, terminated, truncated, info); tensors are reused between calls."""
        if not isinstance(action, torch.Tensor):
            action = torch.as_tensor(np.asarray(action), device=self.device)
        assert tuple(action.shape) == (self.num_envs, 2 * self.num_movers), 'action dim != action_space dim'  # BE:1870
        if action.dtype != self.dtype or action.device != self.device or not action.is_contiguous():
            action = action.to(device=self.device, dtype=self.dtype).contiguous()
        self._step_callback(action)  # BE:1876
        with torch.cuda.device(self.device):
            _native.check(self._lib.gpr_step(self._handle, action.data_ptr(), C.byref(self._out), self._stream()), self._handle)
        info = self._info_dict()
        if self.autoreset_mode == 'same_step':
            info['final_obs'] = self._final  # valid where terminated | truncated
        return self._obs_dict(), self._reward, self._terminated.view(torch.bool), self._truncated.view(torch.bool), info

    # ------------------------------------------------------------------------------------------------------------
    # zero-copy rollout collection: let the kernel write a step's outputs straight into caller-owned tensors
    # ------------------------------------------------------------------------------------------------------------
    def bind_outputs(self, observation: torch.Tensor, achieved_goal: torch.Tensor, desired_goal: torch.Tensor, reward: torch.Tensor,
                     terminated: torch.Tensor, truncated: torch.Tensor, agent_observation: torch.Tensor | None = None,
                     agent_reward: torch.Tensor | None = None) -> _native.GprStepOut:
        """Describe an alternative set of output tensors (e.g. slot ``t`` of a ``distributed.RolloutBuffer``) for ``step_into``.
        The info flags and ``final_obs`` keep going to the env's own tensors. ``agent_observation`` ``[E, M, 4M+2]`` / ``agent_reward``
        ``[E, M]``: the per-agent views of the PettingZoo-parallel layer, written by the step kernel's epilogue (both or neither)."""
        E, M = self.num_envs, self.num_movers
        checks = [(observation, (E, self.obs_dim), self.dtype), (achieved_goal, (E, 2 * M), self.dtype), (desired_goal, (E, 2 * M), self.dtype),
                  (reward, (E,), self.dtype), (terminated, (E,), torch.uint8), (truncated, (E,), torch.uint8)]
        assert (agent_observation is None) == (agent_reward is None), 'bind both agent_observation and agent_reward or neither'
        if agent_observation is not None:
            checks += [(agent_observation, (E, M, 4 * M + 2), self.dtype), (agent_reward, (E, M), self.dtype)]
        for t, shape, dt in checks:
            assert tuple(t.shape) == shape and t.dtype == dt and t.device == self.device and t.is_contiguous(), (tuple(t.shape), t.dtype)
        o = _native.GprStepOut()
        C.memmove(C.byref(o), C.byref(self._out), C.sizeof(o))
        o.observation_dev, o.achieved_goal_dev, o.desired_goal_dev = observation.data_ptr(), achieved_goal.data_ptr(), desired_goal.data_ptr()
        o.reward_dev, o.terminated_dev, o.truncated_dev = reward.data_ptr(), terminated.data_ptr(), truncated.data_ptr()
        if agent_observation is not None:
            o.agent_observation_dev, o.agent_reward_dev = agent_observation.data_ptr(), agent_reward.data_ptr()
        return o

    def step_into(self, action: torch.Tensor, out: _native.GprStepOut) -> None:
        """``step()`` without any host-side bookkeeping: one C-ABI call, outputs land in the tensors ``out`` was bound to.
        ``action`` must already be a contiguous (num_envs, 2*num_movers) tensor of the env's dtype on the env's device."""
        assert action.dtype == self.dtype and action.device == self.device and action.is_contiguous()
        assert tuple(action.shape) == (self.num_envs, 2 * self.num_movers), 'action dim != action_space dim'
        _native.check(self._lib.gpr_step(self._handle, action.data_ptr(), C.byref(out), self._stream()), self._handle)

    # ------------------------------------------------------------------------------------------------------------
    # open-loop rollouts: n env-steps per kernel launch (gpr_step_n)
    # ------------------------------------------------------------------------------------------------------------
    def alloc_rollout(self, n_steps: int, agent_views: bool = False) -> dict[str, torch.Tensor]:
        """[n_steps, num_envs, ...] tensors for everything ``step()`` returns (observation dict, reward, terminated, truncated, the
        three info flags and, with same-step autoreset, the terminal observations), to be filled by ``step_n``. ``agent_views``: also
        the per-agent views of the PettingZoo-parallel layer ([n_steps, num_envs, M, 4M+2] / [n_steps, num_envs, M], include/gpr.h)."""
        E, M, dev, dt, T = self.num_envs, self.num_movers, self.device, self.dtype, int(n_steps)
        z = lambda *shape, dtype=dt: torch.zeros(*shape, dtype=dtype, device=dev)  # noqa: E731
        r = dict(observation=z(T, E, self.obs_dim), achieved_goal=z(T, E, 2 * M), desired_goal=z(T, E, 2 * M), reward=z(T, E))
        for k in ('terminated', 'truncated', 'is_success', 'mover_collision', 'wall_collision'):
            r[k] = z(T, E, dtype=torch.uint8)
        if self.autoreset_mode == 'same_step':
            r.update(final_observation=z(T, E, self.obs_dim), final_achieved_goal=z(T, E, 2 * M), final_desired_goal=z(T, E, 2 * M))
        fin = (lambda k: r[k].data_ptr()) if self.autoreset_mode == 'same_step' else (lambda k: None)  # noqa: E731
        r['_out'] = _native.GprStepOut(r['observation'].data_ptr(), r['achieved_goal'].data_ptr(), r['desired_goal'].data_ptr(), r['reward'].data_ptr(),
                                       r['terminated'].data_ptr(), r['truncated'].data_ptr(), r['is_success'].data_ptr(),
                                       r['mover_collision'].data_ptr(), r['wall_collision'].data_ptr(), fin('final_observation'),
                                       fin('final_achieved_goal'), fin('final_desired_goal'))
        if agent_views:
            r.update(agent_observation=z(T, E, M, 4 * M + 2), agent_reward=z(T, E, M))
            r['_out'].agent_observation_dev, r['_out'].agent_reward_dev = r['agent_observation'].data_ptr(), r['agent_reward'].data_ptr()
        return r

    def step_n(self, actions: torch.Tensor, rollout: dict[str, torch.Tensor]) -> dict[str, torch.Tensor]:
        """``actions.shape[0]`` env-steps in ONE kernel launch (open-loop: random, scripted or replayed actions): row ``i`` of every
        tensor of ``rollout`` (from ``alloc_rollout``) receives what the ``i``-th ``step(actions[i])`` returns, bit for bit. Every
        block of the kernel walks through its envs' steps on its own, so the per-step launch, drain and load burst disappear."""
        T = int(actions.shape[0])
        assert actions.dtype == self.dtype and actions.device == self.device and actions.is_contiguous()
        assert tuple(actions.shape) == (T, self.num_envs, 2 * self.num_movers) and rollout['reward'].shape[0] == T
        with torch.cuda.device(self.device):
            _native.check(self._lib.gpr_step_n(self._handle, T, actions.data_ptr(), C.byref(rollout['_out']), self._stream()), self._handle)
        return rollout

    def compute_reward(self, achieved_goal: torch.Tensor, desired_goal: torch.Tensor, info) -> torch.Tensor:
        """Batched HER reward (PL:502-534). ``info``: a dict of 'mover_collision' / 'wall_collision' tensors of shape (batch,) (what
        ``step()`` returns), or -- as in the reference (PL:666-688) -- an array / list of per-transition info dicts, or one info
        dict of scalars (then a 0-dim tensor is returned, the reference's ``reward[0]``)."""
        return self._reward_and_terminated(achieved_goal, desired_goal, info)[0]

    def compute_terminated(self, achieved_goal: torch.Tensor, desired_goal: torch.Tensor, info: dict[str, torch.Tensor]) -> torch.Tensor:
        """reward in {+50, -50} (PL:459-479)."""
        return self._reward_and_terminated(achieved_goal, desired_goal, info)[1]

    def compute_truncated(self, achieved_goal: torch.Tensor, desired_goal: torch.Tensor, info: dict[str, torch.Tensor] | None = None):
        """Always False: the time limit is applied by the step counter, like the TimeLimit wrapper (PL:481-500)."""
        batch = achieved_goal.shape[0] if achieved_goal.ndim > 1 else 1
        return torch.zeros(batch, dtype=torch.bool, device=self.device)

    def _reward_and_terminated(self, achieved_goal, desired_goal, info):
        M = self.num_movers
        ag = torch.as_tensor(achieved_goal, device=self.device).to(self.dtype).reshape(-1, 2 * M).contiguous()
        dg = torch.as_tensor(desired_goal, device=self.device).to(self.dtype).reshape(-1, 2 * M).contiguous()
        b = ag.shape[0]
        mc, wc, scalar = self._preprocess_info(info)
        assert mc.shape == (b,) and wc.shape == (b,) and dg.shape == ag.shape, 'achieved_goal, desired_goal and info disagree on the batch size'
        rew = torch.empty(b, dtype=self.dtype, device=self.device)
        term = torch.empty(b, dtype=torch.uint8, device=self.device)
        with torch.cuda.device(self.device):
            _native.check(self._lib.gpr_compute_reward(self._handle, b, ag.data_ptr(), dg.data_ptr(), mc.data_ptr(), wc.data_ptr(),
                                                       rew.data_ptr(), term.data_ptr(), self._stream()), self._handle)
        return (rew[0], term.bool()[0]) if scalar else (rew, term.bool())

    def ensure_max_dyn_val(self, current_values, max_value: float, next_derivs) -> tuple[torch.Tensor, torch.Tensor]:
        """``ensure_max_dyn_val`` (PL:610-645), row-wise on the device (f64): returns (next values, derivatives to apply)."""
        cur = self._dev_f64(current_values).reshape(-1, 2)
        der = self._dev_f64(next_derivs).reshape(-1, 2)
        n = cur.shape[0]
        mv = torch.full((n,), float(max_value), dtype=torch.float64, device=self.device)
        vals, ders = torch.empty_like(cur), torch.empty_like(cur)
        with torch.cuda.device(self.device):
            _native.check(self._lib.gpr_ensure_max_dyn_val(n, cur.data_ptr(), mv.data_ptr(), der.data_ptr(), self.cycle_time, vals.data_ptr(),
                                                           ders.data_ptr(), self._stream()))
        return vals, ders

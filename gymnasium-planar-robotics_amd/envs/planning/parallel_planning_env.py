"""``VecBenchmarkPlanningParallelEnv``: a PettingZoo-``ParallelEnv``-shaped view of ``VecBenchmarkPlanningEnv`` -- one agent per
mover, every per-agent array batched over ``num_envs``.

What the reference offers here is ``BasicPlanarRoboticsMultiAgentEnv`` (gymnasium_planar_robotics/envs/basic_envs.py:1629-1693):
an empty PettingZoo shell whose only content is ``agents = possible_agents = mover_names`` (``mover_0`` ...), and which does not
construct at the reference's HEAD. There is therefore **no reference behaviour to match**: the physics, termination and
truncation are exactly those of the single-agent env (one fused kernel launch), and the per-agent observation / reward layout is
this engine's own definition (include/gpr.h, ``gpr_agent_views``):

* ``observations[agent]`` ``[E, 4M+2]``: own velocity, own position, own goal, then for every other mover in ascending index its
  position and velocity relative to the agent -- a lane-permute gather inside the env's wavefront group in the step kernel's own
  epilogue (``gpr_step_out.agent_observation_dev``): ``step()`` is ONE kernel launch, nothing is re-read; only ``reset()`` runs the
  stand-alone gather ``gpr_agent_views``;
* ``rewards[agent]`` ``[E]``: the env reward of PL:519-531 split per agent: -50 / +50 for every agent on collision / joint
  success, otherwise -1 while the agent's own goal is not reached, 0 once it is (summing to the single-agent reward). With
  same-step autoreset the reward of an env that ended in this call (also by TimeLimit) is computed from its TERMINAL
  achieved / desired goal (``info['final_obs']``), not from the first observation of the new episode;
* ``terminations`` / ``truncations`` / ``infos``: the env's, shared by every agent.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ... import _native, spaces
from .benchmark_planning_env import VecBenchmarkPlanningEnv


class VecBenchmarkPlanningParallelEnv(spaces.ParallelEnvBase):
    metadata = {'name': 'VecBenchmarkPlanningParallelEnv', 'render_modes': []}

    def __init__(self, *args, num_sub_batches: int = 1, **kwargs) -> None:
        """Arguments are ``VecBenchmarkPlanningEnv``'s. ``num_sub_batches`` > 1 steps the envs as that many sub-batches on their own HIP
        streams (``VecBenchmarkPlanningEnvSubBatched``: bit-identical results, the tail of one sub-batch's kernel overlaps the next one's
        bulk); the per-agent tensors stay whole ``[E, M, .]`` tensors, every sub-batch's kernel writes its rows."""
        from .sub_batched_env import VecBenchmarkPlanningEnvSubBatched
        self.num_sub_batches = int(num_sub_batches)
        if self.num_sub_batches > 1:
            self.env = VecBenchmarkPlanningEnvSubBatched(*args, num_sub_batches=self.num_sub_batches, **kwargs)
            self.num_sub_batches = self.env.num_sub_batches  # (as many as there are streams that really overlap)
            subs = self.env.subs
        else:
            self.env = VecBenchmarkPlanningEnv(*args, **kwargs)
            subs = [self.env]
        e, e0 = self.env, subs[0]
        self.possible_agents = list(e0.mover_names)  # BE:1692-1693
        self.agents = list(self.possible_agents)
        self.num_envs, self.num_movers = e.num_envs, e0.num_movers
        M = self.num_movers
        self.agent_obs_dim = 4 * M + 2
        lim = e0.j_max if e0.learn_jerk else e0.a_max
        self._obs_space = spaces.Box(-np.inf, np.inf, shape=(self.agent_obs_dim,), dtype=np.float64)
        self._act_space = spaces.Box(-lim, lim, shape=(2,), dtype=np.float64)
        if not e0._host_only:
            self._agent_obs = torch.zeros(e.num_envs, M, self.agent_obs_dim, dtype=e0.dtype, device=e0.device)
            self._agent_reward = torch.zeros(e.num_envs, M, dtype=e0.dtype, device=e0.device)
            self._action = torch.zeros(e.num_envs, 2 * M, dtype=e0.dtype, device=e0.device)
            # the per-agent dict entries are fixed views of the two output tensors: built once
            self._obs_views = {a: self._agent_obs[:, i, :] for i, a in enumerate(self.agents)}
            self._rew_views = {a: self._agent_reward[:, i] for i, a in enumerate(self.agents)}
            # step(): the step kernel writes the views itself; reset(): the stand-alone gather over the stored outputs
            self._out_reset, n0 = [], 0
            for sb in subs:
                o = _native.GprStepOut()
                C.memmove(C.byref(o), C.byref(sb._out), C.sizeof(o))
                self._out_reset.append(o)
                rows_obs, rows_rew = self._agent_obs[n0:n0 + sb.num_envs], self._agent_reward[n0:n0 + sb.num_envs]
                sb._out.agent_observation_dev, sb._out.agent_reward_dev = rows_obs.data_ptr(), rows_rew.data_ptr()
                n0 += sb.num_envs
            self._subs = subs

    def observation_space(self, agent: str) -> spaces.Box:
        return self._obs_space

    def action_space(self, agent: str) -> spaces.Box:
        return self._act_space

    def _views(self):
        """the stand-alone gather (one more launch per sub-batch) over the outputs of the last reset() / step()"""
        n0 = 0
        for sb, o in zip(self._subs, self._out_reset):
            with torch.cuda.device(sb.device):
                _native.check(sb._lib.gpr_agent_views(sb._handle, C.byref(o), self._agent_obs[n0:n0 + sb.num_envs].data_ptr(),
                                                      self._agent_reward[n0:n0 + sb.num_envs].data_ptr(), sb._stream()), sb._handle)
            n0 += sb.num_envs
        if self.num_sub_batches > 1:
            self.env.wait_for_caller()  # (the gathers ran on the caller's stream; the next step kernels overwrite what they read)
        return dict(self._obs_views), dict(self._rew_views)

    def reset(self, seed: int | None = None, options: dict | None = None):
        _, info = self.env.reset(seed=seed, options=options)
        obs, _ = self._views()
        return obs, {a: info for a in self.agents}

    def step(self, actions: dict[str, torch.Tensor] | torch.Tensor):
        """``actions``: dict agent -> ``[E, 2]`` tensor (every agent must act: the env is fully cooperative and simultaneous),
        or one ``[E, M, 2]`` / ``[E, 2M]`` tensor."""
        e, dev, dt = self.env, self._action.device, self._action.dtype
        if isinstance(actions, dict):
            assert set(actions) == set(self.agents), 'every agent must supply an action'
            parts = [torch.as_tensor(actions[a], device=dev) for a in self.agents]
            act = self._action.view(self.num_envs, self.num_movers, 2)
            if all(t.dtype == dt and tuple(t.shape) == (self.num_envs, 2) for t in parts):
                torch.stack(parts, dim=1, out=act)  # one kernel for all agents
            else:
                act.copy_(torch.stack([t.reshape(self.num_envs, 2) for t in parts], dim=1))
            action = self._action
        else:
            action = torch.as_tensor(actions, device=dev).to(dt).reshape(self.num_envs, 2 * self.num_movers)
        _, _, terminated, truncated, info = e.step(action)  # (one launch per sub-batch: the views are written by the step kernel)
        obs, rew = dict(self._obs_views), dict(self._rew_views)
        return obs, rew, {a: terminated for a in self.agents}, {a: truncated for a in self.agents}, {a: info for a in self.agents}

    # zero-copy rollouts (bench.py): outputs incl. the per-agent views bound into caller-owned tensors, ONE C-ABI call per env-step
    def bind_outputs(self, **tensors: torch.Tensor):
        """``VecBenchmarkPlanningEnv.bind_outputs`` / ``VecBenchmarkPlanningEnvSubBatched.bind_outputs``; without ``agent_observation`` /
        ``agent_reward`` the views go to this object's own tensors"""
        tensors.setdefault('agent_observation', self._agent_obs)
        tensors.setdefault('agent_reward', self._agent_reward)
        return self.env.bind_outputs(**tensors)

    def state(self) -> torch.Tensor:
        """global state ``[E, 6M]``: velocities, positions, goals (the single-agent observation dict, concatenated)"""
        if self.num_sub_batches > 1:
            self.env.synchronize()  # (the caller's stream waits for the sub-batches' kernels: their outputs are read below)
        return torch.cat([torch.cat([sb._obs['observation'], sb._obs['achieved_goal'], sb._obs['desired_goal']], dim=1) for sb in self._subs], dim=0)

    def close(self) -> None:
        self.env.close()

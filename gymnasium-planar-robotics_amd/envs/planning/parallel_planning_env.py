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

    def __init__(self, *args, **kwargs) -> None:
        self.env = VecBenchmarkPlanningEnv(*args, **kwargs)
        e = self.env
        self.possible_agents = list(e.mover_names)  # BE:1692-1693
        self.agents = list(self.possible_agents)
        self.num_envs, self.num_movers = e.num_envs, e.num_movers
        M = self.num_movers
        self.agent_obs_dim = 4 * M + 2
        lim = e.j_max if e.learn_jerk else e.a_max
        self._obs_space = spaces.Box(-np.inf, np.inf, shape=(self.agent_obs_dim,), dtype=np.float64)
        self._act_space = spaces.Box(-lim, lim, shape=(2,), dtype=np.float64)
        if not e._host_only:
            self._agent_obs = torch.zeros(e.num_envs, M, self.agent_obs_dim, dtype=e.dtype, device=e.device)
            self._agent_reward = torch.zeros(e.num_envs, M, dtype=e.dtype, device=e.device)
            self._action = torch.zeros(e.num_envs, 2 * M, dtype=e.dtype, device=e.device)
            # the per-agent dict entries are fixed views of the gather kernel's two output tensors: built once
            self._obs_views = {a: self._agent_obs[:, i, :] for i, a in enumerate(self.agents)}
            self._rew_views = {a: self._agent_reward[:, i] for i, a in enumerate(self.agents)}
            # step(): the step kernel writes the views itself; reset(): the stand-alone gather over the stored outputs
            self._out_reset = _native.GprStepOut()
            C.memmove(C.byref(self._out_reset), C.byref(e._out), C.sizeof(self._out_reset))
            e._out.agent_observation_dev, e._out.agent_reward_dev = self._agent_obs.data_ptr(), self._agent_reward.data_ptr()

    def observation_space(self, agent: str) -> spaces.Box:
        return self._obs_space

    def action_space(self, agent: str) -> spaces.Box:
        return self._act_space

    def _views(self):
        """the stand-alone gather (one more launch) over the outputs of the last reset() / step()"""
        e = self.env
        with torch.cuda.device(e.device):
            _native.check(e._lib.gpr_agent_views(e._handle, C.byref(self._out_reset), self._agent_obs.data_ptr(), self._agent_reward.data_ptr(),
                                                 e._stream()), e._handle)
        return dict(self._obs_views), dict(self._rew_views)

    def reset(self, seed: int | None = None, options: dict | None = None):
        _, info = self.env.reset(seed=seed, options=options)
        obs, _ = self._views()
        return obs, {a: info for a in self.agents}

    def step(self, actions: dict[str, torch.Tensor] | torch.Tensor):
        """``actions``: dict agent -> ``[E, 2]`` tensor (every agent must act: the env is fully cooperative and simultaneous),
        or one ``[E, M, 2]`` / ``[E, 2M]`` tensor."""
        e = self.env
        if isinstance(actions, dict):
            assert set(actions) == set(self.agents), 'every agent must supply an action'
            parts = [torch.as_tensor(actions[a], device=e.device) for a in self.agents]
            act = self._action.view(e.num_envs, self.num_movers, 2)
            if all(t.dtype == e.dtype and tuple(t.shape) == (e.num_envs, 2) for t in parts):
                torch.stack(parts, dim=1, out=act)  # one kernel for all agents
            else:
                act.copy_(torch.stack([t.reshape(e.num_envs, 2) for t in parts], dim=1))
            action = self._action
        else:
            action = torch.as_tensor(actions, device=e.device).to(e.dtype).reshape(e.num_envs, 2 * self.num_movers)
        _, _, terminated, truncated, info = e.step(action)  # (one launch: the views are written by the step kernel)
        obs, rew = dict(self._obs_views), dict(self._rew_views)
        return obs, rew, {a: terminated for a in self.agents}, {a: truncated for a in self.agents}, {a: info for a in self.agents}

    def state(self) -> torch.Tensor:
        """global state ``[E, 6M]``: velocities, positions, goals (the single-agent observation dict, concatenated)"""
        o = self.env._obs
        return torch.cat([o['observation'], o['achieved_goal'], o['desired_goal']], dim=1)

    def close(self) -> None:
        self.env.close()

"""``VecBenchmarkPlanningEnvSubBatched``: one env batch stepped as K independent sub-batches, each on its own HIP stream.

Why. ``step()`` of the reference (gymnasium_planar_robotics/envs/basic_envs.py:1835-1950) is one env at a time; the engine's is one
kernel launch for the whole batch, and that kernel lasts as long as its slowest wavefront (one whose movers clamp in all 40 cycles,
DESIGN.md section 6) while most of the SIMDs have already run out of work -- nothing back-fills a launch's tail except another
launch. Envs are independent (BE:1876-1905 touches one env's state only), so the batch can be cut into K contiguous sub-batches with
one handle (``env_id_offset`` = the sub-batch's first global env id) and one stream each: the tail of one sub-batch's kernel then
overlaps with the bulk of another's. Random streams are keyed by GLOBAL env id, so the K sub-batches return bit for bit what the one
big batch returns (``tests/test_gpu_parity.py::test_sub_batches_equal_the_whole_batch``).

This is legal with a policy in the loop: every sub-batch is its own closed loop (observation -> policy forward -> action -> step) on
its own stream, exactly like K smaller vector envs -- which is what they are.

Two ways to drive it:

* ``step_async(k, action_k)`` / ``step_wait(k)``: events order sub-batch k's kernel after whatever the CALLER's current stream did to
  produce ``action_k``, and the caller's stream after the kernel. The policy may live on one stream.
* ``step_all_into(actions, outs)``: ONE C-ABI call (``gpr_step_many``) launches all K kernels, each on its sub-batch's stream, with no
  cross-stream events: for callers that run sub-batch k's policy on ``streams[k]`` too (stream order is then all the ordering
  there is), and for open-loop action rings -- ``bench.py``.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Any

import numpy as np
import torch

from ... import _native
from .benchmark_planning_env import VecBenchmarkPlanningEnv


class VecBenchmarkPlanningEnvSubBatched:
    """``num_envs`` envs as (up to) ``num_sub_batches`` ``VecBenchmarkPlanningEnv`` objects over contiguous env ranges -- as many as there
    are streams whose kernels really overlap on this device (``num_sub_batches`` afterwards: what it became). Keywords are
    ``VecBenchmarkPlanningEnv``'s; ``env_id_offset`` is the global id of this object's first env."""

    def __init__(self, layout_tiles: np.ndarray, num_movers: int, *args, num_envs: int, num_sub_batches: int = 2, env_id_offset: int = 0,
                 device: str | torch.device = 'cuda', **kwargs) -> None:
        K = int(num_sub_batches)
        assert K >= 1 and num_envs % K == 0, 'num_envs must be a multiple of num_sub_batches (equal sub-batches)'
        self.device = torch.device(device)
        self._host_only = bool(kwargs.get('_host_only', False))
        self.stream_probe = {'requested': K, 'tried': 0, 'rejected': 0}
        if not self._host_only:
            if self.device.type != 'cuda':
                raise RuntimeError('the planar-robotics engine runs on MI355X only (device must be a CUDA/ROCm device); there is no CPU path')
            if self.device.index is None:
                self.device = torch.device('cuda', torch.cuda.current_device())
            self._lib = _native.load()
            with torch.cuda.device(self.device):
                # streams first: sub-batches only pay when their kernels really run side by side, so the batch is cut into as many
                # sub-batches as there are streams that do (a power of two, at most the number asked for)
                self.streams = self._pick_streams(K)
            while K > 1 and (K > len(self.streams) or num_envs % K):
                K //= 2
            self.streams = self.streams[:K]
        self.num_envs, self.num_sub_batches, self.sub_envs = int(num_envs), K, int(num_envs) // K
        self.env_id_offset = int(env_id_offset)
        self.subs = [VecBenchmarkPlanningEnv(layout_tiles, num_movers, *args, num_envs=self.sub_envs, env_id_offset=self.env_id_offset + k * self.sub_envs,
                                             device=self.device if not self._host_only else device, **kwargs) for k in range(K)]
        s0 = self.subs[0]
        self.dtype, self.num_movers, self.obs_dim = s0.dtype, s0.num_movers, s0.obs_dim
        self.a_max, self.j_max, self.autoreset_mode = s0.a_max, s0.j_max, s0.autoreset_mode
        self.single_observation_space, self.single_action_space = s0.single_observation_space, s0.single_action_space
        if self._host_only:
            return
        with torch.cuda.device(self.device):
            self._ready = [torch.cuda.Event() for _ in range(K)]  # the caller's stream has produced sub-batch k's action
            self._done = [torch.cuda.Event() for _ in range(K)]   # sub-batch k's step kernel is through
        self._handles = (C.c_void_p * K)(*[s._handle for s in self.subs])
        # thin wavefronts (include/gpr.h: gpr_set_wave_budget): the K handles step side by side, so each may spread a small sub-batch over
        # its K-th of the wavefronts one handle alone would use
        import os
        if K > 1 and not os.environ.get('GPR_THIN_WAVES'):
            lpe = 1 << max(0, int(self.num_movers) - 1).bit_length()
            for s in self.subs:
                _native.check(self._lib.gpr_set_wave_budget(s._handle, 1024 * (4 if lpe >= 16 else 2 if lpe >= 8 else 1) // K), s._handle)
        self._stream_ptrs = (C.c_void_p * K)(*[st.cuda_stream for st in self.streams])
        self._own_outs = (C.POINTER(_native.GprStepOut) * K)(*[C.pointer(s._out) for s in self.subs])

    def _pick_streams(self, K: int) -> list:
        """K streams whose kernels really run side by side. The HIP runtime maps streams onto a few hardware queues (four by default) and
        two streams on one queue execute strictly one after the other (measured: the same two-sub-batch step 38 us on one pair of pool
        streams, 70 us on another). So every candidate is probed against the streams already chosen (``gpr_streams_overlap``: pairs of
        150-us single-wavefront kernels) and skipped if it shares a queue with one of them."""
        import os
        if os.environ.get('GPR_NO_STREAM_PROBE') == '1':  # (profiling runs under a counter-collecting tool, which serialises every launch)
            return [torch.cuda.Stream(device=self.device) for _ in range(K)]
        chosen, tried = [], 0
        while len(chosen) < K and tried < 8 * K:
            cand = torch.cuda.Stream(device=self.device)
            tried += 1
            ok = all(st.cuda_stream != cand.cuda_stream for st in chosen)  # (torch hands its pool's streams out again and again)
            for st in (chosen if ok else []):
                flag = C.c_int32(0)
                _native.check(self._lib.gpr_streams_overlap(st.cuda_stream, cand.cuda_stream, C.byref(flag)))
                ok = ok and bool(flag.value)
            if ok:
                chosen.append(cand)
            else:
                self.stream_probe['rejected'] += 1
        self.stream_probe['tried'] = tried
        return chosen  # (fewer than K: fewer hardware queues than sub-batches asked for -- the caller cuts the batch into fewer)

    # ------------------------------------------------------------------------------------------------------------
    def sub(self, k: int) -> VecBenchmarkPlanningEnv:
        return self.subs[k]

    def sub_slice(self, k: int) -> slice:
        """rows of a ``[num_envs, ...]`` tensor that belong to sub-batch k"""
        return slice(k * self.sub_envs, (k + 1) * self.sub_envs)

    def _cat(self, dicts):
        return OrderedDict((key, torch.cat([d[key] for d in dicts], dim=0)) for key in dicts[0])

    def reset(self, seed: int | None = None, options: dict[str, Any] | None = None):
        """reset() of every sub-batch on the caller's stream (same seed: the streams are keyed by global env id), ordered against the
        sub-batches' own streams on both sides; returns the concatenated observation / info. ``options['mask']`` / ``options['candidates']`` are split along the env axis."""
        self.synchronize()
        obs, infos = [], []
        for k, s in enumerate(self.subs):
            opt = None
            if options is not None:
                opt = {}
                if options.get('mask') is not None:
                    opt['mask'] = torch.as_tensor(options['mask'])[self.sub_slice(k)]
                if options.get('candidates') is not None:
                    c = torch.as_tensor(options['candidates'])
                    opt['candidates'] = c.reshape(c.shape[0], self.num_envs, -1)[:, self.sub_slice(k)]
            o, i = s.reset(seed=seed, options=opt)
            obs.append(o); infos.append(i)
        # the reset kernels ran on the CALLER's stream; the next step_all_into launches on streams[k] with no event of its own, so
        # every sub-batch's stream is made to wait for them here (a step kernel must not read state a reset is still writing)
        self.wait_for_caller()
        return self._cat(obs), {key: torch.cat([i[key] for i in infos], dim=0) for key in infos[0]}

    # ------------------------------------------------------------------------------------------------------------
    def step_async(self, k: int, action: torch.Tensor) -> None:
        """enqueue sub-batch k's env-step on its own stream, behind everything the caller's current stream has enqueued so far
        (which is what produced ``action``); returns at once"""
        s = self.subs[k]
        assert action.dtype == s.dtype and action.device == s.device and action.is_contiguous()
        assert tuple(action.shape) == (self.sub_envs, 2 * self.num_movers), 'action dim != action_space dim'
        self._ready[k].record(torch.cuda.current_stream(self.device))
        self.streams[k].wait_event(self._ready[k])
        _native.check(self._lib.gpr_step(s._handle, action.data_ptr(), C.byref(s._out), self.streams[k].cuda_stream), s._handle)
        self._done[k].record(self.streams[k])

    def step_wait(self, k: int):
        """make the caller's current stream wait for sub-batch k's step (no host synchronisation) and return its
        (obs, reward, terminated, truncated, info) tensors -- ``VecBenchmarkPlanningEnv.step``'s, for ``sub_envs`` envs"""
        torch.cuda.current_stream(self.device).wait_event(self._done[k])
        s = self.subs[k]
        info = s._info_dict()
        if s.autoreset_mode == 'same_step':
            info['final_obs'] = s._final
        return s._obs_dict(), s._reward, s._terminated.view(torch.bool), s._truncated.view(torch.bool), info

    def step(self, action: torch.Tensor):
        """``VecBenchmarkPlanningEnv.step`` for the whole batch (concatenates the sub-batches' results: a convenience, the copies cost
        more than a step -- loops that care keep the sub-batches apart)"""
        if not isinstance(action, torch.Tensor):
            action = torch.as_tensor(np.asarray(action), device=self.device)
        action = action.to(device=self.device, dtype=self.dtype)
        assert tuple(action.shape) == (self.num_envs, 2 * self.num_movers), 'action dim != action_space dim'
        for k in range(self.num_sub_batches):
            self.step_async(k, action[self.sub_slice(k)].contiguous())
        res = [self.step_wait(k) for k in range(self.num_sub_batches)]
        info = {key: torch.cat([r[4][key] for r in res], dim=0) for key in ('is_success', 'mover_collision', 'wall_collision')}
        if self.autoreset_mode == 'same_step':
            info['final_obs'] = self._cat([r[4]['final_obs'] for r in res])
        return (self._cat([r[0] for r in res]), torch.cat([r[1] for r in res]), torch.cat([r[2] for r in res]), torch.cat([r[3] for r in res]), info)

    # ------------------------------------------------------------------------------------------------------------
    def bind_outputs(self, **tensors: torch.Tensor):
        """``VecBenchmarkPlanningEnv.bind_outputs`` for ``[num_envs, ...]`` tensors: one descriptor per sub-batch over its rows, packed for
        ``step_all_into``"""
        outs = [s.bind_outputs(**{key: t[self.sub_slice(k)] for key, t in tensors.items()}) for k, s in enumerate(self.subs)]
        arr = (C.POINTER(_native.GprStepOut) * self.num_sub_batches)(*[C.pointer(o) for o in outs])
        arr._keep = outs
        return arr

    def pack_actions(self, action: torch.Tensor):
        """the K row blocks of a contiguous ``[num_envs, 2M]`` action tensor as a pointer array for ``step_all_into``"""
        assert action.dtype == self.dtype and action.device == self.device and action.is_contiguous()
        assert tuple(action.shape) == (self.num_envs, 2 * self.num_movers), 'action dim != action_space dim'
        step_bytes = self.sub_envs * 2 * self.num_movers * action.element_size()
        return (C.c_void_p * self.num_sub_batches)(*[action.data_ptr() + k * step_bytes for k in range(self.num_sub_batches)])

    def step_all_into(self, action_ptrs, outs=None) -> None:
        """ONE call, K launches: sub-batch k steps with ``action_ptrs[k]`` into ``outs[k]`` (default: its own output tensors) on
        ``streams[k]``. No cross-stream events: whoever produced the actions must have done so on ``streams[k]`` (or have synchronised)."""
        _native.check(self._lib.gpr_step_many(self.num_sub_batches, self._handles, action_ptrs, outs if outs is not None else self._own_outs,
                                              self._stream_ptrs))

    def synchronize(self) -> None:
        """the caller's current stream waits for every sub-batch's stream (no host synchronisation)"""
        cur = torch.cuda.current_stream(self.device)
        for k, st in enumerate(self.streams):
            self._done[k].record(st)
            cur.wait_event(self._done[k])

    def wait_for_caller(self) -> None:
        """every sub-batch's stream waits for what the caller's current stream has enqueued so far (e.g. a collective that reads the
        tensors the next steps will overwrite)"""
        cur = torch.cuda.current_stream(self.device)
        for k, st in enumerate(self.streams):
            self._ready[k].record(cur)
            st.wait_event(self._ready[k])

    # ------------------------------------------------------------------------------------------------------------
    def enable_episode_counters(self, enable: bool = True) -> None:
        for s in self.subs:
            s.enable_episode_counters(enable)

    def read_episode_counters(self) -> torch.Tensor:
        self.synchronize()
        return torch.stack([s.read_episode_counters() for s in self.subs]).sum(dim=0)

    def all_reduce_episode_counters(self) -> torch.Tensor:
        from ... import distributed
        if not all(getattr(s, '_counters_on', False) for s in self.subs):
            self.synchronize()
            self.enable_episode_counters()
        return distributed.all_reduce_sum(self.read_episode_counters())

    # the reference's warnings (counted on the device, see VecBasicPlanarRoboticsEnv) and its state getters, over all sub-batches
    def count_clipped_actions(self, enable: bool = True) -> None:
        for s in self.subs:
            s.count_clipped_actions(enable)

    def diagnostics(self) -> dict[str, int]:
        self.synchronize()
        ds = [s.diagnostics() for s in self.subs]
        return {k: sum(d[k] for d in ds) for k in ds[0]}

    def warn_diagnostics(self) -> dict[str, int]:
        self.synchronize()
        ds = [s.warn_diagnostics() for s in self.subs]
        return {k: sum(d[k] for d in ds) for k in ds[0]}

    def _getter(self, name: str, *args, **kw) -> torch.Tensor:
        self.synchronize()
        out = torch.cat([getattr(s, name)(*args, **kw) for s in self.subs], dim=0)
        self.wait_for_caller()  # (the gathers ran on the caller's stream; the next step kernels rewrite what they read)
        return out

    def get_mover_qpos(self, mover_name: str, add_noise: bool = False) -> torch.Tensor:
        return self._getter('get_mover_qpos', mover_name, add_noise)

    def get_mover_qvel(self, mover_name: str, add_noise: bool = False) -> torch.Tensor:
        return self._getter('get_mover_qvel', mover_name, add_noise)

    def get_mover_qacc(self, mover_name: str, add_noise: bool = False) -> torch.Tensor:
        return self._getter('get_mover_qacc', mover_name, add_noise)

    def get_mover_qpos_arr(self, mover_names: list[str], add_noise: bool = False) -> torch.Tensor:
        return self._getter('get_mover_qpos_arr', mover_names, add_noise)

    def get_mover_qvel_arr(self, mover_names: list[str], add_noise: bool = False) -> torch.Tensor:
        return self._getter('get_mover_qvel_arr', mover_names, add_noise)

    def get_mover_qacc_arr(self, mover_names: list[str], add_noise: bool = False) -> torch.Tensor:
        return self._getter('get_mover_qacc_arr', mover_names, add_noise)

    def state_dict(self) -> dict[str, Any]:
        self.synchronize()
        return {'subs': [s.state_dict() for s in self.subs]}

    def load_state_dict(self, sd: dict[str, Any]) -> None:
        self.synchronize()
        for s, d in zip(self.subs, sd['subs']):
            s.load_state_dict(d)
        self.wait_for_caller()  # (the copies ran on the caller's stream: the sub-batches' streams must see them)

    def close(self) -> None:
        if not self._host_only and torch.cuda.is_available():
            torch.cuda.synchronize(self.device)
        for s in self.subs:
            s.close()

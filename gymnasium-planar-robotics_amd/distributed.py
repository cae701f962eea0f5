"""Multi-GPU plumbing: env batches shard embarrassingly (the reference is literally one env per process): ``step()`` and
``reset()`` exchange nothing. What a rollout flush moves is the caller's choice:

* nothing but the eight episode counters (``env.all_reduce_episode_counters()``: ONE 64-byte all_reduce) when every rank's learner
  consumes its own ``RolloutBuffer`` shard -- the default of ``bench.py``; at this engine's step rate a rank produces
  ~134 B x 3.7e9 env-steps/s = 500 GB/s of rollout data, more than the xGMI links of a GPU can take in from seven peers;
* a learner-sized minibatch (``RolloutBuffer.gather_minibatch``): ONE all-gather of [T, n, .] slices of the fields a learner reads;
* the whole buffer (``RolloutBuffer.flush``): ONE all-gather of every rank's bytes, for small batches.

One process per GPU (``python -m torch.distributed.run``), ``torch.distributed`` backend ``"nccl"`` (= RCCL on ROCm, over xGMI);
the CPU tests drive the same code over ``gloo``. Random streams are keyed by GLOBAL env id (``env_id_offset``), so a sharded
run reproduces the single-device run env for env.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import torch
import torch.distributed as dist


def shard_range(num_envs_global: int, rank: int, world_size: int) -> tuple[int, int]:
    """Contiguous env range [start, stop) owned by ``rank``; the first ``num_envs_global % world_size`` ranks own one more."""
    assert 0 <= rank < world_size and num_envs_global >= world_size
    base, rem = divmod(num_envs_global, world_size)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def init_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """Initialise torch.distributed from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (as set by torch.distributed.run).
    Returns (rank, local_rank, world_size); a single process needs no process group."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world


def make_sharded(env_cls, *args, num_envs_global: int, rank: int | None = None, world_size: int | None = None, **kwargs):
    """This rank's shard of a ``num_envs_global``-env batch: ``env_cls(*args, num_envs=<shard size>, env_id_offset=<shard start>,
    device=cuda:<LOCAL_RANK>, **kwargs)``. Random streams are keyed by global env id, so the shards of any world size reproduce the
    single-device run env for env; there is nothing to exchange between shards in ``step()`` / ``reset()``."""
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else int(os.environ.get('RANK', '0'))
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else int(os.environ.get('WORLD_SIZE', '1'))
    start, stop = shard_range(num_envs_global, rank, world_size)
    kwargs.setdefault('device', torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0'))) if torch.cuda.is_available() else 'cpu')
    return env_cls(*args, num_envs=stop - start, env_id_offset=start, **kwargs)


def _all_gather_bytes(recv: torch.Tensor, send: torch.Tensor) -> None:
    """ONE all_gather_into_tensor; RCCL ("nccl") moves device memory directly, the gloo rehearsal backend goes through the host"""
    if dist.get_backend() == 'gloo' and send.is_cuda:
        h_send = send.cpu()
        h_recv = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_gather_into_tensor(h_recv, h_send)
        recv.copy_(h_recv)
    else:
        dist.all_gather_into_tensor(recv, send)


def all_reduce_sum(t: torch.Tensor) -> torch.Tensor:
    """ONE all_reduce (SUM) of a small tensor over the default process group -- the episode counters' collective. RCCL reduces device
    memory in place; the gloo rehearsal backend goes through the host. Returns the reduced tensor on ``t``'s device (``t`` itself when
    there is nothing to reduce with)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return t
    if dist.get_backend() == 'gloo' and t.is_cuda:
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        return host.to(t.device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


@dataclass
class _Field:
    name: str
    shape: tuple
    dtype: torch.dtype
    offset: int
    nbytes: int


class RolloutBuffer:
    """[T, E_local, ...] rollout tensors of one rank, carved out of ONE byte buffer so that a flush is a single collective.

    Fields: observation [T,E,obs_dim], achieved_goal / desired_goal / action [T,E,2M], reward [T,E] (float dtype), terminated /
    truncated [T,E] (uint8). ``env.step`` can write straight into slot ``t`` (see ``slot``): no copy between the kernel and
    the buffer. ``flush()`` all-gathers the raw bytes of every rank (equal shard sizes) and returns per-field views
    [world, T, E_local, ...].
    """

    def __init__(self, horizon: int, num_envs_local: int, num_movers: int, obs_dim: int, dtype: torch.dtype = torch.float32,
                 device: str | torch.device = 'cuda'):
        self.T, self.E, self.M, self.obs_dim = int(horizon), int(num_envs_local), int(num_movers), int(obs_dim)
        self.dtype, self.device = dtype, torch.device(device)
        specs = [('observation', (obs_dim,), dtype), ('achieved_goal', (2 * num_movers,), dtype), ('desired_goal', (2 * num_movers,), dtype),
                 ('action', (2 * num_movers,), dtype), ('reward', (), dtype), ('terminated', (), torch.uint8), ('truncated', (), torch.uint8)]
        self.fields: list[_Field] = []
        off = 0
        for name, tail, dt in specs:
            shape = (self.T, self.E) + tuple(tail)
            nbytes = int(torch.empty((), dtype=dt).element_size())
            for s in shape:
                nbytes *= s
            self.fields.append(_Field(name, shape, dt, off, nbytes))
            off = (off + nbytes + 255) // 256 * 256  # keep every field 256-byte aligned
        self.nbytes = off
        self.storage = torch.zeros(self.nbytes, dtype=torch.uint8, device=self.device)
        self.views = {f.name: self._view(self.storage, f) for f in self.fields}
        self._gathered = None
        self._mb = {}
        self.t = 0
        self._check_equal_shards()

    def _check_equal_shards(self) -> None:
        """all_gather_into_tensor needs the same byte count from every rank: unequal shards (num_envs_global % world_size != 0
        with ``shard_range``) would hang or fail inside the collective, so they are refused here"""
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return
        on_gpu = dist.get_backend() != 'gloo'
        n = torch.tensor([self.E, -self.E, self.nbytes, -self.nbytes], dtype=torch.int64, device=self.device if on_gpu else 'cpu')
        dist.all_reduce(n, op=dist.ReduceOp.MAX)
        if int(n[0]) != -int(n[1]) or int(n[2]) != -int(n[3]):
            raise ValueError(f'RolloutBuffer: ranks hold different shard sizes ({-int(n[1])}..{int(n[0])} envs); use a num_envs_global '
                             'that is a multiple of the world size')

    @staticmethod
    def _view(storage: torch.Tensor, f: _Field, lead: tuple = ()) -> torch.Tensor:
        flat = storage[..., f.offset:f.offset + f.nbytes]
        return flat.view(f.dtype).reshape(lead + f.shape)

    def __getitem__(self, name: str) -> torch.Tensor:
        return self.views[name]

    def slot(self, t: int) -> dict[str, torch.Tensor]:
        """Views of time slot ``t`` (contiguous [E_local, ...] tensors a kernel can write into)."""
        return {k: v[t] for k, v in self.views.items()}

    def gather_minibatch(self, num_envs: int, start: int = 0, fields=('observation', 'achieved_goal', 'desired_goal', 'action', 'reward',
                                                                     'terminated', 'truncated')) -> dict[str, torch.Tensor]:
        """ONE all-gather of the [T, num_envs, .] slice (local envs start .. start + num_envs) of the named fields: what a learner
        that samples ``world * num_envs`` envs per update reads. Returns per-field views [world, T, num_envs, ...]."""
        num_envs, start = int(num_envs), int(start)
        if not (num_envs >= 1 and 0 <= start and start + num_envs <= self.E):
            raise ValueError(f'gather_minibatch: envs [{start}, {start + num_envs}) are not inside this rank\'s shard of {self.E} envs')
        key = (num_envs, tuple(fields))
        if key not in self._mb:
            sizes = [int(self.views[f][:, :num_envs].numel() * self.views[f].element_size()) for f in fields]
            offs = [0]
            for sz in sizes:
                offs.append((offs[-1] + sz + 15) // 16 * 16)
            world = dist.get_world_size() if dist.is_initialized() else 1
            self._mb[key] = (offs, torch.empty(offs[-1], dtype=torch.uint8, device=self.device),
                             torch.empty(world * offs[-1], dtype=torch.uint8, device=self.device))
        offs, send, recv = self._mb[key]
        for f, o in zip(fields, offs):
            v = self.views[f][:, start:start + num_envs]
            send[o:o + v.numel() * v.element_size()].view(v.dtype).view(v.shape).copy_(v)
        world = 1
        if dist.is_initialized() and dist.get_world_size() > 1:
            world = dist.get_world_size()
            _all_gather_bytes(recv, send)
        else:
            recv.copy_(send)
        g = recv.view(world, offs[-1])
        out = {}
        for f, o in zip(fields, offs):
            v = self.views[f][:, :num_envs]
            out[f] = g[:, o:o + v.numel() * v.element_size()].view(v.dtype).view((world,) + tuple(v.shape))
        return out

    def flush(self) -> dict[str, torch.Tensor]:
        """Single all-gather of the whole buffer over the default process group (no-op gather for one process)."""
        self.t = 0
        if not dist.is_initialized() or dist.get_world_size() == 1:
            return {k: v.unsqueeze(0) for k, v in self.views.items()}
        world = dist.get_world_size()
        if self._gathered is None:
            self._gathered = torch.empty(world * self.nbytes, dtype=torch.uint8, device=self.device)
        _all_gather_bytes(self._gathered, self.storage)
        g = self._gathered.view(world, self.nbytes)
        return {f.name: self._view(g, f, lead=(world,)) for f in self.fields}

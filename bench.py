#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the vectorised BenchmarkPlanningEnv, 4 movers @ 65 536 envs per MI355X.

    python bench.py --gpus N --steps K --warmup W            # N > 1 without a launcher: this process spawns the N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one env-step of every env of every rank (clip -> 40 control cycles of clamp / integrate / wall check / pair sweep ->
observation, reward, done flags -> same-step auto-reset), i.e. what `VecBenchmarkPlanningEnv.step()` does, with the outputs written
straight into a [T = 50, E, .] rollout buffer. Inputs (a ring of pre-generated random actions) are resident in HBM before the timed
region. Episode phases are desynchronised before the warm-up (elapsed-step counters drawn uniformly from [0, 50)), so that even a
short timed window sees the steady-state TimeLimit rate instead of a burst every 50 steps.

Sub-batches (--sub-batches S, default 4): the rank's 65 536 envs are stepped as S contiguous sub-batches, each with its own handle and
its own HIP stream, all S kernels launched by ONE C-ABI call per env-step (`gpr_step_many`). A step kernel lasts as long as its slowest
wavefront while most SIMDs have run out of work; with S independent launches in flight the tail of one sub-batch overlaps with the
bulk of another (measured, MI355X, us per env-step of 65 536 envs: S = 1 17.7, S = 2 14.1, S = 4 13.2; S = 8 43 -- the runtime maps streams
onto four hardware queues and two streams on one queue run strictly one after the other, which is also why the streams are probed for
real overlap when the env is built, gpr_streams_overlap; launches 1..S-1 are issued by worker threads of the library, a launch costs the
host 3.5 us; a hipGraph of S kernel nodes costs 9.5 us per node to launch on ROCm 7.2 -- scratch/ubench/graphlaunch.hip).
Closed-loop legal (every sub-batch is its own observation -> policy -> action -> step loop on its own stream) and bit
for bit the results of the single launch (random streams are keyed by global env id). `config.single_launch` is the S = 1 number.

Timed region: exactly K steps per block, R back-to-back blocks with R chosen (identically on every rank) so that a region lasts at
least --min-seconds (2 s for the headline, --extra-seconds for every other pass); barrier + synchronize on both sides, max over ranks.
The headline is timed over --regions (5) such regions: `value` = envs of all ranks x K x R x regions / the SUM of the regions' times,
`config.headline_spread` = min / median / max of the regions' env-steps/s. If a region comes in more than 10 % below the best one the
pass is rebuilt once or twice with freshly probed streams (`config.headline_attempts`) and the attempt with the best median is reported.
HIP events are recorded on the streams the kernels are launched on.

For N > 1 the env batch is sharded (one process per GPU, contiguous env ranges, random streams keyed by global env id). `value` is WEAK
scaling (65 536 envs per GPU); BASELINE.json's STRONG shape (65 536 envs in total) is timed in the same run (`config.strong_scaling`).
step() has no exchange step. What crosses xGMI is the rollout flush: `config.rollout_gather` times, at the strong shape, ONE
`all_gather_into_tensor` of every rank's [50, E/N, .] rollout buffer (north_star's collective), the learner-minibatch gather and the
64-byte counters all-reduce, each inside a timed region of env-steps. Rank 0 prints one JSON line.

Workload (BASELINE.json): BenchmarkPlanningEnv, 4 movers, layout ones(6,6), reference defaults (num_cycles 40,
dt 1e-3, v_max 2, a_max 10, circle r 0.11, std_noise 1e-5, TimeLimit 50), actions ~ U(-a_max, a_max).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured streaming ceiling
T_ROLLOUT = 50
NUM_SIMDS = 1024       # 256 CUs x 4 SIMDs
SHADER_MHZ = 2400.0    # MI355X_MICROARCH.md


def algorithmic_bytes_per_env_step(num_movers: int, precision: str = 'f32') -> int:
    """learn_jerk=False: read action + pos + vel + goal (8 M values) + step count (4 B); write pos + vel + observation +
    achieved_goal + desired_goal (10 M values) + reward (one value) + terminated, truncated, 3 info flags (5 B) + step count (4 B).
    fp32 storage: 72 M + 17; f64 storage: 144 M + 21."""
    v = 8 if precision == 'f64' else 4
    return 18 * num_movers * v + v + 13


def view_bytes_per_env_step(num_movers: int, precision: str = 'f32') -> int:
    """SURVEY.md 8(d), PettingZoo per-agent views (BASELINE config 5): M rows of 4M + 2 values + M per-agent rewards, written"""
    v = 8 if precision == 'f64' else 4
    return v * num_movers * (4 * num_movers + 2) + v * num_movers


def kernel_name(precision: str, movers: int, tiles: int, shape: str = 'circle', noise: bool = True, multi: bool = False) -> str:
    """the step kernel instance gpr_step launches for this configuration (gpr_env_kernels.hip: launch_one / launch_flags; layouts
    ones(tiles, tiles) with the default radius are `layout_full`)"""
    st, rt = {'f32': ('float', 'float'), 'f32_f64acc': ('float', 'double'), 'f64': ('double', 'double')}[precision]
    lpe = 1
    while lpe < movers:
        lpe *= 2
    full = 'true' if (rt == 'float' or shape == 'circle') else 'false'
    return f'gpr::k_step{"_n" if multi else ""}<{st}, {rt}, {lpe}, false, {"true" if noise else "false"}, {full}, {0 if shape == "circle" else 1}>'


def committed_profile(kind: str, kernel: str, envs_per_launch: int):
    """what the committed rocprofv3 PMC passes hold for this kernel at this launch size (profiles/round*_{hbm_traffic,valu}.json: lists
    of records keyed by kernel name and envs per launch). Counters cannot be read from inside this process; None when the profiled
    configuration is not the one being run."""
    for rnd in (4, 3, 2, 1):
        try:
            with open(os.path.join(ROOT, 'profiles', f'round{rnd}_{kind}.json')) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        for rec in (d if isinstance(d, list) else [d]):
            if rec.get('kernel') == kernel and rec.get('envs') == envs_per_launch:
                return dict(rec, source=f'profiles/round{rnd}_{kind}.json')
    return None


def valu_block(kernel: str, envs_per_launch: int, kernel_us: float, in_flight: int, hbm_frac: float):
    """the VALU figure next to the HBM fraction. From the committed PMC pass: vector instructions per wavefront and
    SQ_ACTIVE_INST_VALU (quad-cycles summed over the SIMDs) of ONE launch; `in_flight` such launches share the machine for `kernel_us`
    (live): busy = in_flight x ACTIVE_INST_VALU x 4 / (SIMDs x kernel clocks). `hbm_frac_at_full_issue` = the HBM fraction this
    instruction stream would reach if the SIMDs issued it without a gap -- the ceiling the achieved fraction is to be read against."""
    rec = committed_profile('valu', kernel, envs_per_launch)
    if rec is None:
        return None
    busy = in_flight * rec['SQ_ACTIVE_INST_VALU'] * 4.0 / (NUM_SIMDS * kernel_us * SHADER_MHZ)
    return {'insts_per_wavefront': rec['SQ_INSTS_VALU'] / rec['SQ_WAVES'], 'salu_insts_per_wavefront': rec['SQ_INSTS_SALU'] / rec['SQ_WAVES'],
            'wavefronts_per_launch': rec['SQ_WAVES'], 'busy_frac': busy, 'hbm_frac_at_full_issue': hbm_frac / busy if busy > 0 else None,
            'source': rec['source'],
            'note': 'busy = launches_in_flight x SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel_us x 2.4 GHz); counters of one launch from the '
                    'committed PMC pass, kernel time live'}


# ----------------------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` with no launcher around it. Runs BEFORE anything touches the GPU (and never re-execs): N
# child processes, one per GPU, rendezvous on 127.0.0.1; rank 0's JSON line is passed through. The children are POLLED: the first one
# that exits non-zero (or the overall timeout) ends the others -- a dead rank must not leave its peers in a collective for ever.
# ----------------------------------------------------------------------------------------------------------------------
def self_launch(n: int, timeout_s: float) -> int:
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    out0 = tempfile.TemporaryFile(mode='w+')
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        # rank 0's stdout carries the JSON line; every rank's stderr goes to ours (a failing rank says why)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    t0, failed = time.monotonic(), None
    while failed is None:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = f'rank {bad[0][0]} exited with code {bad[0][1]}'
        elif all(c == 0 for c in codes):
            break
        elif time.monotonic() - t0 > timeout_s:
            failed = f'no result within {timeout_s:.0f} s'
        else:
            time.sleep(0.1)
    if failed is not None:
        print(f'bench.py: {failed}; stopping the other ranks', file=sys.stderr, flush=True)
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t1 = time.monotonic()
        while any(p.poll() is None for p in procs) and time.monotonic() - t1 < 5.0:
            time.sleep(0.05)
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
    out0.seek(0)
    lines = [ln for ln in out0.read().splitlines() if ln.startswith('{')]
    if lines and failed is None:
        print(lines[-1], flush=True)
        return 0
    return 1


# ----------------------------------------------------------------------------------------------------------------------
class Pass:
    """One timed configuration on this rank: `envs` envs stepped as `sub` sub-batches (1: one handle, one launch per step)."""

    def __init__(self, args, rank, local_rank, world, *, envs, std_noise=1e-5, precision=None, movers=None, tiles=None, flush=None, sub=1,
                 collision_params=None, layout=None, learn_jerk=False, views=False):
        import numpy as np
        import torch

        import gymnasium_planar_robotics_amd as gpr

        self.args, self.rank, self.world = args, rank, world
        self.device = torch.device('cuda', local_rank)
        torch.cuda.set_device(self.device)
        self.M, self.E, self.S = movers or args.movers, envs, sub
        tiles = tiles or args.tiles
        precision = precision or args.precision
        kw = dict(num_envs=envs, device=self.device, std_noise=std_noise, precision=precision, env_id_offset=rank * envs, collision_params=collision_params,
                  learn_jerk=learn_jerk)
        lay = np.ones((tiles, tiles)) if layout is None else layout
        self.views, self.penv = views, None
        if views:  # BASELINE config 5: the PettingZoo-parallel env; its step kernels also write every agent's view
            self.penv = gpr.VecBenchmarkPlanningParallelEnv(lay, self.M, num_sub_batches=sub, **kw)
            self.env = self.penv.env
            self.S = sub = self.penv.num_sub_batches
            self.stream_probe = getattr(self.env, 'stream_probe', None)
            self.streams = self.env.streams if sub > 1 else [torch.cuda.current_stream(self.device)]
        elif sub > 1:
            self.env = gpr.VecBenchmarkPlanningEnvSubBatched(lay, self.M, num_sub_batches=sub, **kw)
            self.S = sub = self.env.num_sub_batches  # (as many as there are streams that really overlap on this device)
            self.stream_probe = self.env.stream_probe
            self.streams = self.env.streams
            if sub == 1:
                self.env, self.streams = self.env.subs[0], [torch.cuda.current_stream(self.device)]
        else:
            self.env = gpr.VecBenchmarkPlanningEnv(lay, self.M, **kw)
            self.streams = [torch.cuda.current_stream(self.device)]
        env, M, E, T = self.env, self.M, self.E, T_ROLLOUT
        self.buf = gpr.distributed.RolloutBuffer(T, E, M, env.obs_dim, dtype=env.dtype, device=self.device)
        gen = torch.Generator(device=self.device).manual_seed(1234 + rank)
        lim = env.j_max if learn_jerk else env.a_max
        ring = (torch.rand(T, E, 2 * M, device=self.device, generator=gen, dtype=torch.float32) * (2 * lim) - lim).to(env.dtype)
        self.buf['action'].copy_(ring)  # the rollout buffer's action field doubles as the action ring
        self.outs, self.actions = [], []
        V = 4  # per-agent views: a ring of 4 [E, M, 4M+2] slots (a [50, E, M, 4M+2] buffer would be 6.9 GB at config 5)
        if views:
            self.view_ring = [(torch.zeros(E, M, 4 * M + 2, dtype=env.dtype, device=self.device), torch.zeros(E, M, dtype=env.dtype, device=self.device))
                              for _ in range(V)]
        for t in range(T):
            s = self.buf.slot(t)
            kw_out = dict(observation=s['observation'], achieved_goal=s['achieved_goal'], desired_goal=s['desired_goal'],
                          reward=s['reward'], terminated=s['terminated'], truncated=s['truncated'])
            if views:
                kw_out.update(agent_observation=self.view_ring[t % V][0], agent_reward=self.view_ring[t % V][1])
                self.outs.append(self.penv.bind_outputs(**kw_out))
            else:
                self.outs.append(env.bind_outputs(**kw_out))
            self.actions.append(env.pack_actions(self.buf['action'][t]) if sub > 1 else self.buf['action'][t])
        (self.penv or env).reset(seed=0)
        # desynchronise the episode phases (TimeLimit counters), as a long-running job would be
        el = torch.randint(0, T, (E,), device=self.device, generator=gen, dtype=torch.int32)
        if sub > 1:
            for k, s in enumerate(env.subs):
                s.elapsed_steps.copy_(el[env.sub_slice(k)])
        else:
            env.elapsed_steps.copy_(el)
        torch.cuda.synchronize(self.device)
        self.flush = flush
        self.t = 0
        self.slots_written = 0

    def do_steps(self, n):
        env, acts, outs = self.env, self.actions, self.outs
        step = env.step_all_into if self.S > 1 else env.step_into
        t = self.t
        for _ in range(n):
            step(acts[t], outs[t])
            t += 1
            if t == T_ROLLOUT:
                if self.flush is not None:
                    self.flush(self)
                t = 0
        self.t = t
        self.slots_written = min(T_ROLLOUT, self.slots_written + n)

    def sync(self):
        import torch
        torch.cuda.synchronize(self.device)
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize(self.device)

    def run(self, steps, warmup, min_seconds, regions=1):
        """-> (list of (wall seconds over R x steps steps [max over ranks], HIP-event ms over the same region [the longest of the streams
        the kernels ran on]) for `regions` timed regions, R). R starts at 1 and grows (identically on every rank: it follows the
        all-reduced wall time) until a region lasts min_seconds; the calibration regions are not reported."""
        import torch

        def one_region(reps):
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in self.streams]
            self.sync()
            for (e0, _), st in zip(ev, self.streams):
                e0.record(st)  # on the stream the kernels are launched on
            t0 = time.perf_counter()
            for _ in range(reps):
                self.do_steps(steps)
            for (_, e1), st in zip(ev, self.streams):
                e1.record(st)
            self.sync()
            wall = time.perf_counter() - t0
            gpu_ms = max(e0.elapsed_time(e1) for e0, e1 in ev)
            if self.world > 1:
                import torch.distributed as dist
                w = torch.tensor([wall], device=self.device, dtype=torch.float64)
                dist.all_reduce(w, op=dist.ReduceOp.MAX)
                wall = float(w.item())
            return wall, gpu_ms

        reps, out = 1, []
        with torch.cuda.device(self.device):
            self.do_steps(warmup)
            for _ in range(8):
                wall, gpu_ms = one_region(reps)
                if wall >= min_seconds:
                    out.append((wall, gpu_ms))
                    break
                reps = int(math.ceil(reps * min_seconds / max(wall, 1e-6) * 1.15))
            else:
                out.append((wall, gpu_ms))
            while len(out) < regions:
                out.append(one_region(reps))
        return out, reps

    def done_fraction(self):
        """mean over the rollout slots that were actually written of (terminated | truncated)"""
        n = max(1, self.slots_written)
        term, trunc = self.buf['terminated'][:n], self.buf['truncated'][:n]
        return float(((term | trunc) != 0).float().mean().item())

    def close(self):
        import torch
        self.env.close()
        del self.buf, self.outs, self.actions
        torch.cuda.empty_cache()


def timed(args, rank, local_rank, world, steps=None, warmup=None, min_seconds=None, regions=1, **kw):
    p = Pass(args, rank, local_rank, world, **kw)
    K = steps or args.steps
    regs, reps = p.run(K, args.warmup if warmup is None else warmup, args.extra_seconds if min_seconds is None else min_seconds, regions)
    n1 = K * reps                                   # env-steps of the rank's batch per region
    n, wall, gpu_ms = n1 * len(regs), sum(w for w, _ in regs), sum(g for _, g in regs)
    rates = sorted(world * p.E * n1 / w for w, _ in regs)
    # step_us: HIP-event time per env-step of the rank. Every stream runs its n launches back to back, so this is also the average
    # duration of ONE launch as a kernel trace shows it -- S of them (one per sub-batch) in flight at any time
    res = {'wall': wall, 'gpu_ms': gpu_ms, 'reps': reps, 'n_steps': n, 'step_us': gpu_ms * 1e3 / n, 'regions': len(regs),
           'region_seconds': [w for w, _ in regs], 'region_step_us': [g * 1e3 / n1 for _, g in regs],
           'rate_min': rates[0], 'rate_median': rates[len(rates) // 2], 'rate_max': rates[-1],
           'stream_probe': getattr(p, 'stream_probe', None),
           'sub_batches': p.S, 'envs_per_launch': p.E // p.S, 'movers': p.M, 'views': p.views,
           'env_steps_per_s': world * p.E * n / wall, 'done_fraction': p.done_fraction(), 'rollout_bytes': p.buf.nbytes}
    p.close()
    return res


def timed_headline(args, rank, local_rank, world, **kw):
    """the headline pass: --regions timed regions of >= --min-seconds each. A region more than 10 % below the best one means the
    launches did not overlap as probed (streams sharing a hardware queue with somebody else's work): the pass is rebuilt with freshly
    probed streams, at most twice, and the attempt with the best median is kept. Every attempt is listed in the line."""
    attempts = []
    for _ in range(3):
        r = timed(args, rank, local_rank, world, min_seconds=args.min_seconds, regions=args.regions, **kw)
        attempts.append(r)
        if r['rate_min'] >= 0.9 * r['rate_max']:
            break
    best = max(attempts, key=lambda r: r['rate_median'])
    best['attempts'] = [{'env_steps_per_s': a['env_steps_per_s'], 'min': a['rate_min'], 'median': a['rate_median'], 'max': a['rate_max'],
                         'stream_probe': a['stream_probe'], 'sub_batches': a['sub_batches']} for a in attempts]
    return best


def timed_rollout_launch(args, rank, local_rank, world, envs, n_steps=T_ROLLOUT):
    """open-loop rollouts through gpr_step_n: n_steps env-steps per kernel launch (same work per step, same results bit for bit)"""
    import numpy as np
    import torch

    import gymnasium_planar_robotics_amd as gpr

    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)
    env = gpr.VecBenchmarkPlanningEnv(np.ones((args.tiles, args.tiles)), args.movers, num_envs=envs, device=device, std_noise=1e-5,
                                      precision=args.precision, env_id_offset=rank * envs)
    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    acts = (torch.rand(n_steps, envs, 2 * args.movers, device=device, generator=gen, dtype=torch.float32) * (2 * env.a_max) - env.a_max).to(env.dtype)
    roll = env.alloc_rollout(n_steps)
    env.reset(seed=0)
    env.elapsed_steps.copy_(torch.randint(0, T_ROLLOUT, (envs,), device=device, generator=gen, dtype=torch.int32))

    def sync():
        torch.cuda.synchronize(device)
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
            torch.cuda.synchronize(device)

    launches = 2
    with torch.cuda.device(device):
        env.step_n(acts, roll)
        for _ in range(6):
            sync()
            t0 = time.perf_counter()
            for _ in range(launches):
                env.step_n(acts, roll)
            sync()
            wall = time.perf_counter() - t0
            if world > 1:
                import torch.distributed as dist
                w = torch.tensor([wall], device=device, dtype=torch.float64)
                dist.all_reduce(w, op=dist.ReduceOp.MAX)
                wall = float(w.item())
            if wall >= args.extra_seconds:
                break
            launches = int(math.ceil(launches * args.extra_seconds / max(wall, 1e-6) * 1.25))
    env.close()
    n = launches * n_steps
    us = wall / n * 1e6
    return {'what': f'gpr_step_n: {n_steps} env-steps per kernel launch (open-loop actions; outputs identical to {n_steps} step() calls)',
            'kernel': kernel_name(args.precision, args.movers, args.tiles, multi=True), 'envs_per_gpu': envs, 'env_steps_per_s': world * envs * n / wall,
            'us_per_step': us, 'hbm_frac': algorithmic_bytes_per_env_step(args.movers, args.precision) * envs / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}


def timed_pushing(args, local_rank, envs, movers):
    """BASELINE config 3: the pushing env (one launch per step; the reference's env has 1 mover, the 2-mover variant is engine-defined)"""
    import torch

    import gymnasium_planar_robotics_amd as gpr

    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)
    env = gpr.VecBenchmarkPushingEnv(num_envs=envs, num_movers=movers, device=device)
    env.reset(seed=0)
    gen = torch.Generator(device=device).manual_seed(1234)
    env.elapsed_steps.copy_(torch.randint(0, T_ROLLOUT, (envs,), device=device, generator=gen, dtype=torch.int32))
    acts = [(torch.rand(envs, 2 * movers, device=device, generator=gen) * 20 - 10) for _ in range(8)]
    n = 100
    with torch.cuda.device(device):
        for i in range(30):
            env.step(acts[i % 8])
        best = float('inf')
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(n):
                env.step(acts[i % 8])
            e1.record()
            torch.cuda.synchronize(device)
            best = min(best, e0.elapsed_time(e1) / n * 1e3)
    env.close()
    return {'envs': envs, 'movers': movers, 'step_us': best, 'env_steps_per_s': envs / best * 1e6}


def cpu_baseline(args):
    """The oracle (plain-C port of the reference semantics, fp64) on this box's host cores: a bounded sample of the same workload."""
    import numpy as np

    from oracle import oracle

    cores = os.cpu_count() or 1
    threads = max(1, min(cores, 64))
    E = 256 * threads
    env = oracle.OraclePlanningEnv(np.ones((args.tiles, args.tiles)), args.movers, E, std_noise=1e-5, num_threads=threads)
    env.reset(seed=0)
    rng = np.random.default_rng(0)
    acts = rng.uniform(-10, 10, size=(8, E, 2 * args.movers))
    env.step(acts[0])
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < args.cpu_seconds:
        env.step(acts[n % 8])
        n += 1
    dt = time.perf_counter() - t0
    return {'value': E * n / dt, 'unit': 'env-steps/s', 'cores': threads, 'kind': 'port',
            'sample': f'{E} envs x {n} steps ({dt:.1f} s), oracle/gpr_oracle.c (fp64, reference-literal loops), {threads} pthreads'}


def roofline_of(r, precision, movers, tiles, envs=None, shape='circle'):
    """the roofline block of one timed pass. With S sub-batches the rank has S launches in flight, one per stream, each stream's launches
    back to back: `kernel_us` = a stream's HIP-event time / its launches = what a kernel trace shows as the average duration of one launch,
    and achieved = S x algorithmic bytes per launch / kernel_us (= the bytes of one env-step of all the rank's envs / the time it takes).
    Passes through the PettingZoo-parallel env (`views`) add the per-agent rows the same kernel writes."""
    B = algorithmic_bytes_per_env_step(movers, precision) + (view_bytes_per_env_step(movers, precision) if r.get('views') else 0)
    S, epl = r['sub_batches'], r['envs_per_launch']
    kernel_us = r['step_us']
    achieved = S * B * epl / (kernel_us * 1e-6) / 1e9
    name = kernel_name(precision, movers, tiles, shape=shape)
    if precision == 'f32' and shape == 'circle':
        # thin launches run k_step_thin (longer blocks of cycles): gpr_env_kernels.hip launch_kernel; the handle's wavefront budget is
        # 1 024 x {1, 2, 4} by lanes per env, shared by the sub-batches (VecBenchmarkPlanningEnvSubBatched)
        lpe = 1
        while lpe < movers:
            lpe *= 2
        full = 1024 * (4 if lpe >= 16 else 2 if lpe >= 8 else 1)
        budget = int(os.environ.get('GPR_THIN_WAVES', full // S if S > 1 else full))
        if 2 * lpe <= 64 and 2 * ((epl * lpe + 63) // 64) <= budget:
            name = name.replace('gpr::k_step<', 'gpr::k_step_thin<')
    tr = committed_profile('hbm_traffic', name, epl)
    frac = achieved / HBM_PEAK_GBS
    return {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': frac,
            'traffic': tr['hbm_bytes_per_launch'] if tr else None, 'kernel': name, 'kernel_us': kernel_us, 'launches_in_flight': S,
            'envs_per_launch': epl, 'algorithmic_bytes_per_env_step': B, 'algorithmic_bytes_per_launch': B * epl,
            'achieved_is': 'launches_in_flight x algorithmic_bytes_per_launch / kernel_us',
            'valu': valu_block(name, epl, kernel_us, S, frac)}


def config_row(r, precision, movers, tiles, what):
    """one BASELINE.json configuration in the line: what was timed, its rate and its own roofline block"""
    return {'what': what, 'env_steps_per_s': r['env_steps_per_s'], 'step_us': r['step_us'], 'sub_batches': r['sub_batches'],
            'per_agent_views_written': bool(r.get('views')), 'roofline': roofline_of(r, precision, movers, tiles)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=100)
    ap.add_argument('--envs-per-gpu', type=int, default=65536)
    ap.add_argument('--movers', type=int, default=4)
    ap.add_argument('--tiles', type=int, default=6)
    ap.add_argument('--precision', default='f32', choices=['f32', 'f32_f64acc', 'f64'])
    ap.add_argument('--sub-batches', type=int, default=4, help='sub-batches (handles + HIP streams) the rank\'s envs are stepped as; 1 = one launch per step')
    ap.add_argument('--min-seconds', type=float, default=2.0, help='minimum length of each timed region of the HEADLINE pass (blocks of --steps are repeated)')
    ap.add_argument('--regions', type=int, default=5, help='timed regions of the headline pass (value = all their env-steps / all their time; spread reported)')
    ap.add_argument('--extra-seconds', type=float, default=0.15, help='minimum length of the timed region of every other pass')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='headline pass only (profiling runs)')
    ap.add_argument('--no-rollout-gather', action='store_true', help='N > 1: skip the timed rollout all-gather passes')
    ap.add_argument('--rollout-gather', action='store_true', help='(accepted for compatibility: the gather passes are on by default for N > 1)')
    ap.add_argument('--gather-timeout', type=float, default=240.0, help='seconds the rollout all-gather passes may take before the line is printed without them')
    ap.add_argument('--launch-timeout', type=float, default=1500.0, help='self-launch: seconds before the ranks are stopped')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend; 'gloo' + --single-device rehearses N > 1 on one GPU")
    ap.add_argument('--single-device', action='store_true', help='rehearsal: every rank uses cuda:0')
    ap.add_argument('--fail-rank', type=int, default=-1, help='test hook: this rank exits with code 7 before the rendezvous')
    ap.add_argument('--watchdog', type=float, default=1500.0, help='seconds after which a rank dumps its Python stacks and exits (a HIP call that never returns)')
    args = ap.parse_args()
    import faulthandler
    faulthandler.dump_traceback_later(args.watchdog, exit=True)           # a hung run ends with a location instead of a silent kill

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args.gpus, args.launch_timeout))  # nothing has touched the GPU in this process
    if args.fail_rank >= 0 and int(os.environ.get('RANK', '0')) == args.fail_rank:
        print(f'bench.py: rank {args.fail_rank} failing on request (--fail-rank)', file=sys.stderr, flush=True)
        sys.exit(7)

    import gymnasium_planar_robotics_amd as gpr

    if args.single_device:
        os.environ['LOCAL_RANK'] = '0'
    rank, local_rank, world = gpr.distributed.init_from_env(args.backend)
    if world != args.gpus and rank == 0:
        print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; running on {world} ranks', file=sys.stderr)
    backend = None
    if world > 1:
        import torch.distributed as dist
        backend = {'backend': dist.get_backend(), 'world_size_seen_by_backend': dist.get_world_size()}

    E, M, K, S = args.envs_per_gpu, args.movers, args.steps, max(1, args.sub_batches)
    while E % S:
        S //= 2
    head = timed_headline(args, rank, local_rank, world, envs=E, sub=S)           # headline: the reference's default noise
    cfg = {
        'workload': f'BenchmarkPlanningEnv, {M} movers, layout ones({args.tiles},{args.tiles}), {E} envs per GPU, num_cycles 40, '
                    'std_noise 1e-5 (reference default), TimeLimit 50 with desynchronised episode phases, same-step auto-reset, '
                    'actions ~ U(-10, 10)',
        'envs_per_gpu': E, 'num_movers': M, 'parallelism': f'env-shard x{world}, no collective in step()',
        'sub_batches': f"{head['sub_batches']} sub-batches of {E // head['sub_batches']} envs per GPU, one handle + one HIP stream each, {head['sub_batches']} launches per env-step issued by one gpr_step_many call",
        'timed_steps': head['n_steps'], 'timed_blocks': head['reps'], 'timed_regions': head['regions'], 'timed_seconds': head['wall'],
        'headline_spread': {'what': 'env-steps/s of each timed region of the reported attempt', 'min': head['rate_min'], 'median': head['rate_median'],
                            'max': head['rate_max'], 'region_seconds': head['region_seconds'], 'region_kernel_us': head['region_step_us']},
        'headline_attempts': head['attempts'],
        'done_fraction_per_step': head['done_fraction'], 'stream_probe': head['stream_probe'],
        'thin_wavefronts': 'gpr_step launches of batches that leave SIMDs idle are spread over up to 1024 x {1, 2, 4} wavefronts (groups of <= 4 / 8 / 16 lanes '
                           'per env; sub-batched handles share the budget) holding fewer envs each -- include/gpr.h: gpr_set_wave_budget; bit-identical results. '
                           'It applies to the small shapes of config.baseline_configs / other_batch_sizes, not to the headline (4 096 wavefronts = 4 per SIMD)',
    }
    if backend:
        cfg['distributed'] = backend
    line = {
        'metric': 'env-steps/sec, BenchmarkPlanningEnv 4 movers @ 65536 envs per MI355X',
        'value': head['env_steps_per_s'], 'unit': 'env-steps/s', 'n_gpus': world, 'steps': K, 'warmup': args.warmup,
        'ms_per_step': head['wall'] / head['n_steps'] * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': {'f32': 'f32', 'f32_f64acc': 'f32 storage / f64 arithmetic', 'f64': 'f64'}[args.precision], 'data': 'synthetic',
        'config': cfg,
        'roofline': roofline_of(head, args.precision, M, args.tiles, E),
    }
    if world == 1:
        line['strong_value'] = head['env_steps_per_s']  # (one GPU: the strong shape IS the weak one)

    if not args.no_extras:
        if S > 1:
            one = timed(args, rank, local_rank, world, envs=E, sub=1)
            cfg['single_launch'] = {'what': 'the same env-steps as ONE launch per step (one handle, one stream): the kernel lasts as long as its slowest wavefront',
                                    'env_steps_per_s': one['env_steps_per_s'], 'roofline': roofline_of(one, args.precision, M, args.tiles, E)}
        s0 = timed(args, rank, local_rank, world, envs=E, std_noise=0.0, sub=S)
        cfg['sigma0_env_steps_per_s'] = s0['env_steps_per_s']
        cfg['sigma0_step_us'] = s0['step_us']
        if world > 1:
            # BASELINE.json's strong-scaling shape: 65 536 envs in total
            Es = max(1, args.envs_per_gpu // world)
            Ss = S
            while Es % Ss:
                Ss //= 2
            st = timed(args, rank, local_rank, world, envs=Es, sub=Ss)
            line['strong_value'] = st['env_steps_per_s']  # BASELINE's metric read as STRONG scaling: 65 536 envs in total over the N GPUs
            cfg['strong_scaling'] = {'envs_total': Es * world, 'envs_per_gpu': Es, 'sub_batches': Ss, 'env_steps_per_s': st['env_steps_per_s'],
                                     'ms_per_step': st['wall'] / st['n_steps'] * 1e3, 'step_us': st['step_us'],
                                     'rollout_launch': timed_rollout_launch(args, rank, local_rank, world, Es)}
        else:
            sizes = {}
            for n in (4096, 262144, 1048576):
                Sn = 1 if n * M <= 65536 else S   # (a batch this small is one launch per step: see config.baseline_configs)
                r = timed(args, rank, local_rank, world, envs=n, steps=max(50, K * E // max(n, E) // 2), warmup=min(args.warmup, 60), sub=Sn)
                sizes[str(n)] = {'env_steps_per_s': r['env_steps_per_s'], 'step_us': r['step_us'], 'sub_batches': r['sub_batches'],
                                 'hbm_frac': algorithmic_bytes_per_env_step(M, args.precision) * n / (r['step_us'] * 1e-6) / 1e9 / HBM_PEAK_GBS}
            cfg['other_batch_sizes'] = sizes
            cfg['rollout_launch'] = {str(n): timed_rollout_launch(args, rank, local_rank, world, n) for n in (8192, E)}
            if (M, args.tiles, args.precision) == (4, 6, 'f32'):
                # BASELINE.json's other configurations AS STATED, at the size ONE GPU of its 8-GPU node gets and whole, each with its own
                # roofline block (parity-test cases, not the headline). Config 5 goes through the PettingZoo-parallel env: the per-agent
                # views (4 M (4M+2) + 4 M bytes per env-step) are written inside the timed region by the same kernel.
                shapes = {}
                for name, kw, what in (
                        ('config2: 4 movers x 4096 envs', dict(envs=4096, movers=4, tiles=6), 'BenchmarkPlanningEnv, 4 movers, 4096 envs on 1 MI355X'),
                        ('config4_shard: 8 movers x 8192 envs', dict(envs=8192, movers=8, tiles=8), 'BenchmarkPlanningEnv, 8 movers: the 8192 envs one of 8 GPUs gets'),
                        ('config4: 8 movers x 65536 envs', dict(envs=65536, movers=8, tiles=8), 'BenchmarkPlanningEnv, 8 movers, all 65536 envs on one GPU'),
                        ('config5_shard: 16 movers x 4096 envs, PettingZoo parallel', dict(envs=4096, movers=16, tiles=10, views=True),
                         'VecBenchmarkPlanningParallelEnv, 16 movers: the 4096 envs one of 8 GPUs gets, per-agent views written'),
                        ('config5: 16 movers x 32768 envs, PettingZoo parallel', dict(envs=32768, movers=16, tiles=10, views=True),
                         'VecBenchmarkPlanningParallelEnv, 16 movers, all 32768 envs on one GPU, per-agent views written'),
                        ('config5_single_agent_shard: 16 movers x 4096 envs', dict(envs=4096, movers=16, tiles=10), 'the same envs without the per-agent views'),
                        ('config5_single_agent: 16 movers x 32768 envs', dict(envs=32768, movers=16, tiles=10), 'the same envs without the per-agent views')):
                    r = timed(args, rank, local_rank, world, steps=100, warmup=40, sub=S, **kw)
                    shapes[name] = config_row(r, 'f32', kw['movers'], kw['tiles'], what)
                    if S > 1 and kw['envs'] * kw['movers'] <= 65536:
                        # a per-GPU shard is small enough for ONE launch per step to be the better way to step it (nothing to overlap,
                        # one launch to issue): time that too, report the faster one as the row and keep the other next to it
                        r1 = timed(args, rank, local_rank, world, steps=100, warmup=40, sub=1, **kw)
                        row1 = config_row(r1, 'f32', kw['movers'], kw['tiles'], what)
                        keep, other = (row1, shapes[name]) if r1['step_us'] < r['step_us'] else (shapes[name], row1)
                        keep['other_way'] = {'sub_batches': other['sub_batches'], 'step_us': other['step_us'], 'env_steps_per_s': other['env_steps_per_s']}
                        shapes[name] = keep
                for nm, B in ((1, 273), (2, 401)):
                    r = timed_pushing(args, local_rank, 4096, nm)
                    shapes[f'config3: pushing env, {nm} mover(s) x 4096 envs, one launch per step'] = {
                        'what': 'BenchmarkPushingEnv (contact model gpr-push-v3, engine-defined: parity unpinned)' + (' -- the reference has 1 mover' if nm == 1 else ' -- BASELINE config 3'),
                        'env_steps_per_s': r['env_steps_per_s'], 'step_us': r['step_us'],
                        'roofline': {'bound': 'hbm', 'achieved': B * 4096 / (r['step_us'] * 1e-6) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                     'frac': B * 4096 / (r['step_us'] * 1e-6) / 1e9 / HBM_PEAK_GBS, 'traffic': None, 'kernel': 'gpr::k_push', 'kernel_us': r['step_us'],
                                     'algorithmic_bytes_per_env_step': B,
                                     'note': 'SURVEY 8(d): 4 (30 + 34) + 17 B with one mover; + 13 state values read and written, 2 action and 4 observation values per further mover'}}
                cfg['baseline_configs'] = shapes
            if args.precision == 'f32':
                # the reference's own arithmetic (float64): its own roofline block, kernel name and bytes
                ref = {}
                for pr in ('f64', 'f32_f64acc'):
                    r = timed(args, rank, local_rank, world, envs=E, precision=pr, steps=max(50, K // 4), warmup=min(args.warmup, 60), sub=S)
                    ref[pr] = dict(roofline_of(r, pr, M, args.tiles, E), env_steps_per_s=r['env_steps_per_s'])
                line['roofline_reference_dtype'] = ref['f64']
                cfg['reference_dtype'] = {k: {'env_steps_per_s': v['env_steps_per_s'], 'step_us': v['kernel_us'], 'hbm_frac': v['frac'],
                                              'kernel': v['kernel']} for k, v in ref.items()}

    if world > 1 and not args.no_rollout_gather:
        # north_star / SURVEY 8(e): "a single RCCL gather of rollout tensors at buffer flush", timed at the strong shape (65 536 envs in
        # total: [50, E/N, .] per rank = 56 MB at N = 8) inside a timed region of env-steps. A watchdog makes sure the headline is
        # printed if a collective hangs, and that the hang is NOT reported as success.
        import threading

        def give_up():
            if rank == 0:
                cfg.setdefault('rollout_gather', {})['status'] = f'no result within {args.gather_timeout:.0f} s'
                print(json.dumps(line), flush=True)
            os._exit(3)

        dog = threading.Timer(args.gather_timeout, give_up)
        dog.daemon = True
        dog.start()
        Es = max(1, args.envs_per_gpu // world)
        Ss = S
        while Es % Ss:
            Ss //= 2

        def fenced(p, fn):  # the collective runs on torch's current stream: behind the sub-batches' kernels, ahead of their next ones
            if p.S > 1:
                p.env.synchronize()
            fn()
            if p.S > 1:
                p.env.wait_for_caller()

        def flush_all(p):
            fenced(p, p.buf.flush)  # ONE all_gather_into_tensor of the rank's whole [50, E/N, .] buffer

        def flush_minibatch(p):
            fenced(p, lambda: p.buf.gather_minibatch(min(1024, p.E)))  # ONE all-gather of a learner-sized [50, 1024, .] slice

        def flush_counters(p):
            p.env.all_reduce_episode_counters()  # ONE all_reduce of eight uint64 counters; rollouts stay sharded

        base = timed(args, rank, local_rank, world, envs=Es, sub=Ss, steps=max(100, K // 4))
        g = timed(args, rank, local_rank, world, envs=Es, sub=Ss, steps=max(100, K // 4), flush=flush_all)
        mb = timed(args, rank, local_rank, world, envs=Es, sub=Ss, steps=max(100, K // 4), flush=flush_minibatch)
        fc = timed(args, rank, local_rank, world, envs=Es, sub=Ss, steps=max(100, K // 4), flush=flush_counters)
        dog.cancel()
        per_flush = lambda r: (r['wall'] / r['n_steps'] - base['wall'] / base['n_steps']) * T_ROLLOUT * 1e3  # noqa: E731
        cfg['rollout_gather'] = {
            'what': f'at the strong shape ({Es} envs per GPU, {world} ranks), every {T_ROLLOUT} env-steps inside the timed region: ONE all_gather_into_tensor of '
                    'every rank\'s [50, E/N, .] rollout buffer; or ONE all-gather of a [50, 1024, .] learner minibatch; or ONE all_reduce of the eight '
                    'episode counters (64 B)',
            'status': 'ok', 'bytes_per_rank_per_flush': g['rollout_bytes'],
            'env_steps_per_s': g['env_steps_per_s'], 'ms_per_flush': per_flush(g),
            'no_flush_env_steps_per_s': base['env_steps_per_s'],
            'minibatch': {'env_steps_per_s': mb['env_steps_per_s'], 'ms_per_flush': per_flush(mb)},
            'counters_only': {'env_steps_per_s': fc['env_steps_per_s'], 'ms_per_flush': per_flush(fc)}}

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args)
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the vectorised BenchmarkPlanningEnv, 4 movers @ 65 536 envs per MI355X.

    python bench.py --gpus N --steps K --warmup W            # N > 1 without a launcher: this process spawns the N ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one env-step of every env of every rank: ONE fused HIP kernel launch per rank per step (clip -> 40 control
cycles of clamp / integrate / wall check / pair sweep -> observation, reward, done flags -> same-step auto-reset), i.e. what
`VecBenchmarkPlanningEnv.step()` does, with the outputs written straight into a [T = 50, E, .] rollout buffer. Inputs
(a ring of pre-generated random actions) are resident in HBM before the timed region. Episode phases are desynchronised
before the warm-up (elapsed-step counters drawn uniformly from [0, 50)), so that even a short timed window sees the
steady-state TimeLimit rate instead of a burst every 50 steps.

Timed region: exactly K steps per block, R back-to-back blocks with R chosen (identically on every rank) so that the
region lasts at least --min-seconds (0.1 s); barrier + synchronize on both sides, max over ranks. `value` = envs of all
ranks x K x R / that time.

For N > 1 the env batch is sharded (one process per GPU, contiguous env ranges, random streams keyed by global env id).
`value` is WEAK scaling (65 536 envs per GPU); BASELINE.json's STRONG shape (65 536 envs in total) is timed in the same run and
reported under config.strong_scaling. step() has no exchange step, so no timed region of an env-step rate holds a
collective. Rollouts stay sharded (every rank's learner consumes its own [T, E/N, .] buffer); what crosses xGMI at a
flush is ONE all_reduce of the six episode counters (48 bytes), timed as config.rollout_flush. Rank 0 prints one JSON line.

Workload (BASELINE.json): BenchmarkPlanningEnv, 4 movers, layout ones(6,6), reference defaults
(num_cycles 40, dt 1e-3, v_max 2, a_max 10, circle r 0.11, std_noise 1e-5, TimeLimit 50), actions ~ U(-a_max, a_max).
Extras in `config` (never `value`): sigma = 0; other batch sizes (4 096, 262 144 and 1 048 576 envs); the reference's dtype
(f64 and f32 storage / f64 arithmetic); open-loop rollouts with 50 env-steps per launch (gpr_step_n) at 65 536 envs and at the
8 192-envs-per-GPU shard of the strong-scaling shape.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured streaming ceiling
T_ROLLOUT = 50


def algorithmic_bytes_per_env_step(num_movers: int) -> int:
    """fp32, learn_jerk=False: read action + pos + vel + goal (32 M) + step count (4); write pos + vel +
    observation + achieved_goal + desired_goal (40 M) + reward (4) + terminated, truncated, 3 info flags (5) + step count (4)."""
    return 72 * num_movers + 17


def measured_traffic(num_envs: int, num_movers: int):
    """HBM bytes per step-kernel launch from the committed rocprofv3 PMC passes (profiles/round*_hbm_traffic.json: FETCH_SIZE
    and WRITE_SIZE in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950). Counters cannot be read
    from inside this process; None when the profiled configuration is not the one being run."""
    for name in ('round2_hbm_traffic.json', 'round1_hbm_traffic.json'):
        try:
            with open(os.path.join(ROOT, 'profiles', name)) as f:
                d = json.load(f)
            if d['envs'] == num_envs and d['movers'] == num_movers:
                return d['hbm_bytes_per_launch']
        except (OSError, KeyError, ValueError):
            pass
    return None


# ----------------------------------------------------------------------------------------------------------------------
# self-launch: `python bench.py --gpus N` with no launcher around it. Runs BEFORE anything touches the GPU (and never
# re-execs): N child processes, one per GPU, rendezvous on 127.0.0.1; rank 0's JSON line is passed through.
# ----------------------------------------------------------------------------------------------------------------------
def self_launch(n: int) -> int:
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    lines = [ln for ln in (out or '').splitlines() if ln.startswith('{')]
    if lines:
        print(lines[-1], flush=True)
    return max(abs(rc) for rc in rcs) if any(rcs) else (0 if lines else 1)


# ----------------------------------------------------------------------------------------------------------------------
class Pass:
    """One timed configuration on this rank."""

    def __init__(self, args, rank, local_rank, world, *, envs, std_noise=1e-5, precision=None, movers=None, tiles=None, flush=None):
        import numpy as np
        import torch

        import gymnasium_planar_robotics_amd as gpr

        self.args, self.rank, self.world = args, rank, world
        self.device = torch.device('cuda', local_rank)
        torch.cuda.set_device(self.device)
        self.M, self.E = movers or args.movers, envs
        tiles = tiles or args.tiles
        precision = precision or args.precision
        self.env = gpr.VecBenchmarkPlanningEnv(np.ones((tiles, tiles)), self.M, num_envs=envs, device=self.device, std_noise=std_noise,
                                               precision=precision, env_id_offset=rank * envs)
        env, M, E, T = self.env, self.M, self.E, T_ROLLOUT
        self.buf = gpr.distributed.RolloutBuffer(T, E, M, env.obs_dim, dtype=env.dtype, device=self.device)
        self.outs = []
        for t in range(T):
            s = self.buf.slot(t)
            self.outs.append(env.bind_outputs(s['observation'], s['achieved_goal'], s['desired_goal'], s['reward'], s['terminated'], s['truncated']))
        gen = torch.Generator(device=self.device).manual_seed(1234 + rank)
        ring = (torch.rand(T, E, 2 * M, device=self.device, generator=gen, dtype=torch.float32) * (2 * env.a_max) - env.a_max).to(env.dtype)
        self.buf['action'].copy_(ring)  # the rollout buffer's action field doubles as the action ring
        self.actions = [self.buf['action'][t] for t in range(T)]
        env.reset(seed=0)
        # desynchronise the episode phases (TimeLimit counters), as a long-running job would be
        env.elapsed_steps.copy_(torch.randint(0, T, (E,), device=self.device, generator=gen, dtype=torch.int32))
        self.flush = flush
        self.t = 0
        self.slots_written = 0

    def do_steps(self, n):
        env, acts, outs = self.env, self.actions, self.outs
        t = self.t
        for _ in range(n):
            env.step_into(acts[t], outs[t])
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

    def run(self, steps, warmup, min_seconds):
        """-> (wall seconds over R x steps steps [max over ranks], HIP-event ms over the same region, R). R starts at 1 and
        grows (identically on every rank: it follows the all-reduced wall time) until the region lasts min_seconds."""
        import torch
        reps = 1
        with torch.cuda.device(self.device):
            self.do_steps(warmup)
            for _ in range(6):
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                self.sync()
                ev0.record()  # on the stream the kernels are launched on (torch's current stream is what step_into passes down)
                t0 = time.perf_counter()
                for _ in range(reps):
                    self.do_steps(steps)
                ev1.record()
                self.sync()
                wall = time.perf_counter() - t0
                gpu_ms = ev0.elapsed_time(ev1)
                if self.world > 1:
                    import torch.distributed as dist
                    w = torch.tensor([wall], device=self.device, dtype=torch.float64)
                    dist.all_reduce(w, op=dist.ReduceOp.MAX)
                    wall = float(w.item())
                if wall >= min_seconds:
                    break
                reps = int(math.ceil(reps * min_seconds / max(wall, 1e-6) * 1.25))
        return wall, gpu_ms, reps

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


def timed(args, rank, local_rank, world, steps=None, warmup=None, **kw):
    p = Pass(args, rank, local_rank, world, **kw)
    wall, gpu_ms, reps = p.run(steps or args.steps, args.warmup if warmup is None else warmup, args.min_seconds)
    n = (steps or args.steps) * reps
    res = {'wall': wall, 'gpu_ms': gpu_ms, 'reps': reps, 'n_steps': n, 'kernel_us': gpu_ms * 1e3 / n,
           'env_steps_per_s': world * p.E * n / wall, 'done_fraction': p.done_fraction(), 'rollout_bytes': p.buf.nbytes}
    p.close()
    return res


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
            if wall >= args.min_seconds:
                break
            launches = int(math.ceil(launches * args.min_seconds / max(wall, 1e-6) * 1.25))
    env.close()
    n = launches * n_steps
    us = wall / n * 1e6
    return {'what': f'gpr_step_n: {n_steps} env-steps per kernel launch (open-loop actions; outputs identical to {n_steps} step() calls)',
            'kernel': 'gpr::k_step_n<float, float, 4, false, true, true, 0>', 'envs_per_gpu': envs, 'env_steps_per_s': world * envs * n / wall,
            'us_per_step': us, 'hbm_frac': algorithmic_bytes_per_env_step(args.movers) * envs / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2000)
    ap.add_argument('--warmup', type=int, default=100)
    ap.add_argument('--envs-per-gpu', type=int, default=65536)
    ap.add_argument('--movers', type=int, default=4)
    ap.add_argument('--tiles', type=int, default=6)
    ap.add_argument('--precision', default='f32', choices=['f32', 'f32_f64acc', 'f64'])
    ap.add_argument('--min-seconds', type=float, default=0.1, help='minimum length of every timed region (blocks of --steps are repeated)')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='headline pass only (profiling runs)')
    ap.add_argument('--rollout-gather', action='store_true', help='also time a full all-gather of the rollout buffers at every flush (N > 1)')
    ap.add_argument('--gather-timeout', type=float, default=240.0, help='seconds the optional rollout all-gather pass may take')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend; 'gloo' + --single-device rehearses N > 1 on one GPU")
    ap.add_argument('--single-device', action='store_true', help='rehearsal: every rank uses cuda:0')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        sys.exit(self_launch(args.gpus))  # nothing has touched the GPU in this process

    import gymnasium_planar_robotics_amd as gpr

    if args.single_device:
        os.environ['LOCAL_RANK'] = '0'
    rank, local_rank, world = gpr.distributed.init_from_env(args.backend)
    if world != args.gpus and rank == 0:
        print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; running on {world} ranks', file=sys.stderr)

    E, M, K = args.envs_per_gpu, args.movers, args.steps
    head = timed(args, rank, local_rank, world, envs=E)                    # headline: the reference's default noise
    B = algorithmic_bytes_per_env_step(M)
    kernel_s = head['kernel_us'] * 1e-6  # HIP events around the timed region / launches: average duration of the step kernel
    achieved = B * E / kernel_s / 1e9
    cfg = {
        'workload': f'BenchmarkPlanningEnv, {M} movers, layout ones({args.tiles},{args.tiles}), {E} envs per GPU, num_cycles 40, '
                    'std_noise 1e-5 (reference default), TimeLimit 50 with desynchronised episode phases, same-step auto-reset, '
                    'actions ~ U(-10, 10)',
        'envs_per_gpu': E, 'num_movers': M, 'parallelism': f'env-shard x{world}, no collective in step()',
        'timed_steps': head['n_steps'], 'timed_blocks': head['reps'], 'timed_seconds': head['wall'],
        'done_fraction_per_step': head['done_fraction'],
    }
    line = {
        'metric': 'env-steps/sec, BenchmarkPlanningEnv 4 movers @ 65536 envs per MI355X',
        'value': head['env_steps_per_s'], 'unit': 'env-steps/s', 'n_gpus': world, 'steps': K, 'warmup': args.warmup,
        'ms_per_step': head['wall'] / head['n_steps'] * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': {'f32': 'f32', 'f32_f64acc': 'f32 storage / f64 arithmetic', 'f64': 'f64'}[args.precision], 'data': 'synthetic',
        'config': cfg,
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                     'traffic': measured_traffic(E, M), 'kernel': 'gpr::k_step<float, float, 4, false, true, true, 0>',
                     'kernel_us': head['kernel_us'], 'algorithmic_bytes_per_env_step': B},
    }

    if not args.no_extras:
        s0 = timed(args, rank, local_rank, world, envs=E, std_noise=0.0)
        cfg['sigma0_env_steps_per_s'] = s0['env_steps_per_s']
        cfg['sigma0_kernel_us'] = s0['kernel_us']
        if world > 1:
            # BASELINE.json's strong-scaling shape: 65 536 envs in total
            Es = max(1, args.envs_per_gpu // world)
            st = timed(args, rank, local_rank, world, envs=Es)
            cfg['strong_scaling'] = {'envs_total': Es * world, 'envs_per_gpu': Es, 'env_steps_per_s': st['env_steps_per_s'],
                                     'ms_per_step': st['wall'] / st['n_steps'] * 1e3, 'kernel_us': st['kernel_us'],
                                     'rollout_launch': timed_rollout_launch(args, rank, local_rank, world, Es)}

            def flush_counters(p):
                p.env.all_reduce_episode_counters()  # ONE all_reduce of six uint64 counters; rollouts stay sharded

            fl = timed(args, rank, local_rank, world, envs=E, flush=flush_counters)
            cfg['rollout_flush'] = {
                'what': 'rollouts stay sharded (each rank\'s learner reads its own [50, E, .] buffer); every 50 steps ONE all_reduce '
                        'of the six on-device episode counters (48 B) inside the timed region',
                'env_steps_per_s': fl['env_steps_per_s'], 'rollout_bytes_per_rank_per_flush': fl['rollout_bytes']}
        else:
            sizes = {}
            for n in (4096, 262144, 1048576):
                r = timed(args, rank, local_rank, world, envs=n, steps=max(50, K * E // max(n, E) // 2), warmup=min(args.warmup, 60))
                sizes[str(n)] = {'env_steps_per_s': r['env_steps_per_s'], 'kernel_us': r['kernel_us'],
                                 'hbm_frac': B * n / (r['kernel_us'] * 1e-6) / 1e9 / HBM_PEAK_GBS}
            cfg['other_batch_sizes'] = sizes
            cfg['rollout_launch'] = {str(n): timed_rollout_launch(args, rank, local_rank, world, n) for n in (8192, E)}
            if args.precision == 'f32':
                prec = {}
                for pr in ('f32_f64acc', 'f64'):
                    r = timed(args, rank, local_rank, world, envs=E, precision=pr, steps=max(50, K // 4), warmup=min(args.warmup, 60))
                    bytes_pr = B if pr == 'f32_f64acc' else 2 * (72 * M) + 17
                    prec[pr] = {'env_steps_per_s': r['env_steps_per_s'], 'kernel_us': r['kernel_us'],
                                'hbm_frac': bytes_pr * E / (r['kernel_us'] * 1e-6) / 1e9 / HBM_PEAK_GBS}
                cfg['reference_dtype'] = prec

    if world > 1 and args.rollout_gather:
        # optional: + a full all-gather of every rank's rollout buffer per flush (GBs over xGMI; bound by the links, not by
        # the env engine -- DESIGN.md section 4). A watchdog makes sure the headline is printed if the collective hangs, and
        # that the hang is NOT reported as success.
        import threading

        def give_up():
            if rank == 0:
                cfg['rollout_gather'] = {'status': f'no result within {args.gather_timeout:.0f} s'}
                print(json.dumps(line), flush=True)
            os._exit(3)

        dog = threading.Timer(args.gather_timeout, give_up)
        dog.daemon = True
        dog.start()
        g = timed(args, rank, local_rank, world, envs=E, flush=lambda p: p.buf.flush())
        dog.cancel()
        cfg['rollout_gather'] = {
            'what': 'ONE all_gather_into_tensor of every rank\'s [50, E, .] rollout buffer per 50 steps inside the timed region',
            'env_steps_per_s': g['env_steps_per_s'], 'bytes_per_rank_per_flush': g['rollout_bytes']}

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

#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the vectorised BenchmarkPlanningEnv, 4 movers @ 65 536 envs per MI355X.

    python bench.py --gpus 1 --steps 2000 --warmup 100
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one env-step of every env of every rank: ONE fused HIP kernel launch per rank per step (clip -> 40 control
cycles of clamp / integrate / wall check / pair sweep -> observation, reward, done flags -> same-step auto-reset), i.e. what
`VecBenchmarkPlanningEnv.step()` does, with the outputs written straight into a [T = 50, E, .] rollout buffer. Inputs
(a ring of pre-generated random actions) are resident in HBM before the timed region. For N > 1 the env batch is sharded
(weak scaling: 65 536 envs per GPU, one process per GPU); step() has no exchange step, so the timed region of `value` holds
no collective (barrier + synchronize on both sides, max over ranks). The one collective the design has -- ONE RCCL
all-gather of the whole rollout buffer at a flush, every 50 steps -- is timed in a second pass of the same run and reported
under config.rollout_gather: at this engine's rate a rank produces ~134 B x 2.9e9 env-steps/s = 390 GB/s of rollout data, and
gathering eight such streams into every GPU is bound by xGMI, not by the env engine. Rank 0 prints one JSON line.

Workload (BASELINE.json): BenchmarkPlanningEnv, 4 movers, layout ones(6,6), reference defaults
(num_cycles 40, dt 1e-3, v_max 2, a_max 10, circle r 0.11, std_noise 1e-5, TimeLimit 50), actions ~ U(-a_max, a_max).
The sigma = 0 variant is timed too and reported in `config` (extras), never as `value`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured streaming ceiling


def algorithmic_bytes_per_env_step(num_movers: int) -> int:
    """fp32, learn_jerk=False: read action + pos + vel + goal (32 M) + step count (4); write pos + vel +
    observation + achieved_goal + desired_goal (40 M) + reward (4) + terminated, truncated, 3 info flags (5) + step count (4)."""
    return 72 * num_movers + 17


def measured_traffic(num_envs: int, num_movers: int):
    """HBM bytes per step-kernel launch from the committed rocprofv3 PMC passes (profiles/round1_hbm_traffic.json: FETCH_SIZE
    and WRITE_SIZE in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950). Counters cannot be read
    from inside this process; None when the profiled configuration is not the one being run."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'round1_hbm_traffic.json')) as f:
            d = json.load(f)
        if d['envs'] == num_envs and d['movers'] == num_movers:
            return d['hbm_bytes_per_launch']
    except (OSError, KeyError, ValueError):
        pass
    return None


def run_gpu(args, rank, local_rank, world, std_noise, gather=False):
    import torch.distributed as dist

    import gymnasium_planar_robotics_amd as gpr

    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)
    M, E, T = args.movers, args.envs_per_gpu, 50
    env = gpr.VecBenchmarkPlanningEnv(np.ones((args.tiles, args.tiles)), M, num_envs=E, device=device, std_noise=std_noise,
                                      precision=args.precision, env_id_offset=rank * E)
    buf = gpr.distributed.RolloutBuffer(T, E, M, env.obs_dim, dtype=env.dtype, device=device)
    outs = []
    for t in range(T):
        s = buf.slot(t)
        outs.append(env.bind_outputs(s['observation'], s['achieved_goal'], s['desired_goal'], s['reward'], s['terminated'], s['truncated']))
    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    ring = (torch.rand(T, E, 2 * M, device=device, generator=gen, dtype=torch.float32) * (2 * env.a_max) - env.a_max).to(env.dtype)
    buf['action'].copy_(ring)  # the rollout buffer's action field doubles as the action ring
    actions = [buf['action'][t] for t in range(T)]
    env.reset(seed=0)

    def do_steps(n, t0):
        t = t0
        for _ in range(n):
            env.step_into(actions[t], outs[t])
            t += 1
            if t == T:
                if gather:
                    buf.flush()  # single RCCL all-gather of the rank's whole rollout buffer
                t = 0
        return t

    def sync():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(device)

    with torch.cuda.device(device):
        t_pos = do_steps(args.warmup, 0)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        sync()
        ev0.record()  # on the stream the kernels are launched on (torch's current stream is what step_into passes down)
        t0 = time.perf_counter()
        do_steps(args.steps, t_pos)
        ev1.record()
        sync()
        wall = time.perf_counter() - t0
    gpu_ms = ev0.elapsed_time(ev1)
    if world > 1:
        w = torch.tensor([wall], device=device, dtype=torch.float64)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        wall = float(w.item())
    done_frac = float((buf['terminated'].float().mean() + buf['truncated'].float().mean()).item())
    nbytes = buf.nbytes
    env.close()
    del buf
    torch.cuda.empty_cache()
    return wall, gpu_ms, done_frac, nbytes


def cpu_baseline(args):
    """The oracle (plain-C port of the reference semantics, fp64) on this box's host cores: a bounded sample of the same workload."""
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
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-rollout-gather', action='store_true', help='skip the extra pass that times the rollout all-gather (N > 1)')
    ap.add_argument('--gather-timeout', type=float, default=240.0, help='seconds the rollout all-gather pass may take (N > 1)')
    ap.add_argument('--backend', default='nccl', help="torch.distributed backend; 'gloo' + --single-device rehearses N > 1 on one GPU")
    ap.add_argument('--single-device', action='store_true', help='rehearsal: every rank uses cuda:0')
    args = ap.parse_args()

    import gymnasium_planar_robotics_amd as gpr

    if args.single_device:
        os.environ['LOCAL_RANK'] = '0'
    rank, local_rank, world = gpr.distributed.init_from_env(args.backend)
    assert world == args.gpus, f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}'

    wall, gpu_ms, done_frac, nbytes = run_gpu(args, rank, local_rank, world, 1e-5)  # headline: the reference's default noise
    wall0, gpu_ms0, _, _ = run_gpu(args, rank, local_rank, world, 0.0)              # extra: sigma = 0

    E, M, K = args.envs_per_gpu, args.movers, args.steps
    kernel_s = gpu_ms * 1e-3 / K  # HIP events around the timed region / launches: average duration of the step kernel
    B = algorithmic_bytes_per_env_step(M)
    achieved = B * E / kernel_s / 1e9
    line = {
        'metric': 'env-steps/sec, BenchmarkPlanningEnv 4 movers @ 65536 envs per MI355X',
        'value': world * E * K / wall, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': K, 'warmup': args.warmup,
        'ms_per_step': wall / K * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': {'f32': 'f32', 'f32_f64acc': 'f32 storage / f64 arithmetic', 'f64': 'f64'}[args.precision], 'data': 'synthetic',
        'config': {
            'workload': f'BenchmarkPlanningEnv, {M} movers, layout ones({args.tiles},{args.tiles}), {E} envs per GPU, num_cycles 40, '
                        'std_noise 1e-5 (reference default), TimeLimit 50, same-step auto-reset, actions ~ U(-10, 10)',
            'envs_per_gpu': E, 'num_movers': M, 'parallelism': f'env-shard x{world}, no collective in step()',
            'done_fraction_per_step': done_frac,
            'sigma0_env_steps_per_s': world * E * K / wall0, 'sigma0_kernel_us': gpu_ms0 * 1e3 / K,
        },
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                     'traffic': measured_traffic(E, M), 'kernel': 'gpr::k_step<float, float, 4, false, true, true, 0>', 'kernel_us': kernel_s * 1e6,
                     'algorithmic_bytes_per_env_step': B},
    }

    if world > 1 and not args.no_rollout_gather:
        # extra pass: + the rollout all-gather. It moves GBs per flush over xGMI; a watchdog makes sure that the line above is
        # printed (without this extra) even if that collective should not complete.
        import threading

        def give_up():
            if rank == 0:
                line['config']['rollout_gather'] = {'status': f'no result within {args.gather_timeout:.0f} s'}
                print(json.dumps(line), flush=True)
            os._exit(0)

        dog = threading.Timer(args.gather_timeout, give_up)
        dog.daemon = True
        dog.start()
        wall_g = run_gpu(args, rank, local_rank, world, 1e-5, gather=True)[0]
        dog.cancel()
        flushes = max(1, (args.warmup % 50 + K) // 50)
        line['config']['rollout_gather'] = {
            'what': 'same run with ONE RCCL all_gather_into_tensor of every rank\'s [50, E, .] rollout buffer per 50 steps inside the timed region',
            'env_steps_per_s': world * E * K / wall_g, 'bytes_per_rank_per_flush': nbytes,
            'ms_per_flush': max(0.0, wall_g - wall) / flushes * 1e3,
            'allgather_busbw_GBps': (world - 1) * nbytes / max(1e-9, (wall_g - wall) / flushes) / 1e9}

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

# how much of a 16-mover step is the reset search: the same batch with max_reset_trials = 10 000 (the reference's unbounded loop) and with 4
# (the first round of four parallel trials always stands: wrong semantics -- overlapping starts collide at once -- but it bounds the search)
import sys, numpy as np, torch
sys.path.insert(0, '.')
import gymnasium_planar_robotics_amd as gpr
for E, M, tiles in ((4096, 16, 10), (32768, 16, 10), (4096, 8, 8)):
    for mrt in (10000, 64, 4):
        env = gpr.VecBenchmarkPlanningEnv(np.ones((tiles, tiles)), M, num_envs=E, max_reset_trials=mrt); env.reset(seed=0)
        env.elapsed_steps.copy_(torch.randint(0, 50, (E,), device='cuda', dtype=torch.int32))
        acts = [(torch.rand(E, 2 * M, device='cuda') * 20 - 10) for _ in range(8)]
        for i in range(100): env.step_into(acts[i % 8], env._out)
        torch.cuda.synchronize()
        best = 1e9; done = 0.0
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(200): env.step_into(acts[i % 8], env._out)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
        done = float((env._out_terminated | env._out_truncated).float().mean()) if hasattr(env, '_out_terminated') else -1
        print(f'{E} envs x {M} movers, max_reset_trials {mrt}: {best:.1f} us/step', flush=True)
        env.close()

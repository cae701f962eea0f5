# deferred resets of the 16-lane kernels (two launches per step): us per step by GPR_DEFER_RESETS (0: one launch; 1: two; 2: the stepping launch alone)
import sys, os, numpy as np, torch
sys.path.insert(0, '.')
import gymnasium_planar_robotics_amd as gpr
for E, M, n in ((4096, 16, 10), (32768, 16, 10), (4096, 12, 8)):
    env = gpr.VecBenchmarkPlanningEnv(np.ones((n, n)), M, num_envs=E, warn_on_reset=False); env.reset(seed=0)
    env.elapsed_steps.copy_(torch.randint(0, 50, (E,), device='cuda', dtype=torch.int32))
    acts = [(torch.rand(E, 2 * M, device='cuda') * 20 - 10) for _ in range(8)]
    for i in range(40): env.step_into(acts[i % 8], env._out)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(100): env.step_into(acts[i % 8], env._out)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 10)
    print(f'DEFER={os.environ.get("GPR_DEFER_RESETS")} {E} x {M}: {best:.1f} us/step (one launch stream)', flush=True)
    env.close()

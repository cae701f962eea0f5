# gpr_step_n us/step for the build GPR_LIB_PATH names, a few shapes (A/B runs in one GPU call)
import sys, os, numpy as np, torch
sys.path.insert(0, '.')
import gymnasium_planar_robotics_amd as gpr
out = []
for E, M, tiles in ((8192, 4, 6), (65536, 4, 6), (65536, 8, 8), (4096, 16, 10)):
    env = gpr.VecBenchmarkPlanningEnv(np.ones((tiles, tiles)), M, num_envs=E); env.reset(seed=0)
    env.elapsed_steps.copy_(torch.randint(0, 50, (E,), device='cuda', dtype=torch.int32))
    acts = (torch.rand(50, E, 2 * M, device='cuda') * 20 - 10)
    roll = env.alloc_rollout(50)
    for _ in range(4): env.step_n(acts, roll)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): env.step_n(acts, roll)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    out.append(f'{E}x{M}: {best:.2f}')
    env.close()
print(os.path.basename(os.environ.get('GPR_LIB_PATH', 'default')), 'step_n us/step:', '  '.join(out), flush=True)

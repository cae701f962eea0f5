# 16-mover step time per library build (GPR_LIB_PATH): 4 096 and 32 768 envs, one launch per step
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
import gymnasium_planar_robotics_amd as gpr
out = []
for E in (4096, 32768):
    env = gpr.VecBenchmarkPlanningEnv(np.ones((10, 10)), 16, num_envs=E); env.reset(seed=0)
    env.elapsed_steps.copy_(torch.randint(0, 50, (E,), device='cuda', dtype=torch.int32))
    acts = [(torch.rand(E, 32, device='cuda') * 20 - 10) for _ in range(8)]
    for i in range(100): env.step_into(acts[i % 8], env._out)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(200): env.step_into(acts[i % 8], env._out)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
    out.append(f'{E} x 16: {best:.1f}')
    env.close()
print(os.path.basename(os.environ.get('GPR_LIB_PATH', 'default')), ' | '.join(out), flush=True)

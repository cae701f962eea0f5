import sys, numpy as np, torch
sys.path.insert(0, '.')
import gymnasium_planar_robotics_amd as gpr
NM=2
for E in (4096, 16384, 65536):
    env = gpr.VecBenchmarkPushingEnv(num_envs=E, num_movers=NM, warn_on_reset=False); env.reset(seed=0)
    env.elapsed_steps.copy_(torch.randint(0, 50, (E,), device='cuda', dtype=torch.int32))
    acts = [(torch.rand(E, 2 * NM, device='cuda') * 20 - 10) for _ in range(8)]
    for i in range(30): env.step(acts[i % 8])
    torch.cuda.synchronize()
    best=1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(100): env.step(acts[i % 8])
        e1.record(); torch.cuda.synchronize()
        best=min(best, e0.elapsed_time(e1) * 10)
    import os
    print(f'PAIR={os.environ.get("GPR_PUSH_PAIR")} LANES={os.environ.get("GPR_PUSH_LANES")} pushing {NM} movers, {E} envs: {best:.1f} us/step', flush=True)
    env._diag_seen = env.diagnostics(); env.close()

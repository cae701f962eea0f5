# step time of the pushing kernel vs batch size (one lane per env: 65 536 envs = one wavefront per SIMD)
import sys, numpy as np, torch
sys.path.insert(0, '.')
import gymnasium_planar_robotics_amd as gpr
for NM in (1, 2):
    for E in (4096, 65536, 262144):
        env = gpr.VecBenchmarkPushingEnv(num_envs=E, num_movers=NM); env.reset(seed=0)
        env.elapsed_steps.copy_(torch.randint(0, 50, (E,), device='cuda', dtype=torch.int32))
        acts = [(torch.rand(E, 2 * NM, device='cuda') * 20 - 10) for _ in range(8)]
        for i in range(30): env.step(acts[i % 8])
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(100): env.step(acts[i % 8])
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 10
        print(f'pushing {NM} mover(s), {E} envs: {us:.1f} us/step, {E / us * 1e6:.3e} env-steps/s', flush=True)
        env.close()

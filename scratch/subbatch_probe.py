# where does the time of a sub-batched env-step go? (1) host issue time per env-step vs GPU time, by number of sub-batches; (2) the same at
# a batch large enough that the host cannot be the limit
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import gymnasium_planar_robotics_amd as gpr
prec = sys.argv[1] if len(sys.argv) > 1 else 'f32'
for E in (65536, 262144):
    for S in (1, 2, 4, 8):
        env = gpr.VecBenchmarkPlanningEnvSubBatched(np.ones((6, 6)), 4, num_envs=E, num_sub_batches=S, precision=prec)
        env.reset(seed=0)
        g = torch.Generator(device='cuda').manual_seed(1)
        for k, s in enumerate(env.subs):
            s.elapsed_steps.copy_(torch.randint(0, 50, (E // S,), device='cuda', generator=g, dtype=torch.int32))
        acts = [(torch.rand(E, 8, device='cuda', generator=g) * 20 - 10).to(env.dtype) for _ in range(8)]
        ptrs = [env.pack_actions(a) for a in acts]
        for i in range(100): env.step_all_into(ptrs[i % 8])
        torch.cuda.synchronize()
        n = 1000 if prec == 'f32' else 300
        best = (1e9, 1e9)
        for rep in range(3):
            t0 = time.perf_counter()
            for i in range(n): env.step_all_into(ptrs[i % 8])
            t_issue = time.perf_counter() - t0
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
            best = min(best, (t_all, t_issue))
        print(f'{prec} E={E:7d} S={S}: issue {best[1] / n * 1e6:6.2f} us/step, done {best[0] / n * 1e6:6.2f} us/step -> {E * n / best[0]:.3e} env-steps/s', flush=True)
        env.close()

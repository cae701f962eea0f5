# how much of the pushing kernel's time is the tail of exhausted object searches (mover starts for which NO object position keeps min_mo_dist:
# 0.75 % of resets; the reference loops for ever there, the engine tries max_reset_trials candidates)?  usage: push_tail.py
import sys, numpy as np, torch
sys.path.insert(0, '.')
import gymnasium_planar_robotics_amd as gpr
for NM in (1, 2):
    for E in (4096, 65536):
        for trials in (10000, 64):
            env = gpr.VecBenchmarkPushingEnv(num_envs=E, num_movers=NM, max_reset_trials=trials, warn_on_reset=False); env.reset(seed=0)
            env.elapsed_steps.copy_(torch.randint(0, 50, (E,), device='cuda', dtype=torch.int32))
            acts = [(torch.rand(E, 2 * NM, device='cuda') * 20 - 10) for _ in range(8)]
            for i in range(30): env.step(acts[i % 8])
            torch.cuda.synchronize()
            best = 1e9
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for i in range(100): env.step(acts[i % 8])
                e1.record(); torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) * 10)
            print(f'pushing {NM} mover(s), {E} envs, max_reset_trials {trials}: {best:.1f} us/step  exhausted {env.diagnostics()["reset_exhausted"]}', flush=True)
            env._diag_seen = env.diagnostics(); env.close()

# one-off soak of the thin launches (k_step_thin: blocks of 10 / 8 / 8 cycles) on the final build: thin (default budget) against whole
# wavefronts (set_wave_budget(0)) against cycle-by-cycle evaluation of everything (GPR_EAGER_NOISE=1), bit for bit, 300 steps each
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
import gymnasium_planar_robotics_amd as gpr
for M, tiles, E in ((4, 6, 4096), (8, 8, 8192), (16, 10, 4096), (2, 4, 6000), (3, 5, 5000)):
    lay = np.ones((tiles, tiles))
    os.environ.pop('GPR_EAGER_NOISE', None)
    thin = gpr.VecBenchmarkPlanningEnv(lay, M, num_envs=E)
    whole = gpr.VecBenchmarkPlanningEnv(lay, M, num_envs=E); whole.set_wave_budget(0)
    os.environ['GPR_EAGER_NOISE'] = '1'
    eager = gpr.VecBenchmarkPlanningEnv(lay, M, num_envs=E)
    os.environ.pop('GPR_EAGER_NOISE', None)
    envs = [thin, whole, eager]
    for e in envs: e.reset(seed=3)
    g = torch.Generator(device='cuda').manual_seed(7)
    diff, done = 0, 0
    for t in range(300):
        a = torch.rand(E, 2 * M, device='cuda', generator=g) * 20 - 10
        if t % 4 == 0: a = a.sign() * 10
        res = [e.step(a) for e in envs]
        same = all(torch.equal(res[0][0][k], r[0][k]) for r in res[1:] for k in res[0][0]) and all(torch.equal(res[0][i], r[i]) for r in res[1:] for i in (1, 2, 3))
        same = same and all(torch.equal(thin.pos, e.pos) and torch.equal(thin.vel, e.vel) for e in envs[1:])
        diff += 0 if same else 1
        done += int((res[0][2] | res[0][3]).sum())
    print(f'{M} movers x {E} envs: 300 steps, steps with any difference between thin / whole / eager: {diff}, terminations {done}', flush=True)
    for e in envs: e.close()

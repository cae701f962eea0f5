# one-off soak: block loop (lazy noise) vs cycle-by-cycle evaluation of everything (GPR_EAGER_NOISE=1), full batch, bit for bit
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
import gymnasium_planar_robotics_amd as gpr
E, M, steps = 65536, 4, 150
for name, layout, kw in (('6x6', np.ones((6, 6)), {}), ('6x6 box', np.ones((6, 6)), {'collision_params': {'shape': 'box', 'size': np.array([0.0775, 0.0775]), 'offset': 0.001}}),
                         ('holes', np.array([[1, 1, 1, 1, 1, 0], [1, 0, 1, 1, 1, 1], [1, 1, 0, 1, 1, 1], [1, 1, 1, 1, 0, 1], [0, 1, 1, 1, 1, 1], [1, 1, 1, 0, 1, 1]]), {})):
    os.environ.pop('GPR_EAGER_NOISE', None)
    lazy = gpr.VecBenchmarkPlanningEnv(layout, M, num_envs=E, std_noise=np.array([2e-5, 1e-5, 0.0]), **kw)
    os.environ['GPR_EAGER_NOISE'] = '1'
    eager = gpr.VecBenchmarkPlanningEnv(layout, M, num_envs=E, std_noise=np.array([2e-5, 1e-5, 0.0]), **kw)
    os.environ.pop('GPR_EAGER_NOISE', None)
    lazy.reset(seed=3); eager.reset(seed=3)
    g = torch.Generator(device='cuda').manual_seed(1)
    bad = 0; done = 0
    for s in range(steps):
        a = (torch.rand(E, 2 * M, device='cuda', generator=g) * 2 - 1) * lazy.a_max
        if s % 4 == 1: a = a.sign() * lazy.a_max
        if s % 4 == 2: a = a * 0.05
        ol = lazy.step(a); oe = eager.step(a)
        same = all(torch.equal(ol[0][k], oe[0][k]) for k in ('observation', 'achieved_goal', 'desired_goal')) and torch.equal(ol[1], oe[1]) \
            and torch.equal(ol[2], oe[2]) and torch.equal(ol[3], oe[3]) and torch.equal(lazy.pos, eager.pos) and torch.equal(lazy.vel, eager.vel)
        bad += 0 if same else 1; done += int(ol[2].sum())
    print(f'{name}: {steps} steps x {E} envs, steps with any difference: {bad}, terminations {done}', flush=True)
    lazy.close(); eager.close()

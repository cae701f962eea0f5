# one-off soak: the f64 build free-running against the fp64 oracle (no teacher forcing), every decision and every auto-reset sample
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import gymnasium_planar_robotics_amd as gpr
from oracle import oracle
HOLES = np.array([[1, 1, 1, 1, 1], [1, 1, 0, 1, 1], [1, 1, 1, 1, 1], [1, 0, 1, 1, 1], [1, 1, 1, 1, 0]])
g = lambda t: t.double().cpu().numpy() if t.dtype in (torch.float32, torch.float64) else t.cpu().numpy()  # noqa: E731
E, steps = 4096, 200
for name, layout, M, kw in (('6x6 circles', np.ones((6, 6)), 4, {}),
                            ('holes', HOLES, 3, {'collision_params': {'shape': 'circle', 'size': 0.1, 'offset': 0.005, 'offset_wall': 0.002}}),
                            ('box', np.ones((5, 5)), 3, {'collision_params': {'shape': 'box', 'size': np.array([0.0775, 0.0775]), 'offset': 0.002}}),
                            ('jerk', np.ones((6, 6)), 4, {'learn_jerk': True}), ('8 movers', np.ones((8, 8)), 8, {}),
                            ('16 movers (shared reset searches)', np.ones((10, 10)), 16, {}), ('12 movers on 8x8 (padded groups, long searches)', np.ones((8, 8)), 12, {})):
    if len(sys.argv) > 1 and sys.argv[1] not in name: continue
    genv = gpr.VecBenchmarkPlanningEnv(layout, M, num_envs=E, precision='f64', **kw)
    oenv = oracle.OraclePlanningEnv(layout, M, E, autoreset='same_step', max_episode_steps=50, storage_f32=False, std_noise=1e-5, num_threads=16, **kw)
    genv.reset(seed=11); oenv.reset(seed=11)
    rng = np.random.default_rng(5); lim = genv.j_max if genv.learn_jerk else genv.a_max
    mism = 0; worst = 0.0; done = 0
    for s in range(steps):
        a = rng.uniform(-lim, lim, size=(E, 2 * M))
        obs, rew, term, trunc, info = genv.step(torch.as_tensor(a, device='cuda'))
        o = oenv.step(a)
        mism += int((g(term) != o['terminated']).sum() + (g(trunc) != o['truncated']).sum() + (g(rew) != o['reward']).sum())
        worst = max(worst, float(np.abs(g(obs['achieved_goal']) - o['achieved_goal']).max()), float(np.abs(g(obs['observation']) - o['observation']).max()),
                    float(np.abs(g(obs['desired_goal']) - o['desired_goal']).max()))
        done += int((o['terminated'] | o['truncated']).sum())
    print(f'{name}: {steps} steps x {E} envs, flag/reward mismatches {mism}, largest |obs - oracle| {worst:.3g}, episodes ended {done}', flush=True)
    genv.close()

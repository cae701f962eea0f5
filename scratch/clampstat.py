import numpy as np, sys
sys.path.insert(0, '.')
from oracle import oracle
E, M = 4096, 4
env = oracle.OraclePlanningEnv(np.ones((6, 6)), M, E, std_noise=0.0, num_threads=8)
env.reset(seed=0)
rng = np.random.default_rng(0)
fr_l, fr_w, ep = [], [], []
for step in range(60):
    a = rng.uniform(-10, 10, size=(E, 2 * M))
    o = env.step(a)
    v = env.vel.reshape(E, M, 2)
    sp = np.linalg.norm(v, axis=2)
    at = sp >= 2.0 - 1e-4                      # at the speed limit at the end of the step
    fr_l.append(at.mean()); fr_w.append(at.reshape(-1, 16 * M).any(axis=1).mean())
print('fraction of movers at the speed limit at step end: %.3f; waves (16 envs) with at least one: %.3f' % (np.mean(fr_l[10:]), np.mean(fr_w[10:])))
print('speed quantiles', np.quantile(sp, [0.5, 0.9, 0.99]))

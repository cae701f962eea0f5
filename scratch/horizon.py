# how often would a wave (16 envs x 4 movers) need its pair / wall checks if a check were skipped while provably unnecessary?
import numpy as np, sys
sys.path.insert(0, '.')
from oracle import oracle
E, M = 4096, 4
env = oracle.OraclePlanningEnv(np.ones((6, 6)), M, E, std_noise=0.0, num_threads=8)
env.reset(seed=0)
rng = np.random.default_rng(0)
thr, vmax, dt = 0.22, 2.0, 1e-3
W, r = 1.44, 0.11
stats = []
for step in range(40):
    env.step(rng.uniform(-10, 10, size=(E, 2 * M)))
    p = env.pos.reshape(E, M, 2)
    d = np.linalg.norm(p[:, :, None] - p[:, None, :], axis=-1) + np.eye(M) * 10
    pair_slack = d.min(axis=(1, 2)) - thr                       # per env
    wall_slack = np.minimum(p.min(axis=(1, 2)) - r, (W - r) - p.max(axis=(1, 2)))
    hp = np.floor(pair_slack / (2 * vmax * dt))                 # cycles until a pair collision is possible
    hw = np.floor(wall_slack / (vmax * dt))
    for g in (1, 4, 16):
        hpw = hp.reshape(-1, g).min(axis=1); hww = hw.reshape(-1, g).min(axis=1)
        # fraction of the 40 cycles that need the check with a single horizon evaluated at step start
        stats.append((g, np.clip(40 - hpw, 0, 40).mean() / 40, np.clip(40 - hww, 0, 40).mean() / 40))
stats = np.array(stats)
for g in (1, 4, 16):
    s = stats[stats[:, 0] == g]
    print(f'envs per group {g:2d}: pair checks needed {s[:,1].mean():.2f}, wall checks needed {s[:,2].mean():.2f}')

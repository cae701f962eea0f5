# rolling skip horizon: after a check with slack s, no collision is possible for floor(s / max closing speed / dt) cycles
import numpy as np, sys
sys.path.insert(0, '.')
from oracle import oracle
E, M = 2048, 4
env = oracle.OraclePlanningEnv(np.ones((6, 6)), M, E, std_noise=0.0, num_threads=8, num_cycles=1, max_episode_steps=2000)
env.reset(seed=0)
rng = np.random.default_rng(0)
thr, vmax, dt = 0.22, 2.0, 1e-3
W, r = 1.44, 0.11
groups = (1, 4, 16)
skip_p = {g: np.zeros(E // g, dtype=int) for g in groups}; skip_w = {g: np.zeros(E // g, dtype=int) for g in groups}
n_p = {g: 0 for g in groups}; n_w = {g: 0 for g in groups}; total = 0
for step in range(30):
    a = rng.uniform(-10, 10, size=(E, 2 * M))
    for cyc in range(40):
        o = env.step(a)
        p = env.pos.reshape(E, M, 2)
        d = np.linalg.norm(p[:, :, None] - p[:, None, :], axis=-1) + np.eye(M) * 10
        hp = np.floor(np.maximum(d.min(axis=(1, 2)) - thr, 0) / (2 * vmax * dt)).astype(int)
        hw = np.floor(np.maximum(np.minimum(p.min(axis=(1, 2)) - r, (W - r) - p.max(axis=(1, 2))), 0) / (vmax * dt)).astype(int)
        if step >= 5:
            total += 1
        for g in groups:
            for skip, h, n in ((skip_p[g], hp, n_p), (skip_w[g], hw, n_w)):
                due = skip == 0
                if step >= 5:
                    n[g] += due.mean()
                hg = h.reshape(-1, g).min(axis=1)
                skip[:] = np.where(due, hg, skip - 1)
                # an env that was reset must be checked again (new positions): force a check for its group
                rs = (o['terminated'] | o['truncated']).reshape(-1, g).any(axis=1)
                skip[rs] = 0
for g in groups:
    print(f'envs per group {g:2d}: pair checks run in {n_p[g]/total:.2f} of cycles, wall checks in {n_w[g]/total:.2f}')

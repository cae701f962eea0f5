# combined wall+pair check with a skip ladder: fraction of cycles in which the check runs, per ladder
import numpy as np, sys
sys.path.insert(0, '.')
from oracle import oracle
E, M, G = 2048, 4, 16
env = oracle.OraclePlanningEnv(np.ones((6, 6)), M, E, std_noise=0.0, num_threads=8, num_cycles=1, max_episode_steps=2000)
env.reset(seed=0)
rng = np.random.default_rng(0)
thr, vmax, dt = 0.22, 2.0, 1e-3
W, r = 1.44, 0.11
ladders = {'exact': None, '2,6,16': (2, 6, 16), '3,10': (3, 10), '4': (4,), '2,5,12,30': (2, 5, 12, 30), '1,3,8,20': (1, 3, 8, 20), '3,8,20': (3, 8, 20)}
skip = {k: np.zeros(E // G, dtype=int) for k in ladders}
n = {k: 0.0 for k in ladders}; total = 0
frozen = np.zeros(E, dtype=bool)
for step in range(30):
    a = rng.uniform(-10, 10, size=(E, 2 * M))
    frozen[:] = False
    for cyc in range(40):
        o = env.step(a)
        p = env.pos.reshape(E, M, 2)
        d = np.linalg.norm(p[:, :, None] - p[:, None, :], axis=-1) + np.eye(M) * 10
        hp = np.floor(np.maximum(d.min(axis=(1, 2)) - thr, 0) / (2 * vmax * dt)).astype(int)
        hw = np.floor(np.maximum(np.minimum(p.min(axis=(1, 2)) - r, (W - r) - p.max(axis=(1, 2))), 0) / (vmax * dt)).astype(int)
        done = o['terminated'] | o['truncated']
        h = np.minimum(hp, hw)
        h[frozen] = 10**6               # frozen envs do not vote (in the kernel they stay frozen until the step ends)
        frozen |= done                  # (the oracle env resets immediately; treat it as frozen for the rest of the 40 cycles)
        hg = h.reshape(-1, G).min(axis=1)
        if step >= 5:
            total += 1
        for k, lad in ladders.items():
            due = skip[k] == 0
            if step >= 5:
                n[k] += due.mean()
            if lad is None:
                s = hg
            else:
                s = np.zeros_like(hg)
                for rung in lad:
                    s = np.where(hg >= rung, rung, s)
            skip[k][:] = np.where(due, np.minimum(s, 10**5), skip[k] - 1)
    for k in ladders:
        skip[k][:] = 0  # new step: new action, check at the first cycle
for k in ladders:
    print(f'ladder {k:12s}: combined wall+pair check runs in {n[k]/total:.2f} of cycles')

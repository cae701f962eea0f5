# every planning shape of DESIGN.md section 6 as 1 / 2 / 4 sub-batches (bench.py's timed region: rollout buffer, desynchronised episodes)
import argparse, sys, numpy as np
sys.path.insert(0, '.')
import bench
args = argparse.Namespace(movers=4, tiles=6, precision='f32', steps=200, warmup=60, min_seconds=0.1, extra_seconds=0.1, regions=1)
box = {'shape': 'box', 'size': np.array([0.0775, 0.0775])}
HOLES = np.ones((6, 6)); HOLES[2, 3] = 0; HOLES[4, 1] = 0
shapes = [('4 movers f32', dict(envs=65536)), ('8 movers', dict(envs=65536, movers=8, tiles=8)), ('16 movers', dict(envs=32768, movers=16, tiles=10)),
          ('box', dict(envs=65536, collision_params=box)), ('f64', dict(envs=65536, precision='f64')), ('f32_f64acc', dict(envs=65536, precision='f32_f64acc')),
          ('4 movers 8192 envs', dict(envs=8192)), ('16 movers 4096 envs', dict(envs=4096, movers=16, tiles=10)),
          ('jerk', dict(envs=65536, learn_jerk=True)), ('holes', dict(envs=65536, layout=HOLES))]
which = sys.argv[1:]
for name, kw in shapes:
    if which and not any(w in name for w in which): continue
    out = []
    for S in (1, 2, 4):
        r = bench.timed(args, 0, 0, 1, sub=S, **kw)
        out.append(f'S={S}: {r["step_us"]:.1f} us')
    print(f'{name:22s} ' + ' | '.join(out), flush=True)

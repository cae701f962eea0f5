# A/B of library builds (GPR_LIB_PATH) on bench.py's timed region, by workload and sub-batches: ab_sub.py "<workloads>" "<S values>"
import argparse, os, sys, numpy as np
sys.path.insert(0, '.')
import bench
args = argparse.Namespace(movers=4, tiles=6, precision='f32', steps=300, warmup=60, min_seconds=0.3, extra_seconds=0.3, regions=1)
HOLES = np.ones((6, 6)); HOLES[2, 3] = 0; HOLES[4, 1] = 0
box = {'shape': 'box', 'size': np.array([0.0775, 0.0775])}
CFG = {'m4': dict(envs=65536), 'm8': dict(envs=65536, movers=8, tiles=8), 'm16': dict(envs=32768, movers=16, tiles=10), 'box': dict(envs=65536, collision_params=box),
       'f64': dict(envs=65536, precision='f64'), 'f64acc': dict(envs=65536, precision='f32_f64acc'), 'm4s': dict(envs=8192), 'm16s': dict(envs=4096, movers=16, tiles=10), 'm4c2': dict(envs=4096), 'm8s': dict(envs=8192, movers=8, tiles=8), 'holes': dict(envs=65536, layout=HOLES),
       'jerk': dict(envs=65536, learn_jerk=True), 'm2': dict(envs=65536, movers=2, tiles=4)}
out = []
for name in sys.argv[1].split():
    for S in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else '1 2').split()]:
        r = bench.timed(args, 0, 0, 1, sub=S, **CFG[name])
        out.append(f'{name} S={S}: {r["step_us"]:.2f}')
print(os.path.basename(os.environ.get('GPR_LIB_PATH', 'default')), ' | '.join(out), flush=True)

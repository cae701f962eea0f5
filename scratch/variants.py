# us per step of the planning step kernel on several configurations, for the library build GPR_LIB_PATH names (A/B runs: scratch/ab.sh)
# usage: variants.py [m4 box f64 f64acc jerk holes m8 m16 ...]
import sys, os, numpy as np, torch
sys.path.insert(0, '.')
import gymnasium_planar_robotics_amd as gpr
CFG = {
    'm4': dict(E=65536, M=4, tiles=6), 'm8': dict(E=65536, M=8, tiles=8), 'm16': dict(E=32768, M=16, tiles=10),
    'box': dict(E=65536, M=4, tiles=6, collision_params={'shape': 'box', 'size': np.array([0.0775, 0.0775])}),
    'f64': dict(E=65536, M=4, tiles=6, precision='f64'), 'f64acc': dict(E=65536, M=4, tiles=6, precision='f32_f64acc'),
    'jerk': dict(E=65536, M=4, tiles=6, precision='f64', learn_jerk=True), 'holes': dict(E=65536, M=4, tiles=6, holes=True),
    # per-GPU shards of BASELINE configs 4 and 5 on 8 GPUs, and config 2
    'm8s': dict(E=8192, M=8, tiles=8), 'm16s': dict(E=4096, M=16, tiles=10), 'm4s': dict(E=4096, M=4, tiles=6),
}
for name in (sys.argv[1:] or ['m4', 'box', 'f64']):
    kw = dict(CFG[name]); E, M, tiles = kw.pop('E'), kw.pop('M'), kw.pop('tiles')
    layout = np.ones((tiles, tiles))
    if kw.pop('holes', False): layout[2, 2] = 0; layout[3, 4] = 0
    env = gpr.VecBenchmarkPlanningEnv(layout, M, num_envs=E, **kw); env.reset(seed=0)
    env.elapsed_steps.copy_(torch.randint(0, 50, (E,), device='cuda', dtype=torch.int32))
    lim = env.j_max if env.learn_jerk else env.a_max
    acts = [((torch.rand(E, 2 * M, device='cuda') * 2 - 1) * lim).to(env.dtype) for _ in range(8)]
    for i in range(100): env.step_into(acts[i % 8], env._out)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(400): env.step_into(acts[i % 8], env._out)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 2.5)
    print(f'{os.path.basename(os.environ.get("GPR_LIB_PATH", "default"))} {name}: {best:.2f} us/step', flush=True)
    env.close()

# every kernel of the engine on its BASELINE.json configuration, a few dozen launches each: the workload the round's rocprofv3
# passes (scratch/profile_round3.sh) run. Prints nothing the profiles need; the kernel names identify the workloads.
import sys, os, numpy as np, torch
sys.path.insert(0, '.')
import gymnasium_planar_robotics_amd as gpr
G = gpr.geometry_2D_utils
N = int(os.environ.get('PROF_STEPS', 60))
dev = 'cuda'

def planning(E, M, tiles, **kw):
    env = gpr.VecBenchmarkPlanningEnv(np.ones((tiles, tiles)), M, num_envs=E, **kw); env.reset(seed=0)
    env.elapsed_steps.copy_(torch.randint(0, 50, (E,), device=dev, dtype=torch.int32))
    lim = env.j_max if env.learn_jerk else env.a_max
    acts = [((torch.rand(E, 2 * M, device=dev) * 2 - 1) * lim).to(env.dtype) for _ in range(8)]
    for i in range(N): env.step_into(acts[i % 8], env._out)
    torch.cuda.synchronize(); env.close()

def pushing(E, **kw):
    env = gpr.VecBenchmarkPushingEnv(num_envs=E, **kw); env.reset(seed=0)
    n_act = env.single_action_space.shape[0]
    acts = [(torch.rand(E, n_act, device=dev) * 20 - 10).to(env.dtype) for _ in range(8)]
    for i in range(N): env.step(acts[i % 8])
    torch.cuda.synchronize(); env.close()

def parallel(E, M, tiles):
    env = gpr.VecBenchmarkPlanningParallelEnv(np.ones((tiles, tiles)), M, num_envs=E); env.reset(seed=0)
    a = torch.rand(E, 2 * M, device=dev) * 20 - 10
    for i in range(N): env.step(a)
    torch.cuda.synchronize(); env.close()

def operators(n=1 << 20):
    env = gpr.VecBenchmarkPlanningEnv(np.ones((6, 6)), 4, num_envs=16)
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: torch.rand(*s, device=dev, generator=g, dtype=torch.float64)  # noqa: E731
    qpos = torch.zeros(n, 7, device=dev, dtype=torch.float64); qpos[:, :2] = r(n, 2) * 1.44; qpos[:, 3] = 1.0
    sets = torch.zeros(n // 4, 4, 7, device=dev, dtype=torch.float64); sets[..., :2] = r(n // 4, 4, 2) * 1.44; sets[..., 3] = 1.0
    p1, p2, q1, q2 = r(n, 2), r(n, 2), r(n, 2), r(n, 2)
    size = torch.full((n, 2), 0.0775, device=dev, dtype=torch.float64)
    ag, dg = (r(n, 8) * 1.44).float(), (r(n, 8) * 1.44).float()
    info = {'mover_collision': torch.zeros(n, device=dev, dtype=torch.bool), 'wall_collision': torch.zeros(n, device=dev, dtype=torch.bool)}
    for _ in range(10):
        env.qpos_is_valid(qpos, 0.11)
        env.check_mover_collision(sets, 0.11)
        G.check_line_segments_intersect(p1, p2, q1, q2)
        G.get_2D_rect_vertices(qpos, size)
        G.check_rectangles_intersect(qpos, torch.roll(qpos, 1, 0), size, size)
        env.ensure_max_dyn_val(p1, 2.0, p2)
        env.compute_reward(ag, dg, info)
    torch.cuda.synchronize(); env.close()

def rollout(E, M, tiles, T=50, launches=6):
    env = gpr.VecBenchmarkPlanningEnv(np.ones((tiles, tiles)), M, num_envs=E); env.reset(seed=0)
    env.elapsed_steps.copy_(torch.randint(0, 50, (E,), device=dev, dtype=torch.int32))
    acts = (torch.rand(T, E, 2 * M, device=dev) * 20 - 10)
    roll = env.alloc_rollout(T)
    for _ in range(launches): env.step_n(acts, roll)
    torch.cuda.synchronize(); env.close()

which = sys.argv[1:] or ['m4', 'm8', 'm16', 'box', 'f64', 'f64acc', 'jerk', 'holes', 'push', 'push2', 'views', 'ops', 'stepn']
if 'm4' in which: planning(65536, 4, 6)
if 'm8' in which: planning(65536, 8, 8)
if 'm16' in which: planning(32768, 16, 10)
if 'box' in which: planning(65536, 4, 6, collision_params={'shape': 'box', 'size': np.array([0.0775, 0.0775])})
if 'f64' in which: planning(65536, 4, 6, precision='f64')
if 'f64acc' in which: planning(65536, 4, 6, precision='f32_f64acc')
if 'jerk' in which: planning(65536, 4, 6, learn_jerk=True)
if 'holes' in which:
    lay = np.ones((6, 6)); lay[2, 3] = 0; lay[4, 1] = 0
    env = gpr.VecBenchmarkPlanningEnv(lay, 4, num_envs=65536); env.reset(seed=0)
    acts = [(torch.rand(65536, 8, device=dev) * 20 - 10) for _ in range(8)]
    for i in range(N): env.step_into(acts[i % 8], env._out)
    torch.cuda.synchronize(); env.close()
if 'push' in which: pushing(65536); pushing(4096)
if 'push2' in which: pushing(65536, num_movers=2); pushing(4096, num_movers=2)
if 'views' in which: parallel(32768, 16, 10)
if 'ops' in which: operators()
if 'stepn' in which: rollout(65536, 4, 6); rollout(8192, 4, 6)

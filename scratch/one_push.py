# usage: one_push.py NM steps [envs] -- a plain stepping loop of the pushing env for profiling runs
import sys, os, numpy as np, torch
sys.path.insert(0, '.')
import gymnasium_planar_robotics_amd as gpr
NM = int(sys.argv[1]) if len(sys.argv) > 1 else 1; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30; E = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
env = gpr.VecBenchmarkPushingEnv(num_envs=E, num_movers=NM, precision=os.environ.get('PREC', 'f32')); env.reset(seed=0)
env.elapsed_steps.copy_(torch.randint(0, 50, (E,), device='cuda', dtype=torch.int32))
acts = [(torch.rand(E, 2 * NM, device='cuda') * 20 - 10).to(env.dtype) for _ in range(8)]
for i in range(steps): env.step(acts[i % 8])
torch.cuda.synchronize()

# usage: one.py sigma steps num_cycles [envs] [movers] -- a plain stepping loop for profiling runs
import sys, os, numpy as np, torch
sys.path.insert(0,'.')
import gymnasium_planar_robotics_amd as gpr
sigma=float(sys.argv[1]); steps=int(sys.argv[2]) if len(sys.argv)>2 else 30; ncyc=int(sys.argv[3]) if len(sys.argv)>3 else 40
E=int(sys.argv[4]) if len(sys.argv)>4 else 65536; M=int(sys.argv[5]) if len(sys.argv)>5 else 4
tiles={1:3,2:4,4:6,8:8,16:10}.get(M,6)
env=gpr.VecBenchmarkPlanningEnv(np.ones((tiles,tiles)),M,num_envs=E,std_noise=sigma,num_cycles=ncyc,precision=os.environ.get('PREC','f32'))
env.reset(seed=0)
env.elapsed_steps.copy_(torch.randint(0, 50, (E,), device='cuda', dtype=torch.int32))
acts=[(torch.rand(E,2*M,device='cuda')*20-10).to(env.dtype) for _ in range(8)]
for i in range(steps): env.step_into(acts[i%8], env._out)
torch.cuda.synchronize()

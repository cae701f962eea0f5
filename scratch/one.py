import sys, os, numpy as np, torch
sys.path.insert(0,'.')
import gymnasium_planar_robotics_amd as gpr
sigma=float(sys.argv[1]); steps=int(sys.argv[2]) if len(sys.argv)>2 else 30; ncyc=int(sys.argv[3]) if len(sys.argv)>3 else 40
E,M=65536,4
env=gpr.VecBenchmarkPlanningEnv(np.ones((6,6)),M,num_envs=E,std_noise=sigma,num_cycles=ncyc)
env.reset(seed=0)
acts=[(torch.rand(E,2*M,device='cuda')*20-10).to(env.dtype) for _ in range(8)]
for i in range(steps): env.step_into(acts[i%8], env._out)
torch.cuda.synchronize()

import sys, os, numpy as np, torch
os.environ['GPR_DBG']='1'
sys.path.insert(0,'.')
import gymnasium_planar_robotics_amd as gpr
for sig in (0.0, 1e-5):
    E,M=65536,4
    env=gpr.VecBenchmarkPlanningEnv(np.ones((6,6)),M,num_envs=E,std_noise=sig)
    env.reset(seed=0)
    acts=[(torch.rand(E,2*M,device='cuda')*20-10).to(env.dtype) for _ in range(8)]
    for i in range(100): env.step_into(acts[i%8], env._out)
    torch.cuda.synchronize()
    print('sigma', sig, flush=True)
    env.close()

import sys, numpy as np, torch
sys.path.insert(0,'.')
import gymnasium_planar_robotics_amd as gpr
def run(sig, ncyc=40, steps=600, mes=50):
    E,M=65536,4
    env=gpr.VecBenchmarkPlanningEnv(np.ones((6,6)),M,num_envs=E,std_noise=np.array(sig),num_cycles=ncyc,max_episode_steps=mes)
    env.reset(seed=0)
    acts=[(torch.rand(E,2*M,device='cuda')*20-10).to(env.dtype) for _ in range(8)]
    for i in range(100): env.step_into(acts[i%8], env._out)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): env.step_into(acts[i%8], env._out)
    e1.record(); torch.cuda.synchronize()
    print(f'std_noise={sig} ncyc={ncyc}: {e0.elapsed_time(e1)*1e3/steps:7.2f} us/step', flush=True)
    env.close()
for sig in ([0,0,0],[1e-5,0,0],[0,1e-5,0],[1e-5,1e-5,0],[1e-5,1e-5,1e-5]):
    run(sig)
run([1e-5,1e-5,0], ncyc=0); run([0,0,0], ncyc=0)

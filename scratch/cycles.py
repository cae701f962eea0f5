import sys, time, numpy as np, torch
sys.path.insert(0,'.')
import gymnasium_planar_robotics_amd as gpr
def run(E, M, tiles, sigma, ncyc, steps=300, precision='f32', mes=50):
    env=gpr.VecBenchmarkPlanningEnv(np.ones((tiles,tiles)),M,num_envs=E,std_noise=sigma,precision=precision,num_cycles=ncyc,max_episode_steps=mes)
    env.reset(seed=0)
    acts=[(torch.rand(E,2*M,device='cuda')*20-10).to(env.dtype) for _ in range(8)]
    for i in range(20): env.step_into(acts[i%8], env._out)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps): env.step_into(acts[i%8], env._out)
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/steps
    done=float((env._terminated.float().mean()+env._truncated.float().mean()).item())
    print(f'E={E:7d} M={M:2d} sigma={sigma:g} ncyc={ncyc:4d}: {us:8.2f} us/step   done/step {done:.3f}', flush=True)
    env.close()
for E in (4096, 65536):
    for s in (0.0, 1e-5):
        for n in (0, 1, 10, 40, 160):
            run(E,4,6,s,n)
